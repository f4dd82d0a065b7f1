// Shared device-side definitions for the correspondence-loss kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace stego {

// Workgroup barrier that first drains this wave's LDS-DMA copies.  global_load_lds writes LDS asynchronously and is
// tracked by vmcnt; the workgroup-scope fence of __syncthreads() only promises lgkmcnt(0), so whether the compiler
// also waits for vmcnt before the barrier depends on its alias analysis of the LDS reads that follow (it did in most
// builds of these kernels, not in all).  Every barrier that publishes an async copy to other waves goes through here.
__device__ __forceinline__ void sync_after_lds_dma()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}


typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half_t;                        // element type of the split operands (see split_f16_pair)
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int TP = 128;        // sample points per side, padded (S*S <= 128)
constexpr int NTHREADS = 256;  // 4 wave64 (the gather team size everywhere)
constexpr int KC = 64;         // channels per staged chunk
constexpr int LDA = KC + 4;    // f32 stage row stride in floats: 272 B rows, conflict-free ds_read_b128
constexpr int LDH = KC + 8;    // fp16 stage row stride in elements: 144 B rows, conflict-free ds_read_b128
constexpr int LDT = 129;       // epilogue tile row stride (odd -> conflict-free column walks)

enum { PREC_F32 = 0, PREC_F16X3 = 1 };

// float32 [N,C,H,W] view; strides in elements. Per-image offsets are < 2^31 (host-checked).
struct MapV {
    const float* p;
    long long sn;
    int sc, sh, sw;
};

struct CorrParams {
    MapV feats, feats_pos, code, code_pos;     // mode 1 (helper): f1, f2, c1, c2
    const float* coords1;
    const float* coords2;
    const long long* perms;                    // [n_neg][B]
    float* intra_cd;                           // [B][P*P]
    float* inter_cd;                           // [B][P*P]
    float* neg_loss;                           // [n_neg*B][P*P]   (mode 1: loss)
    float* neg_cd;                             // [n_neg*B][P*P]   (mode 1: cd)
    float* saved_w;                            // [n_sets*B][P*P] or null
    float* saved_mean;                         // [n_sets] or null
    float* loss_means;                         // [2] or null (mode 1)
    float* stats;                              // workspace [n_sets*B][4]
    const void* fs;                            // sampler outputs: normalised ANCHOR features as LDS images [B][NCH][..]
    const float* cs;                           //                  normalised sampled codes of every set [nset][128][LDK]
    const int4* tapyx;                         //                  bilinear tap pixels  [nset][128]
    const float4* tapw;                        //                  bilinear tap weights [nset][128]
    int NCH, KQ, LDK;
    int B, C, K, H, W, S, P, n_neg, n_sets;
    int mode;                                  // 0 = forward() semantics, 1 = helper() on pre-sampled maps
    int pointwise;
    int debug;                                 // measurement ablations (STEGO_DEBUG env): 1 skip MFMA, 2 skip gather
    float cmin, cmax;
    float shift[3];
};

// Stage-1 (sampling) parameters; see corr_sample.hip.
struct SampleParams {
    MapV feats, feats_pos, code, code_pos;     // helper mode: f1, f2, c1, c2
    const float* coords1;
    const float* coords2;
    const long long* perms;                    // [n_neg][B]
    void* fs;                                  // f32: [B][NCH][128][LDA] float; f16x3: [B][NCH][2][128][LDH] fp16 (roles < feat_roles)
    float* cs;                                 // [nset][128][LDK] normalised sampled codes
    float* nrm;                                // [nset][128] code norms before normalisation
    int4* tapyx;                               // optional [nset][128] packed tap pixels (for the backward)
    float4* tapw;                              // optional [nset][128] tap weights
    int B, C, K, H, W, S, P;
    int n_roles;                               // 2 + n_neg (helper: 2)
    int feat_roles;                            // features are sampled for roles < feat_roles (1: anchors only)
    int NCH;                                   // ceil(C/64)
    int KQ, LDK;                               // round_up(K,8), KQ+4
    int mode;                                  // 0 forward(), 1 helper() (pixel-for-pixel)
    int div_by, div_magic;                     // q -> (q / div_by) by multiply-shift (set by the launcher)
    int debug;                                 // STEGO_DEBUG_SAMPLE: 1 skip stores, 2 skip feature loads, 4 skip code path
};

struct BwdParams {
    MapV code, code_pos;                       // mode 1: c1, c2
    const float* coords1;
    const float* coords2;
    const long long* perms;
    const float* saved_w;
    const float* saved_mean;
    const float* intra_cd;
    const float* inter_cd;
    const float* neg_cd;                       // mode 1: cd
    const float* g_intra;                      // device scalars (mode 0)
    const float* g_inter;
    const float* g_neg_loss;                   // mode 1: g_loss (dense)
    const float* g_intra_cd;
    const float* g_inter_cd;
    const float* g_neg_cd;                     // mode 1: g_cd
    float* d_code;                             // [B][H][W][K] channels-last dense (mode 1: d_c1)
    float* d_code_pos;                         //                                  (mode 1: d_c2)
    const float* cs;                           // forward's saved context: normalised sampled codes [nset][128][LDK]
    const float* nrm;                          //                          code norms [nset][128]
    const int4* tapyx;                         //                          bilinear tap pixels [nset][128]
    const float4* tapw;                        //                          bilinear tap weights [nset][128]
    float* dt;                                 // workspace: raw-sample gradients [n_tiles][2][128][LDK]
    // lists-first backward (corr_bwd_tile_build_kernel + corr_unsample_list_kernel): the unsample's entry lists, in the workspace
    unsigned* uslots;                          // per (destination, image, pixel row, 16-pixel bin): header {count, overflow start} + the first entries; null = row kernel
    unsigned* upool;                           // overflow entries {DT row (float index), x0, w_left, w_right}, 16 bytes each
    unsigned dt_bytes, uslots_bytes, upool_bytes;
    int n_build;                               // builder workgroups at the front of the tile launch
    int KQ, LDK;
    int g_neg_loss_stride;                     // 1 dense, 0 broadcast scalar
    int B, K, H, W, S, P, n_neg, n_sets;
    int mode;
    int precision;                             // PREC_*: F16X3 = the tile kernel's GEMMs as split-fp16 products (mode 0)
    int debug;                                 // 1 skip MFMA, 2 skip scatter, 4 / 16 load ablations, 8 stamps, 32 band unsample, 64 / 128 unsample ablations, 512 fp32-MFMA tile kernel, 1024 tile + row kernels instead of lists first
    float cmin, cmax;
};

// Fused forward (corr_fused.hip): one launch + a scalar kernel.
struct FusedParams {
    MapV feats, feats_pos, code, code_pos;
    const float* coords1;
    const float* coords2;
    const long long* perms;                    // [n_neg][B]
    float* intra_cd;                           // [B][P*P]
    float* inter_cd;                           // [B][P*P]
    float* neg_loss;                           // [n_neg*B][P*P]
    float* neg_cd;                             // [n_neg*B][P*P]
    float* saved_w;                            // [n_sets*B][P*P] or null
    float* saved_mean;                         // [n_sets] or null
    float* loss_means;                         // [2]
    float* stats;                              // [n_sets*B][4]: sum fd, sum lp, sum clamp, 1 = old_mean applied in-kernel
    unsigned* anchor_cnt;                      // [B] points published per anchor           } zeroed before the launch
    unsigned long long* gran;                  // [4][n_sets*B] {tag = 1, value} granules   }  (the last workgroup of a
    unsigned* done_cnt;                        // workgroups that finished                  }   launch zeroes them again)
    unsigned long long* rowg;                  // column-half launch (corr_fused_half.hip): [2 * n_sets * B][128] {tag, value} partial row sums of fd, zeroed by their one reader
    int n_anchor_wg;                           // column-half launch: workgroups at the front of the grid that only sample anchors (a multiple of 8)
    unsigned char* fs;                         // anchor feature operand stages [B][NCH2][16 KB] (ring format H, or F in f32 mode)
    unsigned char* csf;                        // anchor code operand stages    [B][NKC][16 KB]  (ring format F)
    float* cs;                                 // normalised sampled codes of every set [nset][128][LDK] (the backward's context)
    float* nrm;                                // [nset][128] code norms
    int4* tapyx;                               // [nset][128]
    float4* tapw;                              // [nset][128]
    unsigned fs_bytes, csf_bytes, cs_bytes;    // sizes of the fs / csf / cs regions (buffer descriptors)
    int NCH2, NKC, kper;                       // C / 32 feature stages; code stages of kper (multiple of 8, <= 32) channels
    int KQ, LDK;
    int B, C, K, H, W, S, P, n_neg, n_sets;
    int n_owner;                               // workgroups [0, n_owner) share phase 1
    int p1_light;                              // 1: the workgroups without a gather stream carry phase 1 (set by the launcher)
    int ps_round;                              // pair-sets per round of tiles (all of them unless there are more tiles than CUs)
    int pointwise;
    int debug;                                 // 32 never rendezvous (always repair), 64 owners skip phase 1 (always help), 128 even phase-1 shares,
                                               // 256 phase stamps, 16 every workgroup looks its tile up (round 6: the intra / inter tiles know theirs),
                                               // 16384 keep the full-tile launch where the column-half launch would be taken, 1 (column-half
                                               // launch only) no MFMAs - any other bit keeps the full-tile launch
    int timeout_ticks;                         // bound of every spin, 100 MHz ticks
    float cmin, cmax;
    float shift[3];
};

// Bilinear taps of ATen grid_sampler_2d (bilinear, padding_mode=border, align_corners=True),
// which is what the reference's sample() (modules.py:287-288) lowers to.
// Returns pixel coordinates packed as (y<<16)|x per tap and the 4 corner weights (nw,ne,sw,se).
__device__ __forceinline__ void make_taps(float x, float y, int H, int W, int4& yx, float4& w)
{
    float ix = ((x + 1.f) * 0.5f) * (float)(W - 1);
    float iy = ((y + 1.f) * 0.5f) * (float)(H - 1);
    ix = fminf((float)(W - 1), fmaxf(ix, 0.f));
    iy = fminf((float)(H - 1), fmaxf(iy, 0.f));
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    int x0 = (int)fx0, y0 = (int)fy0;
    int x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - fx0, wx0 = (fx0 + 1.f) - ix;
    const float wy1 = iy - fy0, wy0 = (fy0 + 1.f) - iy;
    float nw = wx0 * wy0, ne = wx1 * wy0, sw = wx0 * wy1, se = wx1 * wy1;
    if (x1 > W - 1) { ne = 0.f; se = 0.f; x1 = x0; }     // out-of-range taps contribute zero
    if (y1 > H - 1) { sw = 0.f; se = 0.f; y1 = y0; }
    yx = make_int4((y0 << 16) | x0, (y0 << 16) | x1, (y1 << 16) | x0, (y1 << 16) | x1);
    w = make_float4(nw, ne, sw, se);
}

__device__ __forceinline__ int4 taps_to_offsets(const int4 yx, int sh, int sw)
{
    return make_int4((yx.x >> 16) * sh + (yx.x & 0xffff) * sw, (yx.y >> 16) * sh + (yx.y & 0xffff) * sw,
                     (yx.z >> 16) * sh + (yx.z & 0xffff) * sw, (yx.w >> 16) * sh + (yx.w & 0xffff) * sw);
}

// Tap table entry of one sample point.  Padding points (q >= P) get offset 0 / weight 0: their
// loads are harmless re-reads of pixel (0,0) and they only ever feed padded rows/cols.
// forward mode: point q=(h,w) samples coords[b][w][h]  (sample() permutes the grid, modules.py:288)
// direct mode : point q=(h,w) IS pixel (h,w) of an already-sampled [N,C,S1,S2] map.
__device__ __forceinline__ void tap_for_point(int q, int P, int S, int H, int W, bool direct,
                                              const float* coords_img, int4& yx, float4& w)
{
    yx = make_int4(0, 0, 0, 0);
    w = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q >= P) return;
    if (direct) {
        const int hh = q / W, ww = q - hh * W;
        const int v = (hh << 16) | ww;
        yx = make_int4(v, v, v, v);
        w = make_float4(1.f, 0.f, 0.f, 0.f);
    } else {
        const int hh = q / S, ww = q - hh * S;
        const float* c = coords_img + (size_t)(ww * S + hh) * 2;
        make_taps(c[0], c[1], H, W, yx, w);
    }
}

template <int V> struct VecT;
template <> struct VecT<4> { typedef f32x4 type; };
template <> struct VecT<2> { typedef f32x2 type; };
template <> struct VecT<1> { typedef float type; };

// fp32 -> (hi, lo) fp16 split (v_cvt_pk_f16_f32):  x = hi + lo + e with |e| <= 2^-22 |x| for |x| in the fp16 normal
// range (11 + 11 mantissa bits; below it the absolute error is < 2^-25).  hi*hi + hi*lo + lo*hi on the fp16 matrix
// cores (3 MFMAs, fp32 accumulate) therefore carries products to ~2^-21: fp32-grade, at 3/16 of the cost of the
// fp32 MFMA (which runs at the VALU rate on gfx950).  Callers keep |x| < 65504 (raw gathered features are scaled by
// 2^-4 first, their L2 normalisation cancels the scale).  Returns the two packed dwords for a pair of values.
__device__ __forceinline__ void split_f16_pair(float x, float y, unsigned& hi, unsigned& lo)
{
    const f16x2 h = __builtin_convertvector(f32x2{x, y}, f16x2);
    hi = __builtin_bit_cast(unsigned, h);
    const f32x2 hf = __builtin_convertvector(h, f32x2);
    const f16x2 l = __builtin_convertvector(f32x2{x - hf[0], y - hf[1]}, f16x2);
    lo = __builtin_bit_cast(unsigned, l);
}

}  // namespace stego
