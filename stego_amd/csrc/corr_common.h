// Shared device-side definitions for the correspondence-loss kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace stego {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TP = 128;        // sample points per side, padded (S*S <= 128)
constexpr int NTHREADS = 256;  // 4 wave64 per workgroup, one per SIMD
constexpr int KC = 64;         // channels per staged chunk
constexpr int LDA = KC + 4;    // stage row stride in floats: 272 B rows, conflict-free ds_read_b128
constexpr int LDT = 129;       // epilogue tile row stride (odd -> conflict-free column walks)

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
    int B, C, K, H, W, S, P, n_neg, n_sets;
    int mode;                                  // 0 = forward() semantics, 1 = helper() on pre-sampled maps
    int pointwise;
    float cmin, cmax;
    float shift[3];
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
    int g_neg_loss_stride;                     // 1 dense, 0 broadcast scalar
    int B, K, H, W, S, P, n_neg, n_sets;
    int mode;
    float cmin, cmax;
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

// Per-workgroup description of the two sides of a tile.
struct SideSel {
    const MapV* mf;      // feature map
    const MapV* mc;      // code map
    const float* coords; // [B][S][S][2] or null (direct)
    int img;
};

// Build tap tables for the 2 x 128 points of a tile.  tid<128: A point tid; else B point tid-128.
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

// Gather one chunk of channels [c0, c0+ncols) of 128 sampled points into an LDS tile
// dst[point][col] (row stride LD floats), blending the 4 bilinear taps on the fly and
// accumulating each point's sum of squares (for the L2 norm) in ss[].
// V = channels per lane-load (4/2 need channel stride 1 and 16/8-byte aligned pixels; 1 is generic).
// Thread mapping: SLOTS=KC/V lanes cover one point's chunk -> one contiguous KC*4-byte
// read per tap per point; a wave covers 64/SLOTS points.
template <int V, int LD>
__device__ __forceinline__ void gather_chunk(const float* __restrict__ img, int sc, const int4* __restrict__ tapo,
                                             const float4* __restrict__ tapw, int c0, int Ctot, int ncols, int P,
                                             float* __restrict__ dst, float (&ss)[TP * (KC / V) / NTHREADS])
{
    constexpr int SLOTS = KC / V;
    constexpr int ITEMS = TP * SLOTS / NTHREADS;
    constexpr int PPI = NTHREADS / SLOTS;
    const int tid = threadIdx.x;
    const int slot = tid % SLOTS, prow = tid / SLOTS;
    const int col = slot * V;
    const int ch = c0 + col;
    if (col >= ncols) return;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int q = it * PPI + prow;
        float v[V];
#pragma unroll
        for (int e = 0; e < V; ++e) v[e] = 0.f;
        if (q < P && ch < Ctot) {
            const int4 o = tapo[q];
            const float4 w = tapw[q];
            const float* base = img + (long long)ch * sc;
            if constexpr (V == 4) {
                const f32x4 t0 = *reinterpret_cast<const f32x4*>(base + o.x);
                const f32x4 t1 = *reinterpret_cast<const f32x4*>(base + o.y);
                const f32x4 t2 = *reinterpret_cast<const f32x4*>(base + o.z);
                const f32x4 t3 = *reinterpret_cast<const f32x4*>(base + o.w);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = w.x * t0[e] + w.y * t1[e] + w.z * t2[e] + w.w * t3[e];
            } else if constexpr (V == 2) {
                const f32x2 t0 = *reinterpret_cast<const f32x2*>(base + o.x);
                const f32x2 t1 = *reinterpret_cast<const f32x2*>(base + o.y);
                const f32x2 t2 = *reinterpret_cast<const f32x2*>(base + o.z);
                const f32x2 t3 = *reinterpret_cast<const f32x2*>(base + o.w);
#pragma unroll
                for (int e = 0; e < 2; ++e) v[e] = w.x * t0[e] + w.y * t1[e] + w.z * t2[e] + w.w * t3[e];
            } else {
                v[0] = w.x * base[o.x] + w.y * base[o.y] + w.z * base[o.z] + w.w * base[o.w];
            }
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < V; ++e) s += v[e] * v[e];
            ss[it] += s;
        }
        float* d = dst + q * LD + col;
        if constexpr (V == 4) {
            *reinterpret_cast<f32x4*>(d) = f32x4{v[0], v[1], v[2], v[3]};
        } else if constexpr (V == 2) {
            *reinterpret_cast<f32x2*>(d) = f32x2{v[0], v[1]};
        } else {
            d[0] = v[0];
        }
    }
}

// Reduce the per-thread sum-of-squares partials over the SLOTS lanes that share a point and
// publish nrm (=||t||) for the points this thread group owns.
template <int V>
__device__ __forceinline__ void publish_norms(float (&ss)[TP * (KC / V) / NTHREADS], float* __restrict__ nrm_out)
{
    constexpr int SLOTS = KC / V;
    constexpr int ITEMS = TP * SLOTS / NTHREADS;
    constexpr int PPI = NTHREADS / SLOTS;
    const int tid = threadIdx.x;
    const int slot = tid % SLOTS, prow = tid / SLOTS;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        float s = ss[it];
#pragma unroll
        for (int m = SLOTS / 2; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
        if (slot == 0) nrm_out[it * PPI + prow] = sqrtf(s);
    }
}

__device__ __forceinline__ float block_sum(float v, float* red /*>=4 floats*/)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

}  // namespace stego
