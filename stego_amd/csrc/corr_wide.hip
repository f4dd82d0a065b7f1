// ContrastiveCorrelationLoss for point sets larger than one 128 x 128 tile: cfg.feature_samples 12 .. 16 (144 .. 256 points per image) - gfx950.
//
//   reference: src/modules.py:349-398 (forward), :325-347 (helper), :275-295 (norm / tensor_correlation / sample); the autograd backward of
//              the code side.  configs/train_config.yml:51 leaves feature_samples free; the reference's own value is 11, which the single-launch
//              kernel of corr_fused.hip serves (S * S <= 128).
//
// The same C ABI (stego_corr_fwd / stego_corr_bwd, include/stego_corr.h) dispatches here when S * S > 128: the caller sees the same outputs and
// the same saved tensors, only more launches.  Forward (8 launches):
//   sample_panels_kernel x 3  the sampled, L2-normalised FEATURES and CODES of all 2 + n_neg pair-sets (anchors: images [0, B); orig_*_pos at
//                             coords2: [B, 2 B); orig_*[perm_k] at coords2 through an index: the rest) written as the split-fp16 operand images
//                             of the dense-correspondence kernel - the fp32 rows of the sampled features never exist; the codes' normalised
//                             fp32 rows and 1 / |row| also go to the saved context (the backward needs them).  Both maps in one launch: the same
//                             points, the code map's taps ride in the feature row's round trip
//   dense_rowblock_kernel     fd[n] = anchors(n % B) . image(n)^T for all pair-sets in one launch, row sums on the way     -> saved_w (raw fd)
//   dense_rowpair_kernel      cd[n] (both row blocks of a pair in one workgroup: every B chunk streamed once), written straight into the three cd outputs
//   wide_set_mean_kernel      old_mean of every pair-set (modules.py:331) from the row sums -> saved_mean; the coordinates -> saved context
//   wide_pointwise_kernel     helper()'s elementwise part (:330-345) in place: loss (negative sets), the row sums of the loss of every set, and
//                             w = fd - rowmean + old_mean - shift with the clamp's pass mask in its mantissa LSB             -> saved_w
//   wide_loss_means_kernel    the three means the caller gets (:393-398)
// Backward (4 launches):
//   wide_code_tiles_kernel    the codes' side of the backward's GEMMs: per (image, 128-point block) the normalised rows transposed and split into
//                             fp16 hi | lo, in the layout the GEMM kernel keeps in LDS; 2 B more workgroups build the gather's lists (wide_build_lists)
//   wide_bwd_kernel           one workgroup per pair (set, image) walks its <= 2 x 2 tiles of G = -w * mask * upstream and takes both adjoints of
//                             the code correlation on the fp16 matrix cores (split-fp16 x 3 like the forward): d anchors(n) = G . rows(n),
//                             d rows(n) = G^T . anchors - G is staged once per tile in LDS (its w values loaded one tile ahead) and read along rows
//                             by four waves for the first product, along columns by four more for the second
//   wide_rows_kernel          the backward of norm() and of the bilinear sampling as a GATHER: the gradient of every un-normalised row ...
//   wide_gather_kernel        ... and one wave per pixel that sums the entries of its list (built beside the code tiles: a counting sort of the (point, tap)
//                             pairs per (gradient map, image) in LDS) in key order: no atomics, nothing to zero, bitwise repeatable (maps beyond 4096
//                             pixels fall back to the one-wave-per-point scatter with fp32 atomics, stego_sample_bwd_rows)
#include "corr_common.h"
#include "host_util.h"
#include "../../include/stego_corr.h"
#include "corr_wide.h"

namespace stego {

hipError_t launch_dense_corr_panels_seg(const void* imgA, const float* rsA, int imagesA, const void* imgB, const float* rsB, int B, int C, int M, int N,
                                        float* out, float* out1, float* out2, int seg, float* rowsum, hipStream_t stream);                     // dense_corr.hip
size_t dense_panel_image_bytes(int C, int P);
hipError_t launch_sample_panels2(const StegoMap* map, int C, void* panels, float* row_scale, float* rows_out, float* inv_out,
                                 const StegoMap* map2, int C2, void* panels2, float* row_scale2, float* rows_out2, float* inv_out2,
                                 const long long* index, int N, int H, int W, const float* coords, int n_coords, int S, int normalize,
                                 hipStream_t stream);                                                                                        // sample_sets.hip
hipError_t launch_sample_scatter(const float* g_rows, const float* rows_n, const float* inv, const StegoMap* d_map, const long long* index, int N, int C,
                                 int H, int W, const float* coords, int n_coords, int S, const float* extra, int n_extra, long long extra_stride,
                                 hipStream_t stream);

// ------------------------------------------------------------------------------------------------ forward: the elementwise part
struct WidePwParams {
    float* fd;                  // [n_img][P][P] raw fd in, w (mask in the LSB) out - in place
    const float* cd[3];         // the three cd outputs: images [0, B) | [B, 2 B) | [2 B, n_img)
    const float* rowsum;        // [n_img][P]
    float* mean;                // [n_sets] old_mean (zeros when not pointwise)
    float* neg_loss;            // [n_img - 2 B][P][P]
    float* lrowsum;             // [n_img][P] row sums of the loss
    float* loss_means;          // [3]
    float shift[3];
    float cmin, cmax;
    int n_sets, B, P, pointwise, keep_w;
    const float* co_src[2];     // coords1, coords2 [B][P][2] -> the saved context (blocks behind the pair-sets' in wide_set_mean_kernel)
    float* co_dst[2];
};

// one workgroup per pair-set, a fixed summation order; the workgroups behind them copy the bilinear coordinates into the saved context (the
// backward is handed the permutations only)
constexpr int WIDE_COPY_CHUNK = 4096;
__global__ void __launch_bounds__(1024) wide_set_mean_kernel(const WidePwParams p)
{
    __shared__ float red[16];
    const int s = blockIdx.x, tid = threadIdx.x;
    if (s >= p.n_sets) {
        const int n = p.B * p.P * 2, per = (n + WIDE_COPY_CHUNK - 1) / WIDE_COPY_CHUNK;
        const int c = s - p.n_sets, which = c / per, beg = (c - which * per) * WIDE_COPY_CHUNK;
        for (int i = beg + tid; i < min(beg + WIDE_COPY_CHUNK, n); i += 1024) p.co_dst[which][i] = p.co_src[which][i];
        return;
    }
    float acc = 0.f;
    if (p.pointwise) {
        const float* r = p.rowsum + (size_t)s * p.B * p.P;
#pragma unroll 8
        for (int i = tid; i < p.B * p.P; i += 1024) acc += r[i];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
    }
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += red[w];
        p.mean[s] = p.pointwise ? t / ((float)p.B * (float)p.P * (float)p.P) : 0.f;
    }
}

// one wave per row of one pair: modules.py:330-345
__global__ void __launch_bounds__(256) wide_pointwise_kernel(const WidePwParams p)
{
    const int lane = threadIdx.x & 63;
    const long long rows = (long long)p.n_sets * p.B * p.P;
    const long long r = __builtin_amdgcn_readfirstlane((int)((long long)blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (r >= rows) return;
    const int n = (int)(r / p.P);
    const int s = n / p.B;
    const float shift = p.shift[s < 2 ? s : 2];
    const float rm = p.pointwise ? p.rowsum[r] / (float)p.P : 0.f;
    const float om = p.pointwise ? p.mean[s] : 0.f;
    float* fd = p.fd + r * p.P;
    const float* cd = p.cd[s < 2 ? s : 2] + (r - (long long)(s < 2 ? s : 2) * p.B * p.P) * p.P;
    float* nl = s >= 2 ? p.neg_loss + (r - 2ll * p.B * p.P) * p.P : nullptr;
    float acc = 0.f;
    for (int j = lane; j < p.P; j += 64) {
        const float c = cd[j];
        const float wv = ((fd[j] - rm) + om) - shift;
        const float l = -fminf(fmaxf(c, p.cmin), p.cmax) * wv;
        if (nl) nl[j] = l;
        acc += l;
        if (p.keep_w) {
            // the backward's weight, the clamp's pass mask (inclusive, as torch's clamp backward) in the mantissa LSB
            const unsigned b = (__builtin_bit_cast(unsigned, wv) & ~1u) | ((c >= p.cmin && c <= p.cmax) ? 1u : 0u);
            fd[j] = __builtin_bit_cast(float, b);
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if (lane == 0) p.lrowsum[r] = acc;
}

// one workgroup per returned mean (modules.py:393-398), a fixed summation order
__global__ void __launch_bounds__(1024) wide_loss_means_kernel(const WidePwParams p)
{
    __shared__ float red[16];
    const int tid = threadIdx.x, q = blockIdx.x;
    const int per = p.B * p.P;
    const int beg = q * per, end = q < 2 ? (q + 1) * per : p.n_sets * per;
    float a = 0.f;
#pragma unroll 8
    for (int i = beg + tid; i < end; i += 1024) a += p.lrowsum[i];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m, 64);
    if ((tid & 63) == 0) red[tid >> 6] = a;
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += red[w];
        const float cnt = (float)p.B * (float)p.P * (float)p.P;
        p.loss_means[q] = q < 2 ? t / cnt : (p.n_sets > 2 ? t / (cnt * (float)(p.n_sets - 2)) : 0.f);
    }
}

// ------------------------------------------------------------------------------------------------ backward: both adjoints of the code correlation
constexpr int WB_GS = 132;      // floats per row of the G tile (528 B: conflict-free 16-byte reads along a row, column walks hit 64 banks)
constexpr int WB_TS = 136;      // halves per row of a transposed code tile [channel][128 points] (272 B: conflict-free 16-byte reads)
constexpr int WB_MAXNB = 2;     // point blocks per side: S * S <= 256
constexpr int WB_MAXKB = 3;     // 32-channel blocks of one pass: <= 88 channels (the tiles of both sides + G in 160 KB of LDS); K <= 128 in two passes

struct WideBwdParams {
    const float* w;             // saved_w [n_img][P][P]
    const float* cn;            // [n_img][P][K] normalised sampled codes
    const unsigned char* tiles; // [n_img][nb] code tiles, transposed and split: hi [Kr][WB_TS] | lo [Kr][WB_TS] fp16 of CSCALE * cn, tile_bytes each
    unsigned char* tiles_out;   // (wide_code_tiles_kernel)
    int tile_bytes;
    float* zero[2];             // (wide_code_tiles_kernel: the two code gradients, zeroed by the workgroups behind the tiles')
    long long zero_floats;
    const float* g_intra;       // device scalars: upstreams of loss_means[0 .. 1] (null: 0)
    const float* g_inter;
    const float* g_neg;         // upstream of the negative losses, see g_neg_stride (null: 0)
    const float* g_cd[3];       // optional dense upstreams of the three cd outputs
    float* d_rows;              // [n_img][P][K] gradient of image n's rows as the second operand of pair n
    float* d_anchor;            // [n_img][P][K] gradient of the anchors (image n % B) from pair n
    int g_neg_stride;           // 1 dense [n_neg B][P][P], 0 one scalar per element, -1 one scalar = the upstream of loss_means[2]
    int B, P, K, Kr, n_sets;
    int k0, Kc;                 // the channel window [k0, k0 + Kc) of this pass (Kc <= 88: both code tiles + G in 160 KB of LDS; K > 88 takes two passes); Kr = Kc rounded up to 8
};

__device__ __forceinline__ void split8(const float (&v)[8], float scale, f16x8& hi, f16x8& lo)
{
    unsigned h[4], l[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) split_f16_pair(v[2 * q] * scale, v[2 * q + 1] * scale, h[q], l[q]);
    typedef unsigned int du32x4 __attribute__((ext_vector_type(4)));
    hi = __builtin_bit_cast(f16x8, du32x4{h[0], h[1], h[2], h[3]});
    lo = __builtin_bit_cast(f16x8, du32x4{l[0], l[1], l[2], l[3]});
}

constexpr float WB_CSCALE = 16.f;     // the normalised codes (|x| <= 1) times 16: their lo parts leave the fp16 subnormals

__device__ __forceinline__ void wide_dma_piece(const unsigned char* gsrc_lane, unsigned lds_addr)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc_lane), "s"(lds_addr) : "memory");
}

struct WideGatherParams {
    const float* d_rows;        // [n_img][P][K]
    const float* d_anchor;      // [n_img][P][K]
    const float* cn;            // [n_img][P][K]
    const float* inv;           // [n_img][P]
    const float* co1;           // [B][P][2]
    const float* co2;
    const long long* perms;     // [n_neg * B]
    float* V;                   // [n_img][P][K]
    int* off;                   // [2][B][HW + 1] absolute entry indices
    unsigned* ent;              // pool of {key, weight bits}
    float* d_map[2];            // d_code, d_code_pos: channels-last dense [B][H][W][K]
    int B, P, S, K, H, W, n_sets, n_neg;
};

// the (point, tap) pairs that land in every pixel of image m of gradient map `which`, sorted by pixel: a counting sort in LDS by the calling workgroup
// (T threads).  See "backward: norm() and the sampling, gathered" below.
template <int T>
__device__ __forceinline__ void wide_build_lists(const WideGatherParams& p, int which, int m, unsigned char* smem)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int P = p.P, HW = p.H * p.W;
    constexpr int WG_THREADS = T;
    int* cnt = reinterpret_cast<int*>(smem);            // [HW]
    int* off = cnt + HW;                                // [HW + 1]
    int* srcs = off + HW + 1;                           // [n_neg * B]
    int* misc = srcs + p.n_neg * p.B;                   // [0] sources, [1] negatives of lower images, [2 ..] wave totals of the scan
    for (int i = tid; i < HW; i += WG_THREADS) cnt[i] = 0;
    if (tid < 2) misc[tid] = 0;
    __syncthreads();
    if (which == 0) {
        for (int t = tid; t < p.n_neg * p.B; t += WG_THREADS) {
            const long long pm = p.perms[t];
            if (pm == (long long)m) srcs[atomicAdd(&misc[0], 1)] = t;
            else if (pm < (long long)m) atomicAdd(&misc[1], 1);
        }
        __syncthreads();
    }
    const int n_src = which == 0 ? 1 + misc[0] : 1;
    const int items = n_src * P;
    const long long base = which == 0 ? 4ll * P * (m + misc[1]) : 4ll * P * ((long long)p.B * (1 + p.n_neg) + m);
    auto taps_of = [&](int it, int& rowi, int4& yx, float4& tw) {
        const int src = it / P, q = it - src * P;
        int slot, crow;
        const float* co;
        if (which == 1) { slot = p.B + m; crow = m; co = p.co2; }
        else if (src == 0) { slot = m; crow = m; co = p.co1; }
        else { const int t = srcs[src - 1]; slot = 2 * p.B + t; crow = t % p.B; co = p.co2; }
        const int hh = q / p.S, ww = q - hh * p.S;                      // point (h, w) reads coords[w][h] (the permute of modules.py:288)
        const float* c = co + ((size_t)crow * P + (size_t)ww * p.S + hh) * 2;
        make_taps(c[0], c[1], p.H, p.W, yx, tw);
        rowi = slot * P + q;
    };
    // counts per pixel (taps of weight zero - beyond the border - are not entries)
    for (int it = tid; it < items; it += WG_THREADS) {
        int rowi;
        int4 yx;
        float4 tw;
        taps_of(it, rowi, yx, tw);
        const int px[4] = {yx.x, yx.y, yx.z, yx.w};
        const float wt[4] = {tw.x, tw.y, tw.z, tw.w};
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (wt[t] != 0.f) atomicAdd(&cnt[(px[t] >> 16) * p.W + (px[t] & 0xffff)], 1);
    }
    __syncthreads();
    // exclusive prefix over the pixels: E consecutive pixels per thread, wave scan, wave totals
    const int E = (HW + WG_THREADS - 1) / WG_THREADS;
    int loc = 0;
    for (int e = 0; e < E; ++e) { const int i = tid * E + e; if (i < HW) loc += cnt[i]; }
    int inc = loc;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(inc, d, 64); if (lane >= d) inc += v; }
    if (lane == 63) misc[2 + (tid >> 6)] = inc;
    __syncthreads();
    int wbase = 0;
    for (int w2 = 0; w2 < (tid >> 6); ++w2) wbase += misc[2 + w2];
    int run = wbase + inc - loc;
    for (int e = 0; e < E; ++e) { const int i = tid * E + e; if (i < HW) { off[i] = run; run += cnt[i]; } }
    if (tid == WG_THREADS - 1) off[HW] = run;
    __syncthreads();
    int* goff = p.off + (size_t)(which * p.B + m) * (HW + 1);
    for (int i = tid; i <= HW; i += WG_THREADS) goff[i] = (int)base + off[i];
    for (int i = tid; i < HW; i += WG_THREADS) cnt[i] = 0;
    __syncthreads();
    // fill
    for (int it = tid; it < items; it += WG_THREADS) {
        int rowi;
        int4 yx;
        float4 tw;
        taps_of(it, rowi, yx, tw);
        const int px[4] = {yx.x, yx.y, yx.z, yx.w};
        const float wt[4] = {tw.x, tw.y, tw.z, tw.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (wt[t] != 0.f) {
                const int pix = (px[t] >> 16) * p.W + (px[t] & 0xffff);
                const long long pos = base + off[pix] + atomicAdd(&cnt[pix], 1);
                p.ent[2 * pos] = (unsigned)(rowi * 4 + t);
                p.ent[2 * pos + 1] = __builtin_bit_cast(unsigned, wt[t]);
            }
        }
    }
}

// The operands of the backward's GEMMs that come from the codes, once per backward: tile (image, 128-point block) = the block's normalised rows
// TRANSPOSED ([channel][point]: a fragment of either product is then 8 consecutive points of one channel = one 16-byte LDS read) and split
// into fp16 hi | lo, in the layout the GEMM kernel keeps in LDS - staging a tile there is a linear LDS-DMA copy.
constexpr int WB_ZERO_CHUNK = 16384;      // floats per zeroing workgroup
__global__ void __launch_bounds__(512) wide_code_tiles_kernel(const WideBwdParams p, const WideGatherParams q, const int list_blocks)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef unsigned int du32x4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x;
    const int nbp = (p.P + TP - 1) / TP;
    const int n_tiles = p.n_sets * p.B * nbp;
    if ((int)blockIdx.x >= n_tiles && (int)blockIdx.x < n_tiles + list_blocks) {
        // the lists of the pixel gather (they depend on the coordinates and permutations only: built here, beside the tiles, in front of the GEMMs)
        const int lb = blockIdx.x - n_tiles;
        wide_build_lists<512>(q, lb / q.B, lb % q.B, smem);
        return;
    }
    if ((int)blockIdx.x >= n_tiles) {
        // (maps beyond 4096 pixels) the scatter launches ADD into the code gradients: zero them here
        const long long per = (p.zero_floats + WB_ZERO_CHUNK - 1) / WB_ZERO_CHUNK;
        const long long c = (long long)blockIdx.x - n_tiles - list_blocks;
        const int which = (int)(c / per);
        const long long beg = (c - which * per) * WB_ZERO_CHUNK, end = min(beg + (long long)WB_ZERO_CHUNK, p.zero_floats);
        float* z = p.zero[which];
        for (long long i = beg + 4 * tid; i < end; i += 4 * 512) {
            if (i + 3 < end && (reinterpret_cast<uintptr_t>(z) & 15) == 0) *reinterpret_cast<f32x4*>(z + i) = f32x4{0.f, 0.f, 0.f, 0.f};
            else for (long long q = i; q < end; ++q) z[q] = 0.f;
        }
        return;
    }
    const int img = blockIdx.x / nbp, blk = blockIdx.x - img * nbp;
    const int P = p.P, K = p.K, Kc = p.Kc, Kr = p.Kr, p0 = blk * TP;
    half_t* T = reinterpret_cast<half_t*>(smem);
    for (int i = tid; i < p.tile_bytes / 16; i += 512) reinterpret_cast<du32x4*>(smem)[i] = du32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    const float* src = p.cn + ((size_t)img * P + p0) * K + p.k0;       // channels [k0, k0 + Kc) of the block's rows
    const int rows = min(TP, P - p0);
    constexpr int UN = 22;                                              // 128 x 88 / 512: every load of a thread in flight at once
    float v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
        const int idx = u * 512 + tid;
        const int jl = idx / Kc, k = idx - jl * Kc;
        v[u] = idx < rows * Kc ? src[(size_t)jl * K + k] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
        const int idx = u * 512 + tid;
        if (idx < rows * Kc) {
            const int jl = idx / Kc, k = idx - jl * Kc;
            unsigned hh, ll;
            split_f16_pair(v[u] * WB_CSCALE, 0.f, hh, ll);
            reinterpret_cast<unsigned short*>(T)[k * WB_TS + jl] = (unsigned short)(hh & 0xffffu);
            reinterpret_cast<unsigned short*>(T + Kr * WB_TS)[k * WB_TS + jl] = (unsigned short)(ll & 0xffffu);
        }
    }
    __syncthreads();
    du32x4* dst = reinterpret_cast<du32x4*>(p.tiles_out + (size_t)blockIdx.x * p.tile_bytes);
    for (int i = tid; i < p.tile_bytes / 16; i += 512) dst[i] = reinterpret_cast<const du32x4*>(smem)[i];
}

constexpr int WB_THREADS = 512;       // 8 waves: waves 0-3 take the anchors' product, waves 4-7 the second operand's, two per SIMD

__global__ void __launch_bounds__(WB_THREADS) wide_bwd_kernel(const WideBwdParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* Gs = reinterpret_cast<float*>(smem);                                       // [128][WB_GS]
    half_t* Ta = reinterpret_cast<half_t*>(smem + 128 * WB_GS * 4);                   // anchors' tile: hi [Kr][WB_TS] | lo
    half_t* Tb = reinterpret_cast<half_t*>(smem + 128 * WB_GS * 4 + p.tile_bytes);   // the second operand's tile
    float* red = Gs;                                                                  // (the first pass over G below: before any tile is staged)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave >> 2, wq = wave & 3;                                        // 0: d anchors, 1: d rows of the second operand
    const int r = lane & 31, h = lane >> 5;
    const int n = blockIdx.x, P = p.P, K = p.K, Kr = p.Kr;
    const int s = n / p.B, na = n - s * p.B;
    const int nbp = (P + TP - 1) / TP, NK = (Kr + 31) >> 5;
    const float cnt = (float)p.B * (float)P * (float)P;
    // the scalar part of the upstream of this pair's loss elements
    float up = 0.f;
    if (s == 0) up = p.g_intra ? p.g_intra[0] / cnt : 0.f;
    else if (s == 1) up = p.g_inter ? p.g_inter[0] / cnt : 0.f;
    else if (p.g_neg && p.g_neg_stride <= 0) up = p.g_neg_stride == 0 ? p.g_neg[0] : p.g_neg[0] / (cnt * (float)(p.n_sets - 2));
    const float* gneg = (s >= 2 && p.g_neg && p.g_neg_stride == 1) ? p.g_neg + (size_t)(n - 2 * p.B) * P * P : nullptr;
    const float* gcd = p.g_cd[s < 2 ? s : 2] ? p.g_cd[s < 2 ? s : 2] + (size_t)(n - (s < 2 ? s : 2) * p.B) * P * P : nullptr;
    const float* wn = p.w + (size_t)n * P * P;
    auto g_of = [&](int i, int j, float wv) {
        float u = up;
        if (gneg) u += gneg[(size_t)i * P + j];
        float g = (__builtin_bit_cast(unsigned, wv) & 1u) ? -(wv * u) : 0.f;
        if (gcd) g += gcd[(size_t)i * P + j];
        return g;
    };
    // one power-of-two scale for G so that its fp16 hi / lo parts stay in the normal range: |w| < 8 bounds the scalar case; dense upstreams take a
    // pass over the pair's G first
    float gmax = fabsf(up) * 8.f;
    if (gneg || gcd) {
        float m = 0.f;
        for (int idx = tid; idx < P * P; idx += WB_THREADS) {
            const int i = idx / P, j = idx - i * P;
            m = fmaxf(m, fabsf(g_of(i, j, wn[idx])));
        }
#pragma unroll
        for (int q = 32; q >= 1; q >>= 1) m = fmaxf(m, __shfl_xor(m, q, 64));
        if (lane == 0) red[wave] = m;
        __syncthreads();
        gmax = 0.f;
        for (int q = 0; q < WB_THREADS / 64; ++q) gmax = fmaxf(gmax, red[q]);
    }
    const float gscale = gmax > 0.f ? __builtin_ldexpf(1.f, -__builtin_amdgcn_frexp_expf(gmax)) : 1.f;
    const float unscale = 1.f / (gscale * WB_CSCALE);

    // a prepared code tile into LDS: linear LDS-DMA, 1 KB per wave instruction (asynchronous: waited for with the G tile's barrier)
    const unsigned smem_addr = (unsigned)(uintptr_t)((__attribute__((address_space(3))) const unsigned char*)smem);
    auto stage_codes = [&](const half_t* T, int img, int blk) {
        const unsigned char* src = p.tiles + ((size_t)img * nbp + blk) * p.tile_bytes + lane * 16;
        const unsigned dst = smem_addr + (unsigned)(reinterpret_cast<const unsigned char*>(T) - smem);
        for (int pc = wave; pc < p.tile_bytes / 1024; pc += WB_THREADS / 64) wide_dma_piece(src + pc * 1024, dst + pc * 1024);
    };
    // the w values of tile (mi, nj) this thread turns into G: rows (tid >> 5) + 16 it, columns 4 (tid & 31) .. + 3.  Loaded one tile AHEAD, while
    // the matrix cores work on the current one
    const int jl = 4 * (tid & 31);
    f32x4 wv[8];
    auto fetch = [&](int mi, int nj) {
        const int i0 = mi * TP, j0 = nj * TP;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int il = (tid >> 5) + 16 * it;
            wv[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (i0 + il < P) {
                const float* wr = wn + (size_t)(i0 + il) * P + j0 + jl;
                if ((P & 3) == 0) {
                    if (j0 + jl < P) wv[it] = *reinterpret_cast<const f32x4*>(wr);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (j0 + jl + q < P) wv[it][q] = wr[q];
                }
            }
        }
    };

    // accumulators: role 0 - acc[0][kb] = the anchors' rows of block mi (reset per mi); role 1 - acc[nj][kb] = the second operand's rows of block nj
    f32x16 acc[WB_MAXNB][WB_MAXKB];
#pragma unroll
    for (int a = 0; a < WB_MAXNB; ++a)
#pragma unroll
        for (int b = 0; b < WB_MAXKB; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    fetch(0, 0);
    for (int mi = 0; mi < nbp; ++mi) {
#pragma unroll
        for (int nj = 0; nj < WB_MAXNB; ++nj) {
            if (nj < nbp) {
                __syncthreads();                                        // the previous tile's fragments are read
                if (nj == 0) stage_codes(Ta, na, mi);
                stage_codes(Tb, n, nj);
                // G tile (mi, nj): rows i0 .. + 127 of the anchors, columns j0 .. + 127 of the second operand
                const int i0 = mi * TP, j0 = nj * TP;
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int il = (tid >> 5) + 16 * it;
                    f32x4 g4 = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (i0 + il < P) {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (j0 + jl + q < P) g4[q] = g_of(i0 + il, j0 + jl + q, wv[it][q]);
                    }
                    *reinterpret_cast<f32x4*>(Gs + il * WB_GS + jl) = g4;
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the code tiles' copies
                __syncthreads();
                {                                                         // the next tile's w: in flight during the products below
                    const int nj2 = nj + 1 < nbp ? nj + 1 : 0, mi2 = nj + 1 < nbp ? mi : mi + 1;
                    if (mi2 < nbp) fetch(mi2, nj2);
                }
                if (role == 0) {
                    // ---- d anchors: rows i (32 per wave) x channels, contraction over the tile's columns j
#pragma unroll 1
                    for (int ks = 0; ks < TP / 16; ++ks) {
                        float gv[8];
                        const float* gp = Gs + (32 * wq + r) * WB_GS + 16 * ks + 8 * h;
                        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp), g1 = *reinterpret_cast<const f32x4*>(gp + 4);
#pragma unroll
                        for (int q = 0; q < 4; ++q) { gv[q] = g0[q]; gv[4 + q] = g1[q]; }
                        f16x8 ah, al;
                        split8(gv, gscale, ah, al);
#pragma unroll
                        for (int kb = 0; kb < WB_MAXKB; ++kb) {
                            if (kb < NK) {
                                const bool kv = 32 * kb + r < Kr;
                                const half_t* tp = Tb + (size_t)(kv ? 32 * kb + r : 0) * WB_TS + 16 * ks + 8 * h;
                                f16x8 bh = *reinterpret_cast<const f16x8*>(tp), bl = *reinterpret_cast<const f16x8*>(tp + Kr * WB_TS);
                                if (!kv) { bh = f16x8{}; bl = f16x8{}; }
                                acc[0][kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[0][kb], 0, 0, 0);
                                acc[0][kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[0][kb], 0, 0, 0);
                                acc[0][kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[0][kb], 0, 0, 0);
                            }
                        }
                    }
                } else {
                    // ---- d rows of the second operand: rows j (32 per wave) x channels, contraction over the tile's rows i: G read along columns
#pragma unroll 1
                    for (int ks = 0; ks < TP / 16; ++ks) {
                        float gv[8];
                        const float* gp = Gs + (16 * ks + 8 * h) * WB_GS + 32 * wq + r;
#pragma unroll
                        for (int e = 0; e < 8; ++e) gv[e] = gp[e * WB_GS];
                        f16x8 ah, al;
                        split8(gv, gscale, ah, al);
#pragma unroll
                        for (int kb = 0; kb < WB_MAXKB; ++kb) {
                            if (kb < NK) {
                                const bool kv = 32 * kb + r < Kr;
                                const half_t* tp = Ta + (size_t)(kv ? 32 * kb + r : 0) * WB_TS + 16 * ks + 8 * h;
                                f16x8 bh = *reinterpret_cast<const f16x8*>(tp), bl = *reinterpret_cast<const f16x8*>(tp + Kr * WB_TS);
                                if (!kv) { bh = f16x8{}; bl = f16x8{}; }
                                acc[nj][kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[nj][kb], 0, 0, 0);
                                acc[nj][kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[nj][kb], 0, 0, 0);
                                acc[nj][kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[nj][kb], 0, 0, 0);
                            }
                        }
                    }
                }
            }
        }
        if (role == 0) {
            // the anchors' rows of block mi are complete.  C/D layout: column (channel) = lane & 31 (+ 32 kb), row = (e & 3) + 8 (e >> 2) + 4 h
            float* da = p.d_anchor + (size_t)n * P * K + p.k0;
#pragma unroll
            for (int kb = 0; kb < WB_MAXKB; ++kb) {
                const int k = 32 * kb + r;
                if (kb < NK && k < p.Kc) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int i = mi * TP + 32 * wq + (e & 3) + 8 * (e >> 2) + 4 * h;
                        if (i < P) da[(size_t)i * K + k] = acc[0][kb][e] * unscale;
                    }
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[0][kb][e] = 0.f;
            }
        }
    }
    if (role == 1) {
        float* dr = p.d_rows + (size_t)n * P * K + p.k0;
#pragma unroll
        for (int nj = 0; nj < WB_MAXNB; ++nj) {
            if (nj < nbp) {
#pragma unroll
                for (int kb = 0; kb < WB_MAXKB; ++kb) {
                    const int k = 32 * kb + r;
                    if (kb < NK && k < p.Kc) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int j = nj * TP + 32 * wq + (e & 3) + 8 * (e >> 2) + 4 * h;
                            if (j < P) dr[(size_t)j * K + k] = acc[nj][kb][e] * unscale;
                        }
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward: norm() and the sampling, gathered
// The adjoint of norm() (F.normalize, modules.py:275-276) and of the bilinear sampling (:287-288, incl. the orig_code[perm] gather of :385) as a
// GATHER per destination pixel - no atomics, nothing to zero, a fixed summation order:
//   wide_build_lists         (workgroups of the code-tile launch) one workgroup per (gradient map, image): the (point, tap) pairs that land in every pixel
//                            of the image - its own anchors and the negative sets that drew it (a scan of the permutations) for orig_code, the
//                            positive set for orig_code_pos - as a counting sort by pixel in LDS: counts, prefix, fill; entry = {4 (row) + tap,
//                            weight}; the image's slice of the pool starts at 4 P (m + #{negatives of lower images}): no allocation
//   wide_rows_kernel         one wave per sampled row: V = the gradient of the UN-normalised row (norm backward applied, the anchors'
//                            first-operand gradients of all pair-sets summed in)
//   wide_gather_kernel       one wave per pixel: its <= 64 entries sorted by key (so that the sum has one order whatever order the fill took),
//                            then out[pixel][:] = sum w * V[row][:], K floats per entry in one or two coalesced loads.  A pixel with more than 64
//                            taps is summed chunk by chunk (order of the chunks as filled)
// Against the one-wave-per-point scatter with global fp32 atomics (88 us for three launches at S = 16, 52 us of them the atomics).
__global__ void __launch_bounds__(256) wide_rows_kernel(const WideGatherParams p)
{
    // V rows, one wave per (image slot, point): the gradient of the UN-normalised sampled row
    const int tid = threadIdx.x, lane = tid & 63;
    const int P = p.P, K = p.K;
    const long long row = (long long)blockIdx.x * 4 + (tid >> 6);
    if (row >= (long long)p.n_sets * p.B * P) return;
    const size_t o = (size_t)row * K;
    const bool anchor = row < (long long)p.B * P;
    const size_t set_stride = (size_t)p.B * P * K;
    float g[2], y[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int c = lane + 64 * e;
        g[e] = 0.f;
        y[e] = 0.f;
        if (c < K) {
            float gg = p.d_rows[o + c];
            if (anchor)
                for (int s2 = 0; s2 < p.n_sets; ++s2) gg += p.d_anchor[(size_t)s2 * set_stride + o + c];
            g[e] = gg;
            y[e] = p.cn[o + c];
        }
    }
    // norm() backward: y = x / max(|x|, eps) -> dx = (g - y <y, g>) / |x|, or g / eps below eps
    float proj = __builtin_fmaf(y[0], g[0], y[1] * g[1]);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) proj += __shfl_xor(proj, m, 64);
    const float iv = p.inv[row];
    if (iv > 0.99e10f) proj = 0.f;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int c = lane + 64 * e;
        if (c < K) p.V[o + c] = iv * (g[e] - y[e] * proj);
    }
}

__global__ void __launch_bounds__(256) wide_gather_kernel(const WideGatherParams p)
{
    const int lane = threadIdx.x & 63;
    const int HW = p.H * p.W, K = p.K;
    const long long gpix = __builtin_amdgcn_readfirstlane((int)((long long)blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (gpix >= 2ll * p.B * HW) return;
    const int wm = (int)(gpix / HW), pix = (int)(gpix - (long long)wm * HW);          // wm = which * B + m
    const int* goff = p.off + (size_t)wm * (HW + 1);
    const int beg = goff[pix], end = goff[pix + 1];
    float a0 = 0.f, a1 = 0.f;
    for (int b0 = beg; b0 < end; b0 += 64) {
        const int n = min(64, end - b0);
        unsigned key = 0xffffffffu;
        float w = 0.f;
        if (lane < n) { key = p.ent[2 * (size_t)(b0 + lane)]; w = __builtin_bit_cast(float, p.ent[2 * (size_t)(b0 + lane) + 1]); }
        // sort the chunk by key: rank = entries with a smaller key (keys are distinct), then every lane sends its entry to lane `rank`
        int rank = 0;
        for (int i = 0; i < n; ++i) rank += (unsigned)__builtin_amdgcn_readlane((int)key, i) < key ? 1 : 0;
        const unsigned skey = (unsigned)__builtin_amdgcn_ds_permute(rank << 2, (int)key);
        const float sw = __builtin_bit_cast(float, __builtin_amdgcn_ds_permute(rank << 2, __builtin_bit_cast(int, w)));
        int i = 0;
        for (; i + 4 <= n; i += 4) {
            float v0[4], v1[4], wi[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned k2 = (unsigned)__builtin_amdgcn_readlane((int)skey, i + u);
                wi[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sw), i + u));
                const float* vr = p.V + (size_t)(k2 >> 2) * K;
                v0[u] = lane < K ? vr[lane] : 0.f;             // (K < 64: the row ends before lane 63)
                v1[u] = lane + 64 < K ? vr[lane + 64] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { a0 = __builtin_fmaf(wi[u], v0[u], a0); a1 = __builtin_fmaf(wi[u], v1[u], a1); }
        }
        for (; i < n; ++i) {
            const unsigned k2 = (unsigned)__builtin_amdgcn_readlane((int)skey, i);
            const float wi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sw), i));
            const float* vr = p.V + (size_t)(k2 >> 2) * K;
            a0 = __builtin_fmaf(wi, lane < K ? vr[lane] : 0.f, a0);
            if (lane + 64 < K) a1 = __builtin_fmaf(wi, vr[lane + 64], a1);
        }
    }
    const int which = wm / p.B, m = wm - which * p.B;
    float* o = p.d_map[which] + ((size_t)m * HW + pix) * K;
    if (lane < K) o[lane] = a0;
    if (lane + 64 < K) o[lane + 64] = a1;
}

// ------------------------------------------------------------------------------------------------ host side
bool wide_supported(int B, int C, int K, int S, int n_neg)
{
    const int P = S * S;
    return P > TP && P <= WB_MAXNB * TP && K <= 128 && B >= 1 && C >= 1 && n_neg >= 0 &&
           (long long)(2 + n_neg) * B <= 65535 && (long long)(2 + n_neg) * B * P < (1ll << 31) / 4;
}

static size_t up256(size_t v) { return (v + 255) / 256 * 256; }

WideGeom wide_geometry(int B, int C, int K, int H, int W, int S, int n_neg)
{
    WideGeom g;
    g.n_sets = 2 + n_neg;
    g.n_img = g.n_sets * B;
    g.P = S * S;
    g.nb = (g.P + TP - 1) / TP;
    // the backward's GEMM kernel takes the code channels in windows of <= 88 (its LDS): one pass up to K = 88, two beyond
    g.nwin = (K + 87) / 88;
    g.Kc = (((K + g.nwin - 1) / g.nwin) + 7) & ~7;
    g.Kr = g.Kc;
    g.fimg = dense_panel_image_bytes(C, g.P);
    g.cimg = dense_panel_image_bytes(K, g.P);
    const size_t rows = (size_t)g.n_img * g.P;
    g.c_cn = 0;
    g.c_inv = up256(rows * K * 4);
    g.c_co1 = g.c_inv + up256(rows * 4);
    g.c_co2 = g.c_co1 + up256((size_t)B * g.P * 8);
    g.ctx_bytes = g.c_co2 + up256((size_t)B * g.P * 8);
    size_t o = 0;
    g.o_fpan = o; o += up256((size_t)g.n_img * g.fimg);
    g.o_frs = o; o += up256((size_t)g.n_img * g.nb * TP * 4);
    g.o_cpan = o; o += up256((size_t)g.n_img * g.cimg);
    g.o_crs = o; o += up256((size_t)g.n_img * g.nb * TP * 4);
    g.o_rowsum = o; o += up256(rows * 4);
    g.o_lrowsum = o; o += up256(rows * 4);
    g.o_mean = o; o += 256;
    g.o_fd = o; o += up256(rows * g.P * 4);                 // fd when the caller keeps nothing for a backward
    g.o_ctx = o; o += g.ctx_bytes;                          // the context likewise
    g.ws_bytes = o + 256;
    g.tile_bytes = (2 * g.Kr * WB_TS * 2 + 1023) / 1024 * 1024;
    g.b_rows = 0;
    g.b_anchor = up256(rows * K * 4);
    g.b_tiles = g.b_anchor + up256(rows * K * 4);
    g.b_v = g.b_tiles + up256((size_t)g.n_img * g.nb * g.tile_bytes);
    g.b_ent = g.b_v + up256(rows * K * 4);                              // {key, weight} per (point, tap)
    g.b_off = g.b_ent + up256(rows * 4 * 8);
    g.bwd_ws_bytes = g.b_off + up256((size_t)2 * B * ((size_t)H * W + 1) * 4) + 256;
    return g;
}

hipError_t launch_wide_fwd(const WideFwdArgs& a, hipStream_t stream)
{
    const WideGeom g = wide_geometry(a.B, a.C, a.K, a.H, a.W, a.S, a.n_neg);
    unsigned char* ws = static_cast<unsigned char*>(a.workspace);
    ws += (256 - (reinterpret_cast<uintptr_t>(ws) & 255)) & 255;
    unsigned char* ctx = a.saved_ctx ? static_cast<unsigned char*>(a.saved_ctx) : ws + g.o_ctx;
    float* fd = a.saved_w ? a.saved_w : reinterpret_cast<float*>(ws + g.o_fd);
    float* mean = a.saved_mean ? a.saved_mean : reinterpret_cast<float*>(ws + g.o_mean);
    float* cn = reinterpret_cast<float*>(ctx + g.c_cn);
    float* inv = reinterpret_cast<float*>(ctx + g.c_inv);
    float* rowsum = reinterpret_cast<float*>(ws + g.o_rowsum);
    const int B = a.B, P = g.P, nnb = a.n_neg * B;
    hipError_t e;
    struct Src { const StegoMap* f; const StegoMap* c; const float* co; const long long* idx; int first, n; };
    const Src src[3] = {{a.feats, a.code, a.coords1, nullptr, 0, B}, {a.feats_pos, a.code_pos, a.coords2, nullptr, B, B},
                        {a.feats, a.code, a.coords2, a.perms, 2 * B, nnb}};
    float* frs = reinterpret_cast<float*>(ws + g.o_frs);
    float* crs = reinterpret_cast<float*>(ws + g.o_crs);
    // features and codes of a source in ONE launch: the same points, the code map's taps ride in the feature row's round trip
    for (const Src& q : src) {
        if (q.n == 0) continue;
        if ((e = launch_sample_panels2(q.f, a.C, ws + g.o_fpan + (size_t)q.first * g.fimg, frs + (size_t)q.first * g.nb * TP, nullptr, nullptr,
                                       q.c, a.K, ws + g.o_cpan + (size_t)q.first * g.cimg, crs + (size_t)q.first * g.nb * TP,
                                       cn + (size_t)q.first * P * a.K, inv + (size_t)q.first * P, q.idx, q.n, a.H, a.W, q.co, B, a.S, 1, stream)) != hipSuccess) return e;
    }
    if ((e = launch_dense_corr_panels_seg(ws + g.o_fpan, frs, B, ws + g.o_fpan, frs, g.n_img, a.C, P, P, fd, nullptr, nullptr, 0,
                                          a.pointwise ? rowsum : nullptr, stream)) != hipSuccess) return e;
    if ((e = launch_dense_corr_panels_seg(ws + g.o_cpan, crs, B, ws + g.o_cpan, crs, g.n_img, a.K, P, P, a.intra_cd, a.inter_cd, a.neg_cd, B,
                                          nullptr, stream)) != hipSuccess) return e;
    WidePwParams p{};
    p.fd = fd; p.cd[0] = a.intra_cd; p.cd[1] = a.inter_cd; p.cd[2] = a.neg_cd;
    p.rowsum = rowsum; p.mean = mean; p.neg_loss = a.neg_loss;
    p.lrowsum = reinterpret_cast<float*>(ws + g.o_lrowsum);
    p.loss_means = a.loss_means;
    p.shift[0] = a.shift[0]; p.shift[1] = a.shift[1]; p.shift[2] = a.shift[2];
    p.cmin = a.cmin; p.cmax = a.cmax;
    p.n_sets = g.n_sets; p.B = B; p.P = P; p.pointwise = a.pointwise; p.keep_w = a.saved_w ? 1 : 0;
    p.co_src[0] = a.coords1; p.co_src[1] = a.coords2;
    p.co_dst[0] = reinterpret_cast<float*>(ctx + g.c_co1); p.co_dst[1] = reinterpret_cast<float*>(ctx + g.c_co2);
    const int copy_blocks = 2 * ((B * P * 2 + WIDE_COPY_CHUNK - 1) / WIDE_COPY_CHUNK);
    hipLaunchKernelGGL(wide_set_mean_kernel, dim3(g.n_sets + copy_blocks), dim3(1024), 0, stream, p);
    const long long rows = (long long)g.n_img * P;
    hipLaunchKernelGGL(wide_pointwise_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, p);
    hipLaunchKernelGGL(wide_loss_means_kernel, dim3(3), dim3(1024), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_wide_bwd(const WideBwdArgs& a, hipStream_t stream)
{
    const WideGeom g = wide_geometry(a.B, a.C, a.K, a.H, a.W, a.S, a.n_neg);
    unsigned char* ws = static_cast<unsigned char*>(a.workspace);
    ws += (256 - (reinterpret_cast<uintptr_t>(ws) & 255)) & 255;
    const unsigned char* ctx = static_cast<const unsigned char*>(a.saved_ctx);
    const float* cn = reinterpret_cast<const float*>(ctx + g.c_cn);
    const float* inv = reinterpret_cast<const float*>(ctx + g.c_inv);
    const float* co1 = reinterpret_cast<const float*>(ctx + g.c_co1);
    const float* co2 = reinterpret_cast<const float*>(ctx + g.c_co2);
    const int B = a.B, P = g.P, K = a.K, nnb = a.n_neg * B;
    hipError_t e;
    WideBwdParams p{};
    p.w = a.saved_w; p.cn = cn;
    p.g_intra = a.g_intra; p.g_inter = a.g_inter; p.g_neg = a.g_neg; p.g_neg_stride = a.g_neg_stride;
    p.g_cd[0] = a.g_intra_cd; p.g_cd[1] = a.g_inter_cd; p.g_cd[2] = a.g_neg_cd;
    p.d_rows = reinterpret_cast<float*>(ws + g.b_rows);
    p.d_anchor = reinterpret_cast<float*>(ws + g.b_anchor);
    p.B = B; p.P = P; p.K = K; p.Kr = g.Kr; p.n_sets = g.n_sets;
    p.tiles_out = ws + g.b_tiles;
    p.tiles = p.tiles_out;
    p.tile_bytes = g.tile_bytes;
    const int HW = a.H * a.W;
    // (debug bit 20: the scatter with global atomics.  ADVICE round 5: the list builder keeps 2 HW + n_neg B + ... words in LDS - beyond the
    // 160 KB a shape that the forward took must not fail here: it takes the scatter path)
    const bool gather = HW <= 4096 && (size_t)(2 * HW + 1 + nnb + 2 + 512 / 64 + 4) * 4 <= (size_t)160 * 1024 - 1024 &&
                        !(knob(KNOB_DEBUG_BWD) & (1 << 20));
    WideGatherParams q{};
    q.d_rows = p.d_rows; q.d_anchor = p.d_anchor; q.cn = cn; q.inv = inv; q.co1 = co1; q.co2 = co2; q.perms = a.perms;
    q.V = reinterpret_cast<float*>(ws + g.b_v);
    q.off = reinterpret_cast<int*>(ws + g.b_off);
    q.ent = reinterpret_cast<unsigned*>(ws + g.b_ent);
    q.d_map[0] = a.d_code; q.d_map[1] = a.d_code_pos;
    q.B = B; q.P = P; q.S = a.S; q.K = K; q.H = a.H; q.W = a.W; q.n_sets = g.n_sets; q.n_neg = a.n_neg;
    const int list_blocks = gather ? 2 * B : 0;
    const int lds_lists = (2 * HW + 1 + nnb + 2 + 512 / 64 + 4) * 4;
    const int lds_tiles = gather && lds_lists > g.tile_bytes ? lds_lists : g.tile_bytes;
    if ((e = ensure_dynamic_lds(reinterpret_cast<const void*>(&wide_code_tiles_kernel), lds_tiles)) != hipSuccess) return e;
    p.zero[0] = a.d_code; p.zero[1] = a.d_code_pos;
    p.zero_floats = gather ? 0 : (long long)B * a.H * a.W * K;      // (the gather writes every pixel)
    const int zero_blocks = 2 * (int)((p.zero_floats + WB_ZERO_CHUNK - 1) / WB_ZERO_CHUNK);
    const int lds = 128 * WB_GS * 4 + 2 * g.tile_bytes;
    if ((e = ensure_dynamic_lds(reinterpret_cast<const void*>(&wide_bwd_kernel), lds)) != hipSuccess) return e;
    for (int win = 0; win < g.nwin; ++win) {
        // one pass per window of code channels (K <= 88: one): its tiles, then both GEMMs into the window's columns of the row gradients.  The
        // lists / the zeroing ride with the first pass
        p.k0 = win * g.Kc;
        p.Kc = K - p.k0 < g.Kc ? K - p.k0 : g.Kc;
        const int extra = win == 0 ? list_blocks + zero_blocks : 0;
        hipLaunchKernelGGL(wide_code_tiles_kernel, dim3(g.n_img * g.nb + extra), dim3(512), lds_tiles, stream, p, q, win == 0 ? list_blocks : 0);
        hipLaunchKernelGGL(wide_bwd_kernel, dim3(g.n_img), dim3(WB_THREADS), lds, stream, p);
    }
    if ((e = hipGetLastError()) != hipSuccess) return e;
    // the backward of norm() and of the sampling
    if (gather) {
        const long long rows_total = (long long)g.n_img * P;
        hipLaunchKernelGGL(wide_rows_kernel, dim3((unsigned)((rows_total + 3) / 4)), dim3(256), 0, stream, q);
        hipLaunchKernelGGL(wide_gather_kernel, dim3((unsigned)((2ll * B * HW + 3) / 4)), dim3(256), 0, stream, q);
        return hipGetLastError();
    }
    // (maps beyond 4096 pixels: one wave per point, global fp32 atomics into the maps the tile kernel's spare workgroups zeroed.  Measured and
    // dropped before the gather above: one workgroup per (map, image, band of pixel rows) accumulating its band in LDS - 190 us against 88:
    // ds_add_f32 runs at ~140 cycles per wave instruction, profiles/r05g_wide_path.txt)
    StegoMap dm{a.d_code, (int64_t)a.H * a.W * K, 1, (int64_t)a.W * K, K}, dmp{a.d_code_pos, (int64_t)a.H * a.W * K, 1, (int64_t)a.W * K, K};
    const size_t rowsB = (size_t)B * P;
    if ((e = launch_sample_scatter(p.d_rows, cn, inv, &dm, nullptr, B, K, a.H, a.W, co1, B, a.S, p.d_anchor, g.n_sets, (long long)rowsB * K,
                                   stream)) != hipSuccess) return e;
    if ((e = launch_sample_scatter(p.d_rows + rowsB * K, cn + rowsB * K, inv + rowsB, &dmp, nullptr, B, K, a.H, a.W, co2, B, a.S, nullptr, 0, 0,
                                   stream)) != hipSuccess) return e;
    if (nnb > 0 && (e = launch_sample_scatter(p.d_rows + 2 * rowsB * K, cn + 2 * rowsB * K, inv + 2 * rowsB, &dm, a.perms, nnb, K, a.H, a.W, co2, B, a.S,
                                              nullptr, 0, 0, stream)) != hipSuccess) return e;
    return hipSuccess;
}

}  // namespace stego
