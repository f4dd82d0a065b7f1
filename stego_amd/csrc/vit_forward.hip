// Frozen DINO ViT forward for gfx950 (include/stego_vit.h): the producer of the feature maps the correspondence loss
// reads.  Reference: src/dino/vision_transformer.py:69-130 (Attention / Mlp / Block), :123-133 (PatchEmbed),
// :195-205 (prepare_tokens), :225-237 (get_intermediate_feat, n = 1), called from src/modules.py:83-107.
//
// Data flow (B images, M = B * ntok token rows, D channels):
//   im2col -> patch GEMM (+bias +pos) -> fp32 residual stream [M][D]
//   per block:  LN1 -> fp16 panels -> QKV GEMM (+bias) -> Q,K [b][head][token][64], V^T [b][head][64][token] (fp16)
//               attention (flash style, S^T = K Q^T so that every per-query statistic lives in one lane)
//               -> fp16 panels -> proj GEMM (+bias) accumulated into the residual
//               LN2 -> panels -> FC1 GEMM (+bias, GELU) -> panels -> FC2 GEMM (+bias) accumulated into the residual
//   final LN -> fp32 tokens [B][ntok][D]
// "Panel" = the operand image every GEMM stages with linear LDS-DMA copies: [row block of 128][k chunk of 64]
// [128 rows][72 halves] (64 data + 8 pad: 144-byte rows, conflict-free ds_read_b128), 18 KiB each.  Producers
// (LayerNorm, attention, the GELU epilogue, im2col) write panels directly, so no GEMM ever re-lays-out its input.
#include <cstdlib>
#include <type_traits>

#include "../../include/stego_vit.h"
#include "corr_common.h"
#include "host_util.h"

namespace stego {
namespace vit {

constexpr int VP_ROWS = 128;
constexpr int VP_KC = 64;
constexpr int VP_LD = 72;
constexpr int VP_BYTES = VP_ROWS * VP_LD * 2;        // 18432 = 18 x 1 KiB
constexpr float LOG2E = 1.4426950408889634f;
// Precision F16X3 (StegoVitDesc.precision = 1, the default of the Python host): every operand of every matrix product is the pair
// (hi, lo) of fp16 values with x = hi + lo to 2^-22 (split_f16_pair, corr_common.h) and every product is hi*hi + hi*lo + lo*hi on the
// fp16 matrix cores, fp32 accumulate: the fp32-class arithmetic of the loss kernels and of the segmentation head, three MFMAs per
// product and twice the operand bytes.  A panel keeps its geometry and holds 32 k values instead of 64: halves [0, 32) of a row are
// the hi parts, [32, 64) the lo parts - the copies, the fragment reads and the producers' addressing stay what they are.
// Ranges: weights are packed times a power of two that puts the tensor's largest magnitude into [1024, 2048) (lo parts stay out of the
// fp16 subnormals; the power is undone in the GEMM epilogue, exact), activation panels hold 16 x the value (same reason), the softmax
// probabilities 1024 x.
constexpr float VIT_ASCALE = 16.f;
// an activation beyond +-4094 would make hi = inf and lo = NaN (the fp32 torch model stays finite): the scaled value saturates at the
// fp16 maximum instead - one v_med3_f32; no DINO checkpoint comes near it (LayerNorm bounds its outputs by sqrt(D) gamma)
__device__ __forceinline__ float x3_sat(float x) { return __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f); }
constexpr float VIT_PSCALE = 1024.f;

__device__ __forceinline__ int amax_exp(unsigned bits)       // e with amax = m * 2^e, m in [0.5, 1); 11 for an all-zero tensor (scale 1)
{
    const float a = __builtin_bit_cast(float, bits);
    return a > 0.f ? __builtin_amdgcn_frexp_expf(a) : 11;
}

enum { EPI_RESID = 0, EPI_EMBED = 1, EPI_GELU = 2, EPI_QKV = 3 };

struct GemmParams {
    const unsigned char* A;      // panels [mb][nkc]
    const unsigned char* W;      // panels [nb][nkc]   (rows = output features: nn.Linear weight [out][in])
    const float* bias;           // [N]
    int M, N, nkc;
    float* resid;                // RESID / EMBED: fp32 [rows][D]
    int ldr;
    unsigned char* outp;         // GELU: output panels [mb][out_nkc]
    int out_nkc;
    half_t* q;                   // QKV outputs
    half_t* k;
    half_t* vt;
    int D, heads, ntok, ntok_pad, hw;
    float inv_ntok, inv_hw, qscale;
    const float* pos;            // EMBED: [ntok][D]
    const unsigned* wamax;       // F16X3: max |w| of this GEMM's weight tensor (float bits) - the power of two it was packed with
    size_t plane;                // F16X3, QKV: halves from the hi plane of q / k / vt to the lo plane
    int debug;                   // STEGO_DEBUG_VIT ablations: 1 = no MFMAs, 2 = no stage copies after the first, 4 = no epilogue
};

__device__ __forceinline__ void lds_copy_kib(const unsigned char* __restrict__ gsrc, unsigned char* lds_dst, int lane)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + lane * 16),
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// ------------------------------------------------------------------------------------------------- packing
// fp32 [R][K] row-major -> panels [ceil(R/128)][K/64] (F16X3: [..][K/32], scaled), rows beyond R zero.  One thread per 8 columns.
template <bool X3>
__global__ void __launch_bounds__(256) vit_pack_kernel(const float* __restrict__ src, int R, int K, unsigned char* __restrict__ dst,
                                                       int nrb, int nkc, const unsigned* __restrict__ amax)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)nrb * 128 * nkc * 8;
    if (t >= total) return;
    const int g = (int)(t & 7);
    const long long u = t >> 3;
    const int kc = (int)(u % nkc);
    const int row = (int)(u / nkc);
    f16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (half_t)0.f;
    if (row < R) {
        const int col = X3 ? kc * 32 + (g & 3) * 8 : kc * 64 + g * 8;
        const float* s = src + (size_t)row * K + col;
        f32x4 a = *reinterpret_cast<const f32x4*>(s), b = *reinterpret_cast<const f32x4*>(s + 4);
        if constexpr (X3) {
            const float sc = __builtin_ldexpf(1.f, 11 - amax_exp(*amax));
            a = a * sc;
            b = b * sc;
        }
        v = f16x8{(half_t)a[0], (half_t)a[1], (half_t)a[2], (half_t)a[3], (half_t)b[0], (half_t)b[1], (half_t)b[2], (half_t)b[3]};
        if (X3 && g >= 4)
            v = f16x8{(half_t)(a[0] - (float)v[0]), (half_t)(a[1] - (float)v[1]), (half_t)(a[2] - (float)v[2]), (half_t)(a[3] - (float)v[3]),
                      (half_t)(b[0] - (float)v[4]), (half_t)(b[1] - (float)v[5]), (half_t)(b[2] - (float)v[6]), (half_t)(b[3] - (float)v[7])};
    }
    unsigned char* d = dst + ((size_t)(row >> 7) * nkc + kc) * VP_BYTES + (((row & 127) * VP_LD) + g * 8) * 2;
    *reinterpret_cast<f16x8*>(d) = v;
}

// max |w| of a tensor as float bits (non-negative floats order like unsigned integers); *out zeroed by the caller
__global__ void __launch_bounds__(256) vit_absmax_kernel(const float* __restrict__ src, long long n, unsigned* __restrict__ out)
{
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) m = fmaxf(m, fabsf(src[i]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __builtin_bit_cast(unsigned, m));
}

// Patches as GEMM rows (the conv of PatchEmbed, vision_transformer.py:123-133, is a GEMM over k = (c, iy, ix)):
// img fp32 [B][3][H][W] -> panels [ceil(B*hw/128)][3*ps*ps/64] (F16X3: [..][3*ps*ps/32], 16 x the pixel).  One thread per 8 consecutive ix.
template <bool X3>
__global__ void __launch_bounds__(256) vit_im2col_kernel(const float* __restrict__ img, int B, int H, int W, int ps,
                                                         unsigned char* __restrict__ dst, int nrb, int nkc)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)nrb * 128 * nkc * 8;
    if (t >= total) return;
    const int g = (int)(t & 7);
    const long long u = t >> 3;
    const int kc = (int)(u % nkc);
    const int row = (int)(u / nkc);
    const int w = W / ps, hw = (H / ps) * w;
    f16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (half_t)0.f;
    if (row < B * hw) {
        const int b = row / hw, pi = row - b * hw, py = pi / w, px = pi - py * w;
        const int k0 = X3 ? kc * 32 + (g & 3) * 8 : kc * 64 + g * 8, pp = ps * ps;
        const int c = k0 / pp, rem = k0 - c * pp, iy = rem / ps, ix = rem - iy * ps;
        const float* s = img + (((size_t)b * 3 + c) * H + (size_t)py * ps + iy) * W + (size_t)px * ps + ix;
        f32x4 a = *reinterpret_cast<const f32x4*>(s), bb = *reinterpret_cast<const f32x4*>(s + 4);
        if constexpr (X3) {
            a = a * VIT_ASCALE;
            bb = bb * VIT_ASCALE;
        }
        v = f16x8{(half_t)a[0], (half_t)a[1], (half_t)a[2], (half_t)a[3], (half_t)bb[0], (half_t)bb[1], (half_t)bb[2], (half_t)bb[3]};
        if (X3 && g >= 4)
            v = f16x8{(half_t)(a[0] - (float)v[0]), (half_t)(a[1] - (float)v[1]), (half_t)(a[2] - (float)v[2]), (half_t)(a[3] - (float)v[3]),
                      (half_t)(bb[0] - (float)v[4]), (half_t)(bb[1] - (float)v[5]), (half_t)(bb[2] - (float)v[6]), (half_t)(bb[3] - (float)v[7])};
    }
    unsigned char* d = dst + ((size_t)(row >> 7) * nkc + kc) * VP_BYTES + (((row & 127) * VP_LD) + g * 8) * 2;
    *reinterpret_cast<f16x8*>(d) = v;
}

// Row 0 of every image: cls_token + pos_embed[0] (prepare_tokens, vision_transformer.py:199-203).
__global__ void __launch_bounds__(256) vit_cls_kernel(const float* __restrict__ cls, const float* __restrict__ pos, float* __restrict__ resid,
                                                      int B, int D, int ntok)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= B * D) return;
    const int b = t / D, n = t - b * D;
    resid[(size_t)b * ntok * D + n] = cls[n] + pos[n];
}

// ------------------------------------------------------------------------------------------------- LayerNorm
// One wave per token row (D <= 768: up to 12 values per lane), fp32 statistics, biased variance (nn.LayerNorm).
template <bool TO_PANEL, bool X3>
__global__ void __launch_bounds__(256) vit_layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int M, int D, float eps,
                                                            unsigned char* __restrict__ outp, float* __restrict__ outf)
{
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int ni = D >> 6;
    const float* xr = x + (size_t)m * D;
    float v[12];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        v[i] = i < ni ? xr[lane + 64 * i] : 0.f;
        s += v[i];
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const float dlt = i < ni ? v[i] - mean : 0.f;
        q += dlt * dlt;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = rsqrtf(q / (float)D + eps);
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        if (i < ni) {
            const int n = lane + 64 * i;
            const float y = (v[i] - mean) * rstd * gamma[n] + beta[n];
            if constexpr (TO_PANEL && X3) {           // column n = 64 i + lane: k chunk 2 i + (lane >> 5), hi at lane & 31, lo 32 further
                const float ys = x3_sat(y * VIT_ASCALE);
                const half_t hi = (half_t)ys;
                unsigned char* d = outp + ((size_t)(m >> 7) * (2 * ni) + 2 * i + (lane >> 5)) * VP_BYTES + (((m & 127) * VP_LD) + (lane & 31)) * 2;
                *reinterpret_cast<half_t*>(d) = hi;
                *reinterpret_cast<half_t*>(d + 64) = (half_t)(ys - (float)hi);
            } else if constexpr (TO_PANEL) {
                unsigned char* d = outp + ((size_t)(m >> 7) * ni + i) * VP_BYTES + (((m & 127) * VP_LD) + lane) * 2;
                *reinterpret_cast<half_t*>(d) = (half_t)y;
            } else {
                outf[(size_t)m * D + n] = y;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------- GEMM
// C[m][n] = sum_k A[m][k] W[n][k]  (+ epilogue): 256 x 192 tile per workgroup (every N of a ViT is a multiple of
// 192 = 3 x 64), 8 waves as 4 x 2, each 64 x 96 (2 x 3 MFMA blocks, 96 accumulator registers), fp16 operands on
// v_mfma_f32_32x32x16_f16, fp32 accumulate.  Both operands arrive as panels, so one 64-deep k stage is five linear
// LDS-DMA copies (two A panels + 192 W rows, 63 KiB), double buffered: 126 KiB of LDS, one workgroup per CU.
// The first version used 128 x 128 tiles (64 flop per staged byte): its k loop ran at the latency of the copies
// (~1 us per 64-deep stage for 0.2 us of MFMA work, 230-330 TFLOP/s); this shape stages 97 flop per byte and gives
// the MFMA stream of one stage about the time the next stage's copies need.
// Workgroup -> tile map: the workgroups of one XCD (blockIdx % 8) walk the column tiles of the same row tiles, so an
// A row tile is pulled into one L2 only (round-robin dispatch would send its column tiles to 8 different L2s).
constexpr int GT_M = 256, GT_N = 192;
constexpr int GT_STAGE = (GT_M + GT_N) * VP_LD * 2;       // 64512
constexpr int GT_LDS = GT_STAGE;                         // ONE stage per workgroup, two workgroups per CU (see below)
constexpr int GT_THREADS = 512;

template <int EPI, bool X3>
__global__ void __launch_bounds__(GT_THREADS, 4) vit_gemm_kernel(const GemmParams p, const int mtiles, const int ntiles)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int mt = xcd + 8 * (slot / ntiles), nt = slot % ntiles;
    if (mt >= mtiles) return;
    const unsigned char* Ap = p.A + (size_t)(2 * mt) * p.nkc * VP_BYTES;        // row panels 2mt, 2mt+1
    // the 192 W rows of column tile nt: rows [r0, 128) of panel p0, then the first 192 - (128 - r0) rows of panel p0 + 1
    const int p0 = (GT_N * nt) >> 7, r0 = (GT_N * nt) & 127;
    const unsigned char* W0 = p.W + (size_t)p0 * p.nkc * VP_BYTES + r0 * (VP_LD * 2);
    const unsigned char* W1 = p.W + (size_t)(p0 + 1) * p.nkc * VP_BYTES;
    const int w0_pieces = (128 - r0) * (VP_LD * 2) / 1024;                      // 18 or 9
    auto issue = [&](int c) {
        unsigned char* dst = smem;
        const size_t ko = (size_t)c * VP_BYTES;
        for (int pc = wave; pc < 63; pc += 8) {
            const unsigned char* src;
            if (pc < 18) src = Ap + ko + pc * 1024;
            else if (pc < 36) src = Ap + (size_t)p.nkc * VP_BYTES + ko + (pc - 18) * 1024;
            else if (pc - 36 < w0_pieces) src = W0 + ko + (pc - 36) * 1024;
            else src = W1 + ko + (pc - 36 - w0_pieces) * 1024;
            lds_copy_kib(src, dst + pc * 1024, lane);
        }
    };
    f32x16 acc[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int r = lane & 31, half = lane >> 5;
    for (int c = 0; c < p.nkc; ++c) {
        // One stage per workgroup and no prefetch inside it: the copies of a stage are hidden by the OTHER workgroup of
        // this CU, which is in its MFMAs or its epilogue meanwhile (two 63 KiB stages + the epilogue parks fit the LDS
        // only this way; a double-buffered single workgroup per CU left the CU idle during every epilogue).
        if (c > 0) __syncthreads();                        // everyone is done reading chunk c-1
        if (c == 0 || !(p.debug & 2)) issue(c);
        sync_after_lds_dma();                              // chunk c landed
        if (p.debug & 1) continue;
        const half_t* As = reinterpret_cast<const half_t*>(smem);
        const half_t* Bs = As + GT_M * VP_LD;
        const half_t* ap = As + (64 * wr + r) * VP_LD + 8 * half;
        const half_t* bp = Bs + (96 * wc + r) * VP_LD + 8 * half;
        if constexpr (X3) {
            // a chunk holds 32 k values: halves [kk, kk + 16) of a row are hi parts, 32 further their lo parts; lo*hi and hi*lo first,
            // one term at a time over the six accumulators (no MFMA waits for the one issued before it)
#pragma unroll
            for (int kk = 0; kk < 32; kk += 16) {
                const f16x8 a0h = *reinterpret_cast<const f16x8*>(ap + kk), a1h = *reinterpret_cast<const f16x8*>(ap + 32 * VP_LD + kk);
                const f16x8 a0l = *reinterpret_cast<const f16x8*>(ap + 32 + kk), a1l = *reinterpret_cast<const f16x8*>(ap + 32 * VP_LD + 32 + kk);
                const f16x8 b0h = *reinterpret_cast<const f16x8*>(bp + kk), b1h = *reinterpret_cast<const f16x8*>(bp + 32 * VP_LD + kk);
                const f16x8 b2h = *reinterpret_cast<const f16x8*>(bp + 64 * VP_LD + kk);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0l, b0h, acc[0][0], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l, b0h, acc[1][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0l, b1h, acc[0][1], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l, b1h, acc[1][1], 0, 0, 0);
                acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0l, b2h, acc[0][2], 0, 0, 0);
                acc[1][2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l, b2h, acc[1][2], 0, 0, 0);
                const f16x8 b0l = *reinterpret_cast<const f16x8*>(bp + 32 + kk), b1l = *reinterpret_cast<const f16x8*>(bp + 32 * VP_LD + 32 + kk);
                const f16x8 b2l = *reinterpret_cast<const f16x8*>(bp + 64 * VP_LD + 32 + kk);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, b0l, acc[0][0], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, b0l, acc[1][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, b1l, acc[0][1], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, b1l, acc[1][1], 0, 0, 0);
                acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, b2l, acc[0][2], 0, 0, 0);
                acc[1][2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, b2l, acc[1][2], 0, 0, 0);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, b0h, acc[0][0], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, b0h, acc[1][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, b1h, acc[0][1], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, b1h, acc[1][1], 0, 0, 0);
                acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, b2h, acc[0][2], 0, 0, 0);
                acc[1][2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, b2h, acc[1][2], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < VP_KC; kk += 16) {
                const f16x8 a0 = *reinterpret_cast<const f16x8*>(ap + kk), a1 = *reinterpret_cast<const f16x8*>(ap + 32 * VP_LD + kk);
                const f16x8 b0 = *reinterpret_cast<const f16x8*>(bp + kk), b1 = *reinterpret_cast<const f16x8*>(bp + 32 * VP_LD + kk);
                const f16x8 b2 = *reinterpret_cast<const f16x8*>(bp + 64 * VP_LD + kk);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0][0], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[1][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[1][1], 0, 0, 0);
                acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b2, acc[0][2], 0, 0, 0);
                acc[1][2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b2, acc[1][2], 0, 0, 0);
            }
        }
    }
    // ---- epilogue.  C/D layout of the accumulators: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
    if ((p.debug & 4) && acc[0][0][0] != 12345.678f) return;
    // F16X3: the accumulators hold (16 x activation) . (2^s x weight): one exact power of two puts them back
    const float osc = X3 ? __builtin_ldexpf(1.f / VIT_ASCALE, amax_exp(*p.wamax) - 11) : 1.f;
    if constexpr (EPI == EPI_GELU || EPI == EPI_QKV) {
        // fp16 outputs go through LDS (the stage buffer is free now) and leave in the layout of their consumer with
        // wide, coalesced stores, 128 rows per pass.  Storing straight from the accumulators (a lane owns one column:
        // 2-byte stores, 64 B runs, or fully scattered for V^T) cost more than the whole k loop.
        // F16X3 has twice the bytes to park and the same LDS: four passes - GELU by column half (three 32-column chunk panels, hi |
        // lo in one row) in two balanced sets, QKV by plane (the hi values of all three 64-column groups, then the lo values into the lo arrays).
        half_t* T = reinterpret_cast<half_t*>(smem);
        constexpr int GRP = VP_ROWS * VP_LD;            // halves per 64-column group region of one pass (18 KiB)
        constexpr int LDV = VP_ROWS + 8;                // V^T rows: 128 tokens + pad
        // nn.GELU (exact form) with erf from Abramowitz-Stegun 7.1.26 (|err| < 1.5e-7: below the fp16 rounding of the F16 result
        // and - as an ABSOLUTE error of 0.75e-7 |v| on the product - inside the fp32 class of F16X3, tests/test_vit_native.py;
        // ocml erff is ~3x the instructions and was a third of this kernel's time).  (Finishing all 96 values of a wave before the
        // passes - so that no wave waits at a barrier while two compute - made the compiler spill 400 registers inside the k loop.)
        auto gelu = [](float v) {
            const float x = fabsf(v) * 0.70710678118654752f;
            const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * x);
            const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
            const float erfa = 1.f - poly * __builtin_amdgcn_exp2f(-x * x * LOG2E);
            return 0.5f * v * (1.f + copysignf(erfa, v));
        };
        for (int pass = 0; pass < (X3 ? 4 : 2); ++pass) {
            const int rh = X3 ? pass >> 1 : pass;       // row half of the tile
            const int sub = pass & 1;                   // F16X3: column half (GELU) / plane (QKV)
            __syncthreads();                            // stage buffer / previous pass's park region is free
            if ((wr >> 1) == rh) {
#pragma unroll
                for (int ni = 0; ni < 3; ++ni) {
                    // F16X3 GELU: the six 32-column chunk panels of a row half leave three per pass, and all four waves of the row
                    // half work in both passes (two blocks and one, then one and two) - by column half only two of eight waves computed
                    // at a time.  Slot of (wc, ni) in its pass: sub 0 = {(0,0) (0,1) (1,0)}, sub 1 = {(0,2) (1,1) (1,2)}
                    const int gslot = sub == 0 ? (wc == 0 ? (ni < 2 ? ni : -1) : (ni == 0 ? 2 : -1)) : (wc == 0 ? (ni == 2 ? 0 : -1) : (ni >= 1 ? ni : -1));
                    if (X3 && EPI == EPI_GELU && gslot < 0) continue;
                    const int col = 96 * wc + 32 * ni + r, cg = col >> 6, cl = col & 63;
                    const int n = nt * GT_N + col;
                    const float bias = (p.bias && n < p.N) ? p.bias[n] : 0.f;
                    const int n0 = nt * GT_N + 64 * cg;     // first column of this lane's 64-column group (wave-uniform)
                    const int which = EPI == EPI_QKV ? n0 / p.D : 0;
                    const float oscale = which == 0 && EPI == EPI_QKV ? p.qscale : 1.f;
                    auto park = [&](auto transposed) {
#pragma unroll
                        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                            for (int e = 0; e < 16; ++e) {
                                const int lrow = 64 * (wr & 1) + 32 * mi + (e & 3) + 8 * (e >> 2) + 4 * half;
                                const float v = acc[mi][ni][e] * osc + bias;
                                if constexpr (EPI == EPI_GELU && X3) {
                                    const float g = x3_sat(gelu(v) * VIT_ASCALE);
                                    const half_t hi = (half_t)g;
                                    T[gslot * GRP + lrow * VP_LD + r] = hi;
                                    T[gslot * GRP + lrow * VP_LD + 32 + r] = (half_t)(g - (float)hi);
                                } else if constexpr (EPI == EPI_GELU) {
                                    T[cg * GRP + lrow * VP_LD + cl] = (half_t)gelu(v);
                                } else {
                                    const float w = v * oscale;
                                    half_t o = (half_t)w;
                                    if (X3 && sub) o = (half_t)(w - (float)o);
                                    if constexpr (decltype(transposed)::value) T[cg * GRP + cl * LDV + lrow] = o;       // V^T
                                    else T[cg * GRP + lrow * VP_LD + cl] = o;                                       // Q (pre-scaled), K
                                }
                            }
                    };
                    if (EPI == EPI_QKV && which == 2) park(std::true_type{});
                    else park(std::false_type{});
                }
            }
            __syncthreads();
            if constexpr (EPI == EPI_GELU) {
                // the three [128][72] images are panels already: linear 16-byte copies
                for (int cg = 0; cg < 3; ++cg) {
                    // (F16X3: slot cg of pass sub holds chunk 3 wc + ni of the tile, see the park above)
                    const int kc = X3 ? nt * 6 + (sub == 0 ? (cg < 2 ? cg : 3) : (cg == 0 ? 2 : 3 + cg)) : nt * 3 + cg;
                    if (kc >= p.out_nkc) continue;
                    unsigned char* dst = p.outp + ((size_t)(2 * mt + rh) * p.out_nkc + kc) * VP_BYTES;
                    const unsigned char* src = smem + cg * VP_BYTES;
                    for (int i = tid; i < VP_BYTES / 16; i += GT_THREADS)
                        *reinterpret_cast<f32x4*>(dst + i * 16) = *reinterpret_cast<const f32x4*>(src + i * 16);
                }
            } else {
                const size_t po = X3 && sub ? p.plane : 0;
                for (int cg = 0; cg < 3; ++cg) {
                    const int n0 = nt * GT_N + 64 * cg;
                    if (n0 >= p.N) continue;
                    const int which = n0 / p.D, head = (n0 - which * p.D) >> 6;
                    if (which < 2) {                     // Q / K: [b][head][token][64], one 128-byte row per token
                        half_t* dstb = (which == 0 ? p.q : p.k) + po;
                        for (int i = tid; i < VP_ROWS * 8; i += GT_THREADS) {
                            const int lrow = i >> 3, ch = i & 7;
                            const int m = mt * GT_M + 128 * rh + lrow;
                            if (m >= p.M) continue;
                            const int b = (int)(((float)m + 0.5f) * p.inv_ntok);
                            const int t = m - b * p.ntok;
                            half_t* dst = dstb + (((size_t)b * p.heads + head) * p.ntok_pad + t) * 64 + ch * 8;
                            *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(T + cg * GRP + lrow * VP_LD + ch * 8);
                        }
                    } else {                             // V^T: [b][head][64][token]; lanes run along the tokens
                        const int lrow = tid & (VP_ROWS - 1);
                        const int m = mt * GT_M + 128 * rh + lrow;
                        if (m < p.M) {
                            const int b = (int)(((float)m + 0.5f) * p.inv_ntok);
                            const int t = m - b * p.ntok;
                            half_t* dst = p.vt + po + (((size_t)b * p.heads + head) * 64) * p.ntok_pad + t;
                            for (int d = tid >> 7; d < 64; d += GT_THREADS / VP_ROWS)
                                dst[(size_t)d * p.ntok_pad] = T[cg * GRP + d * LDV + lrow];
                        }
                    }
                }
            }
        }
    } else {
        const size_t rbytes = (size_t)p.M * p.ldr * 4;
        const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(p.resid, 0, rbytes < 0xFFFFFFFFull ? (unsigned)rbytes : 0xFFFFFFFFu, 0x00020000);
        const unsigned rstride = (unsigned)p.ldr * 4u;
        constexpr int RB = 8;                         // read-modify-writes in flight per accumulator block (16: same time, 6 spilled registers)
#pragma unroll
        for (int ni = 0; ni < 3; ++ni) {
            const int n = nt * GT_N + 96 * wc + 32 * ni + r;
            if (n >= p.N) continue;
            const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int mbase = mt * GT_M + 64 * wr + 32 * mi + 4 * half;
                if constexpr (EPI == EPI_RESID) {
                    // all 16 loads of an accumulator block first (interleaved read-modify-writes through one pointer would be serialised
                    // by the compiler: every load ordered behind the previous store, one memory round trip each), then the 16 stores.
                    // Buffer accesses: rows beyond M are out of the descriptor's range (loads return 0, stores are dropped) - no branch
                    // per element - and an address is ONE 32-bit register (round 4: B = 64 forward 14.27 -> 13.83 ms, F16 6.95 -> 6.53 ms
                    // same box against 64-bit pointers + a bounds branch per element)
                    const unsigned off0 = ((unsigned)mbase * (unsigned)p.ldr + (unsigned)n) * 4u;
#pragma unroll
                    for (int e0 = 0; e0 < 16; e0 += RB) {
                        float old[RB];
#pragma unroll
                        for (int e = 0; e < RB; ++e)
                            old[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrsrc, off0 + (unsigned)(((e0 + e) & 3) + 8 * ((e0 + e) >> 2)) * rstride, 0, 0));
#pragma unroll
                        for (int e = 0; e < RB; ++e)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, old[e] + (acc[mi][ni][e0 + e] * osc + bias)), rrsrc,
                                                                  off0 + (unsigned)(((e0 + e) & 3) + 8 * ((e0 + e) >> 2)) * rstride, 0, 0);
                    }
                } else {                              // EPI_EMBED: patch rows -> token rows 1.. of their image, + pos_embed
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int m = mbase + (e & 3) + 8 * (e >> 2);
                        if (m >= p.M) continue;
                        const int b = (int)(((float)m + 0.5f) * p.inv_hw);
                        const int pi = m - b * p.hw;
                        p.resid[((size_t)b * p.ntok + 1 + pi) * p.ldr + n] = acc[mi][ni][e] * osc + bias + p.pos[(size_t)(1 + pi) * p.D + n];
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------- attention
// softmax(Q K^T / sqrt(64)) V for one (image, head, 128-query block) per workgroup; each wave owns 32 queries.
// The scores are computed TRANSPOSED, S^T = K Q^T (keys x queries): in the MFMA result layout a lane then holds one
// query (column) and 16 of the 32 keys of a block in its registers, so the running max / sum / rescale of the online
// softmax are per-lane scalars (one exchange with lane^32 per key block) and P^T can be fed back as the B operand of
// O^T = V^T P^T without leaving the registers: the MFMA sums over k, so any k order works as long as both operands
// use the same one - V^T is read in the order the accumulator registers happen to hold the keys.
// K and V^T tiles (64 keys) are shared by the 4 waves through LDS: linear LDS-DMA copies whose SOURCE addresses are
// permuted (16-byte chunk c of row r lands in slot c ^ ((r >> 1) & 7)) so that the fragment reads are conflict-free
// without padding.  Q is pre-scaled by log2(e)/8 in the QKV epilogue: p = exp2(s - m).
struct AttnParams {
    const half_t* q;
    const half_t* k;
    const half_t* vt;
    unsigned char* outp;         // panels [mb][D/64]: row = b*ntok + query, k chunk = head
    int out_nkc, heads, ntok, ntok_pad;
    int nqb, units;              // query blocks per (image, head); number of (image, head) pairs
    size_t plane;                // F16X3: halves from the hi plane of q / k / vt to the lo plane
};

__device__ __forceinline__ const unsigned char* swz16(const unsigned char* base, int row, int cb)
{
    return base + ((row * 8 + (cb ^ ((row >> 1) & 7))) << 4);
}

template <bool X3>
__global__ void __launch_bounds__(256, 2) vit_attn_kernel(const AttnParams p)
{
    constexpr int NPL = X3 ? 2 : 1;                      // operand planes: hi (| lo)
    __shared__ __attribute__((aligned(16))) unsigned char smem[2][2 * NPL][8192];       // [stage][K hi | V^T hi (| K lo | V^T lo)]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // The query blocks of one (image, head) run on ONE XCD (blockIdx % 8), next to each other in launch order, so its K / V^T
    // are pulled into one L2 once: with a plain (qb, h, b) grid the 7 blocks of a head went to 7 different L2s and the
    // kernel fetched 599 MB for 116 MB of q, k, v (rocprofv3 FETCH_SIZE), i.e. it ran at the HBM rate.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int unit = xcd + 8 * (slot / p.nqb), qb = slot % p.nqb;
    if (unit >= p.units) return;
    const int b = unit / p.heads, h = unit - b * p.heads;
    const size_t bh = (size_t)unit;
    const half_t* Kg = p.k + bh * p.ntok_pad * 64;
    const half_t* Vg = p.vt + bh * 64 * p.ntok_pad;
    const int li = lane & 31, half = lane >> 5;
    const int q0 = qb * 128 + wave * 32;
    const int nkb = p.ntok_pad >> 6;

    auto issue = [&](int kb) {
        for (int pc = wave; pc < 8; pc += 4) {
            const int s = pc * 64 + lane;
            const int row = s >> 3, cb = (s & 7) ^ ((row >> 1) & 7);
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Kg + pl * p.plane + ((size_t)(kb * 64 + row)) * 64 + cb * 8),
                                                 (__attribute__((address_space(3))) void*)(&smem[kb & 1][2 * pl][pc * 1024]), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Vg + pl * p.plane + (size_t)row * p.ntok_pad + kb * 64 + cb * 8),
                                                 (__attribute__((address_space(3))) void*)(&smem[kb & 1][2 * pl + 1][pc * 1024]), 16, 0, 0);
            }
        }
    };
    issue(0);

    // Q fragments (B operand of S^T): lane (query li, k half) holds Q[q][8*(2s+half) .. +7]
    f16x8 qf[NPL][4];
    {
        const int qrow = min(q0 + li, p.ntok_pad - 1);
        const half_t* qp = p.q + (bh * p.ntok_pad + qrow) * 64 + 8 * half;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int s = 0; s < 4; ++s) qf[pl][s] = *reinterpret_cast<const f16x8*>(qp + pl * p.plane + 16 * s);
    }
    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[i][e] = 0.f;
    float mrun = -INFINITY, lrun = 0.f;

    auto tile = [&](const int kb, auto last_tile) {
        sync_after_lds_dma();                              // tile kb landed; tile kb-1's buffer is free
        if (kb + 1 < nkb) issue(kb + 1);
        const unsigned char* Ks = &smem[kb & 1][0][0];
        const unsigned char* Vs = &smem[kb & 1][1][0];
        // ---- S^T = K Q^T for the 64 keys of this tile
        f32x16 st[2];
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) {                      // the two 32-key accumulators alternate: no MFMA waits for the
            if constexpr (X3) {                            // result of the one issued just before it
                f16x8 ah[2], al[2];
#pragma unroll
                for (int kblk = 0; kblk < 2; ++kblk) {
                    ah[kblk] = *reinterpret_cast<const f16x8*>(swz16(Ks, 32 * kblk + li, 2 * s + half));
                    al[kblk] = *reinterpret_cast<const f16x8*>(swz16(Ks + 2 * 8192, 32 * kblk + li, 2 * s + half));
                }
#pragma unroll
                for (int kblk = 0; kblk < 2; ++kblk) st[kblk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[kblk], qf[0][s], s == 0 ? zero16 : st[kblk], 0, 0, 0);
#pragma unroll
                for (int kblk = 0; kblk < 2; ++kblk) st[kblk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kblk], qf[NPL - 1][s], st[kblk], 0, 0, 0);
#pragma unroll
                for (int kblk = 0; kblk < 2; ++kblk) st[kblk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kblk], qf[0][s], st[kblk], 0, 0, 0);
            } else {
#pragma unroll
                for (int kblk = 0; kblk < 2; ++kblk) {
                    const f16x8 a = *reinterpret_cast<const f16x8*>(swz16(Ks, 32 * kblk + li, 2 * s + half));
                    st[kblk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, qf[0][s], s == 0 ? zero16 : st[kblk], 0, 0, 0);
                }
            }
        }
        if constexpr (decltype(last_tile)::value) {       // keys beyond the sequence (the padding of the last tile)
#pragma unroll
            for (int kblk = 0; kblk < 2; ++kblk)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = kb * 64 + 32 * kblk + (e & 3) + 8 * (e >> 2) + 4 * half;
                    st[kblk][e] = key < p.ntok ? st[kblk][e] : -INFINITY;
                }
        }
        // ---- online softmax: this lane = one query; the other 16 keys of each block sit in lane ^ 32
        float mloc = st[0][0];
#pragma unroll
        for (int kblk = 0; kblk < 2; ++kblk)
#pragma unroll
            for (int e = 0; e < 16; ++e) mloc = fmaxf(mloc, st[kblk][e]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float mnew = fmaxf(mrun, mloc);
        const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
        mrun = mnew;
        float psum = 0.f;
#pragma unroll
        for (int kblk = 0; kblk < 2; ++kblk)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                st[kblk][e] = __builtin_amdgcn_exp2f(st[kblk][e] - mnew);
                psum += st[kblk][e];
            }
        lrun = lrun * alpha + psum;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[i][e] *= alpha;
        // ---- O^T += V^T P^T, 16 keys per MFMA; registers 8j..8j+7 of a 32-key block hold (per k half)
        //      keys 16j + {0..3} + 4*half and 16j + 8 + {0..3} + 4*half
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        auto vfrag = [&](const unsigned char* base, int drow, int step) {
            const u32x2 lo = *reinterpret_cast<const u32x2*>(swz16(base, drow, 2 * step) + 8 * half);
            const u32x2 hi = *reinterpret_cast<const u32x2*>(swz16(base, drow, 2 * step + 1) + 8 * half);
            const u32x4 packed = {lo[0], lo[1], hi[0], hi[1]};
            return __builtin_bit_cast(f16x8, packed);
        };
#pragma unroll
        for (int step = 0; step < 4; ++step) {
            const int kblk = step >> 1, j = step & 1;
            f16x8 pf, pl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if constexpr (X3) {
                    const float ps = st[kblk][8 * j + e] * VIT_PSCALE;
                    pf[e] = (half_t)ps;
                    pl[e] = (half_t)(ps - (float)pf[e]);
                } else {
                    pf[e] = (half_t)st[kblk][8 * j + e];
                }
            }
            if constexpr (X3) {
                f16x8 vh[2], vl[2];
#pragma unroll
                for (int dblk = 0; dblk < 2; ++dblk) {
                    vh[dblk] = vfrag(Vs, 32 * dblk + li, step);
                    vl[dblk] = vfrag(Vs + 2 * 8192, 32 * dblk + li, step);
                }
#pragma unroll
                for (int dblk = 0; dblk < 2; ++dblk) o[dblk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl[dblk], pf, o[dblk], 0, 0, 0);
#pragma unroll
                for (int dblk = 0; dblk < 2; ++dblk) o[dblk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[dblk], pl, o[dblk], 0, 0, 0);
#pragma unroll
                for (int dblk = 0; dblk < 2; ++dblk) o[dblk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[dblk], pf, o[dblk], 0, 0, 0);
            } else {
#pragma unroll
                for (int dblk = 0; dblk < 2; ++dblk)
                    o[dblk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfrag(Vs, 32 * dblk + li, step), pf, o[dblk], 0, 0, 0);
            }
        }
    };
    for (int kb = 0; kb + 1 < nkb; ++kb) tile(kb, std::false_type{});
    tile(nkb - 1, std::true_type{});
    // ---- normalise and store: O^T[d][q], this lane's query, 4 consecutive d per register group
    lrun += __shfl_xor(lrun, 32, 64);
    const float inv = X3 ? VIT_ASCALE / (lrun * VIT_PSCALE) : 1.f / lrun;
    const int qi = q0 + li;
    if (qi < p.ntok) {
        const int m = b * p.ntok + qi;
        if constexpr (X3) {       // head h = k chunks 2h, 2h + 1 of the proj GEMM's operand: hi at the channel's slot, lo 32 halves further
#pragma unroll
            for (int dblk = 0; dblk < 2; ++dblk) {
                unsigned char* orow = p.outp + ((size_t)(m >> 7) * p.out_nkc + 2 * h + dblk) * VP_BYTES + ((m & 127) * VP_LD) * 2;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c0 = 8 * g + 4 * half;
                    unsigned h01, l01, h23, l23;
                    split_f16_pair(o[dblk][4 * g] * inv, o[dblk][4 * g + 1] * inv, h01, l01);
                    split_f16_pair(o[dblk][4 * g + 2] * inv, o[dblk][4 * g + 3] * inv, h23, l23);
                    *reinterpret_cast<u32x2*>(orow + c0 * 2) = u32x2{h01, h23};
                    *reinterpret_cast<u32x2*>(orow + (32 + c0) * 2) = u32x2{l01, l23};
                }
            }
        } else {
            unsigned char* orow = p.outp + ((size_t)(m >> 7) * p.out_nkc + h) * VP_BYTES + ((m & 127) * VP_LD) * 2;
#pragma unroll
            for (int dblk = 0; dblk < 2; ++dblk)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d0 = 32 * dblk + 8 * g + 4 * half;
                    const f16x2 v01 = {(half_t)(o[dblk][4 * g] * inv), (half_t)(o[dblk][4 * g + 1] * inv)};
                    const f16x2 v23 = {(half_t)(o[dblk][4 * g + 2] * inv), (half_t)(o[dblk][4 * g + 3] * inv)};
                    *reinterpret_cast<u32x2*>(orow + d0 * 2) = u32x2{__builtin_bit_cast(unsigned, v01), __builtin_bit_cast(unsigned, v23)};
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------- host side
size_t up256(size_t v) { return (v + 255) / 256 * 256; }

struct Layout {                    // byte offsets inside the packed weight blob
    int ntok, hw, Kp;
    size_t patch_w, patch_b, cls, pos;
    struct Blk { size_t ln1_w, ln1_b, qkv_w, qkv_b, proj_w, proj_b, ln2_w, ln2_b, fc1_w, fc1_b, fc2_w, fc2_b; };
    size_t blk0, blk_stride;       // blocks are identical in size
    Blk rel;                       // offsets relative to a block's start
    size_t norm_w, norm_b, amax, total;      // amax: F16X3, 1 + 4 * depth words - max |w| of the patch GEMM, then qkv / proj / fc1 / fc2 per block
    int kw;                        // k values per panel chunk: 64, F16X3 32
};

int nblk128(int n) { return (n + 127) / 128; }
int ntiles192(int n) { return (n + 191) / 192; }
int wpanels(int rows) { return nblk128(ntiles192(rows) * 192); }       // weight panels incl. the padding of the last column tile
int apanels(int rows) { return (rows + 255) / 256 * 2; }               // activation panels: whole 256-row tiles

Layout make_layout(const StegoVitDesc& d)
{
    Layout L;
    L.hw = (d.H / d.patch) * (d.W / d.patch);
    L.ntok = L.hw + 1;
    L.Kp = 3 * d.patch * d.patch;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o = up256(o + bytes); return at; };
    // weight panels cover whole 192-row column tiles of the GEMM (rows beyond the matrix are zero)
    L.kw = d.precision == STEGO_VIT_F16X3 ? 32 : 64;
    auto panels = [&](int rows, int K) { return (size_t)wpanels(rows) * (K / L.kw) * VP_BYTES; };
    L.patch_w = take(panels(d.D, L.Kp));
    L.patch_b = take((size_t)d.D * 4);
    L.cls = take((size_t)d.D * 4);
    L.pos = take((size_t)L.ntok * d.D * 4);
    L.blk0 = o;
    const size_t start = o;
    L.rel.ln1_w = take((size_t)d.D * 4) - start;
    L.rel.ln1_b = take((size_t)d.D * 4) - start;
    L.rel.qkv_w = take(panels(3 * d.D, d.D)) - start;
    L.rel.qkv_b = take((size_t)3 * d.D * 4) - start;
    L.rel.proj_w = take(panels(d.D, d.D)) - start;
    L.rel.proj_b = take((size_t)d.D * 4) - start;
    L.rel.ln2_w = take((size_t)d.D * 4) - start;
    L.rel.ln2_b = take((size_t)d.D * 4) - start;
    L.rel.fc1_w = take(panels(d.hidden, d.D)) - start;
    L.rel.fc1_b = take((size_t)d.hidden * 4) - start;
    L.rel.fc2_w = take(panels(d.D, d.hidden)) - start;
    L.rel.fc2_b = take((size_t)d.D * 4) - start;
    L.blk_stride = o - start;
    o = start + L.blk_stride * (size_t)d.depth;
    L.norm_w = take((size_t)d.D * 4);
    L.norm_b = take((size_t)d.D * 4);
    L.amax = take((size_t)(1 + 4 * d.depth) * 4);
    L.total = o;
    return L;
}

struct Workspace {
    int M, mb, ntok_pad;
    size_t resid, xa, ya, ha, q, k, vt, qkv_bytes, plane, total;      // plane: halves between the hi and the lo plane of q / k / vt (F16X3)
};

Workspace make_workspace(const StegoVitDesc& d, const Layout& L)
{
    Workspace w;
    w.M = d.B * L.ntok;
    w.mb = apanels(w.M);
    w.ntok_pad = (L.ntok + 63) / 64 * 64;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o = up256(o + bytes); return at; };
    w.resid = take((size_t)w.M * d.D * 4);
    w.xa = take((size_t)w.mb * (d.D / L.kw) * VP_BYTES);
    w.ya = take((size_t)w.mb * (d.D / L.kw) * VP_BYTES);
    const size_t ha_bytes = (size_t)w.mb * (d.hidden / L.kw) * VP_BYTES;
    const size_t im_bytes = (size_t)apanels(d.B * L.hw) * (L.Kp / L.kw) * VP_BYTES;     // im2col panels alias the MLP buffer
    w.ha = take(ha_bytes > im_bytes ? ha_bytes : im_bytes);
    const size_t one = up256((size_t)d.B * d.heads * w.ntok_pad * 64 * 2);
    const int npl = d.precision == STEGO_VIT_F16X3 ? 2 : 1;
    w.plane = one / 2;
    w.q = take(one * npl);
    w.k = take(one * npl);
    w.vt = take(one * npl);
    w.qkv_bytes = o - w.q;
    w.total = o;
    return w;
}

int check_desc(const StegoVitDesc* d)
{
    if (!d) return STEGO_ERR_NULL;
    if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->D <= 0 || d->depth <= 0 || d->heads <= 0 || d->hidden <= 0) return STEGO_ERR_SHAPE;
    if (d->patch != 8 && d->patch != 16) return STEGO_ERR_UNSUPPORTED;
    if (d->precision != STEGO_VIT_F16 && d->precision != STEGO_VIT_F16X3) return STEGO_ERR_UNSUPPORTED;
    if (d->H % d->patch || d->W % d->patch || d->W % 8) return STEGO_ERR_SHAPE;
    if (d->D % 64 || d->D > 768 || d->hidden % 64 || d->heads * 64 != d->D) return STEGO_ERR_UNSUPPORTED;
    const long long ntok = (long long)(d->H / d->patch) * (d->W / d->patch) + 1;
    if ((long long)d->B * ntok >= (1ll << 20)) return STEGO_ERR_UNSUPPORTED;      // row index arithmetic (float reciprocal)
    return STEGO_OK;
}

template <int EPI, bool X3> hipError_t launch_gemm_p(const GemmParams& p, hipStream_t stream)
{
    hipError_t ea = stego::ensure_dynamic_lds(reinterpret_cast<const void*>(&vit_gemm_kernel<EPI, X3>), GT_LDS);
    if (ea != hipSuccess) return ea;
    const int mtiles = (p.M + GT_M - 1) / GT_M, ntiles = ntiles192(p.N);
    const int grid = 8 * ((mtiles + 7) / 8) * ntiles;
    hipLaunchKernelGGL((vit_gemm_kernel<EPI, X3>), dim3(grid), dim3(GT_THREADS), GT_LDS, stream, p, mtiles, ntiles);
    return hipGetLastError();
}

template <int EPI> hipError_t launch_gemm(const GemmParams& p, bool x3, hipStream_t stream)
{
    return x3 ? launch_gemm_p<EPI, true>(p, stream) : launch_gemm_p<EPI, false>(p, stream);
}

// weight [R][K] -> panels; F16X3: its max |w| to *amax first (zeroed by the caller), the panels scaled by the power of two it implies
hipError_t pack(const float* src, int R, int K, unsigned char* dst, bool x3, unsigned* amax, hipStream_t stream)
{
    const int nrb = wpanels(R), nkc = K / (x3 ? 32 : 64);
    const long long total = (long long)nrb * 128 * nkc * 8;
    if (x3) {
        hipLaunchKernelGGL(vit_absmax_kernel, dim3(256), dim3(256), 0, stream, src, (long long)R * K, amax);
        hipLaunchKernelGGL((vit_pack_kernel<true>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, src, R, K, dst, nrb, nkc, amax);
    } else {
        hipLaunchKernelGGL((vit_pack_kernel<false>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, src, R, K, dst, nrb, nkc, amax);
    }
    return hipGetLastError();
}

}  // namespace vit
}  // namespace stego

using namespace stego;
using namespace stego::vit;

#define VIT_TRY(expr)                                        \
    do {                                                     \
        hipError_t e_ = (expr);                              \
        if (e_ != hipSuccess) return STEGO_ERR_HIP + (int)e_; \
    } while (0)

extern "C" {

int32_t stego_vit_param_count(const StegoVitDesc* d) { return d ? 4 + 12 * d->depth + 2 : 0; }

size_t stego_vit_weights_bytes(const StegoVitDesc* d) { return check_desc(d) == STEGO_OK ? make_layout(*d).total : 0; }

size_t stego_vit_workspace_bytes(const StegoVitDesc* d)
{
    if (check_desc(d) != STEGO_OK) return 0;
    const Layout L = make_layout(*d);
    return make_workspace(*d, L).total;
}

int stego_vit_pack_weights(const StegoVitDesc* d, const float* const* params, int32_t n_params, void* packed,
                           size_t packed_bytes, stego_stream_t stream_)
{
    const int rc = check_desc(d);
    if (rc != STEGO_OK) return rc;
    if (!params || !packed) return STEGO_ERR_NULL;
    if (n_params != stego_vit_param_count(d)) return STEGO_ERR_SHAPE;
    for (int i = 0; i < n_params; ++i)
        if (!params[i]) return STEGO_ERR_NULL;
    const Layout L = make_layout(*d);
    if (packed_bytes < L.total) return STEGO_ERR_WORKSPACE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    unsigned char* base = static_cast<unsigned char*>(packed);
    auto vec = [&](const float* src, size_t off, size_t count) {
        return hipMemcpyAsync(base + off, src, count * 4, hipMemcpyDeviceToDevice, stream);
    };
    const int D = d->D;
    const bool x3 = d->precision == STEGO_VIT_F16X3;
    unsigned* amax = reinterpret_cast<unsigned*>(base + L.amax);
    VIT_TRY(hipMemsetAsync(amax, 0, (size_t)(1 + 4 * d->depth) * 4, stream));
    int i = 0;
    VIT_TRY(pack(params[i++], D, L.Kp, base + L.patch_w, x3, amax, stream));
    VIT_TRY(vec(params[i++], L.patch_b, D));
    VIT_TRY(vec(params[i++], L.cls, D));
    VIT_TRY(vec(params[i++], L.pos, (size_t)L.ntok * D));
    for (int l = 0; l < d->depth; ++l) {
        const size_t s = L.blk0 + L.blk_stride * (size_t)l;
        VIT_TRY(vec(params[i++], s + L.rel.ln1_w, D));
        VIT_TRY(vec(params[i++], s + L.rel.ln1_b, D));
        VIT_TRY(pack(params[i++], 3 * D, D, base + s + L.rel.qkv_w, x3, amax + 1 + 4 * l, stream));
        VIT_TRY(vec(params[i++], s + L.rel.qkv_b, (size_t)3 * D));
        VIT_TRY(pack(params[i++], D, D, base + s + L.rel.proj_w, x3, amax + 2 + 4 * l, stream));
        VIT_TRY(vec(params[i++], s + L.rel.proj_b, D));
        VIT_TRY(vec(params[i++], s + L.rel.ln2_w, D));
        VIT_TRY(vec(params[i++], s + L.rel.ln2_b, D));
        VIT_TRY(pack(params[i++], d->hidden, D, base + s + L.rel.fc1_w, x3, amax + 3 + 4 * l, stream));
        VIT_TRY(vec(params[i++], s + L.rel.fc1_b, d->hidden));
        VIT_TRY(pack(params[i++], D, d->hidden, base + s + L.rel.fc2_w, x3, amax + 4 + 4 * l, stream));
        VIT_TRY(vec(params[i++], s + L.rel.fc2_b, D));
    }
    VIT_TRY(vec(params[i++], L.norm_w, D));
    VIT_TRY(vec(params[i++], L.norm_b, D));
    return STEGO_OK;
}

int stego_vit_forward(const StegoVitDesc* d, const void* packed, const float* img, float* tokens_out, void* workspace,
                      size_t workspace_bytes, stego_stream_t stream_)
{
    const int rc = check_desc(d);
    if (rc != STEGO_OK) return rc;
    if (!packed || !img || !tokens_out || !workspace) return STEGO_ERR_NULL;
    if ((reinterpret_cast<uintptr_t>(img) & 15) || (reinterpret_cast<uintptr_t>(workspace) & 255) ||
        (reinterpret_cast<uintptr_t>(packed) & 255))
        return STEGO_ERR_ALIGN;
    const Layout L = make_layout(*d);
    const Workspace w = make_workspace(*d, L);
    if (workspace_bytes < w.total) return STEGO_ERR_WORKSPACE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const unsigned char* wb = static_cast<const unsigned char*>(packed);
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    const int D = d->D, M = w.M;
    const bool x3 = d->precision == STEGO_VIT_F16X3;
    const int kw = L.kw;
    const unsigned* amax = reinterpret_cast<const unsigned*>(wb + L.amax);
    const float eps = 1e-6f;                                 // norm_layer = partial(nn.LayerNorm, eps=1e-6) (vision_transformer.py:243)
    auto fvec = [&](size_t off) { return reinterpret_cast<const float*>(wb + off); };
    float* resid = reinterpret_cast<float*>(ws + w.resid);

    // padding keys of V^T must be finite zeros (0 * garbage could be NaN); Q/K padding is cleared with them
    VIT_TRY(hipMemsetAsync(ws + w.q, 0, w.qkv_bytes, stream));

    GemmParams g{};
    g.debug = stego::knob(stego::KNOB_DEBUG_VIT);
    g.D = D;
    g.heads = d->heads;
    g.ntok = L.ntok;
    g.ntok_pad = w.ntok_pad;
    g.hw = L.hw;
    g.inv_ntok = 1.f / (float)L.ntok;
    g.inv_hw = 1.f / (float)L.hw;
    g.qscale = 0.125f * LOG2E;
    g.resid = resid;
    g.ldr = D;
    g.q = reinterpret_cast<half_t*>(ws + w.q);
    g.k = reinterpret_cast<half_t*>(ws + w.k);
    g.vt = reinterpret_cast<half_t*>(ws + w.vt);
    g.plane = w.plane;

    // ---- prepare_tokens
    {
        const int Mp = d->B * L.hw, nrb = apanels(Mp), nkc = L.Kp / kw;
        const long long total = (long long)nrb * 128 * nkc * 8;
        if (x3)
            hipLaunchKernelGGL((vit_im2col_kernel<true>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, img, d->B, d->H, d->W,
                               d->patch, ws + w.ha, nrb, nkc);
        else
            hipLaunchKernelGGL((vit_im2col_kernel<false>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, img, d->B, d->H, d->W,
                               d->patch, ws + w.ha, nrb, nkc);
        VIT_TRY(hipGetLastError());
        GemmParams e = g;
        e.A = ws + w.ha;
        e.W = wb + L.patch_w;
        e.bias = fvec(L.patch_b);
        e.M = Mp;
        e.N = D;
        e.nkc = nkc;
        e.pos = fvec(L.pos);
        e.wamax = amax;
        VIT_TRY(launch_gemm<EPI_EMBED>(e, x3, stream));
        hipLaunchKernelGGL(vit_cls_kernel, dim3((d->B * D + 255) / 256), dim3(256), 0, stream, fvec(L.cls), fvec(L.pos), resid, d->B,
                           D, L.ntok);
        VIT_TRY(hipGetLastError());
    }
    AttnParams a{};
    a.q = g.q;
    a.k = g.k;
    a.vt = g.vt;
    a.outp = ws + w.ya;
    a.out_nkc = D / kw;
    a.plane = w.plane;
    a.heads = d->heads;
    a.ntok = L.ntok;
    a.ntok_pad = w.ntok_pad;
    a.nqb = (L.ntok + 127) / 128;
    a.units = d->B * d->heads;
    const dim3 ln_grid((M + 3) / 4);
    for (int l = 0; l < d->depth; ++l) {
        const size_t s = L.blk0 + L.blk_stride * (size_t)l;
        if (x3)
            hipLaunchKernelGGL((vit_layernorm_kernel<true, true>), ln_grid, dim3(256), 0, stream, resid, fvec(s + L.rel.ln1_w),
                               fvec(s + L.rel.ln1_b), M, D, eps, ws + w.xa, (float*)nullptr);
        else
            hipLaunchKernelGGL((vit_layernorm_kernel<true, false>), ln_grid, dim3(256), 0, stream, resid, fvec(s + L.rel.ln1_w),
                               fvec(s + L.rel.ln1_b), M, D, eps, ws + w.xa, (float*)nullptr);
        VIT_TRY(hipGetLastError());
        GemmParams p1 = g;
        p1.A = ws + w.xa;
        p1.W = wb + s + L.rel.qkv_w;
        p1.bias = fvec(s + L.rel.qkv_b);
        p1.M = M;
        p1.N = 3 * D;
        p1.nkc = D / kw;
        p1.wamax = amax + 1 + 4 * l;
        VIT_TRY(launch_gemm<EPI_QKV>(p1, x3, stream));
        if (x3) hipLaunchKernelGGL((vit_attn_kernel<true>), dim3(8 * ((a.units + 7) / 8) * a.nqb), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((vit_attn_kernel<false>), dim3(8 * ((a.units + 7) / 8) * a.nqb), dim3(256), 0, stream, a);
        VIT_TRY(hipGetLastError());
        GemmParams p2 = g;
        p2.A = ws + w.ya;
        p2.W = wb + s + L.rel.proj_w;
        p2.bias = fvec(s + L.rel.proj_b);
        p2.M = M;
        p2.N = D;
        p2.nkc = D / kw;
        p2.wamax = amax + 2 + 4 * l;
        VIT_TRY(launch_gemm<EPI_RESID>(p2, x3, stream));
        if (x3)
            hipLaunchKernelGGL((vit_layernorm_kernel<true, true>), ln_grid, dim3(256), 0, stream, resid, fvec(s + L.rel.ln2_w),
                               fvec(s + L.rel.ln2_b), M, D, eps, ws + w.xa, (float*)nullptr);
        else
            hipLaunchKernelGGL((vit_layernorm_kernel<true, false>), ln_grid, dim3(256), 0, stream, resid, fvec(s + L.rel.ln2_w),
                               fvec(s + L.rel.ln2_b), M, D, eps, ws + w.xa, (float*)nullptr);
        VIT_TRY(hipGetLastError());
        GemmParams p3 = g;
        p3.A = ws + w.xa;
        p3.W = wb + s + L.rel.fc1_w;
        p3.bias = fvec(s + L.rel.fc1_b);
        p3.M = M;
        p3.N = d->hidden;
        p3.nkc = D / kw;
        p3.outp = ws + w.ha;
        p3.out_nkc = d->hidden / kw;
        p3.wamax = amax + 3 + 4 * l;
        VIT_TRY(launch_gemm<EPI_GELU>(p3, x3, stream));
        GemmParams p4 = g;
        p4.A = ws + w.ha;
        p4.W = wb + s + L.rel.fc2_w;
        p4.bias = fvec(s + L.rel.fc2_b);
        p4.M = M;
        p4.N = D;
        p4.nkc = d->hidden / kw;
        p4.wamax = amax + 4 + 4 * l;
        VIT_TRY(launch_gemm<EPI_RESID>(p4, x3, stream));
    }
    hipLaunchKernelGGL((vit_layernorm_kernel<false, false>), ln_grid, dim3(256), 0, stream, resid, fvec(L.norm_w), fvec(L.norm_b), M, D, eps,
                       (unsigned char*)nullptr, tokens_out);
    VIT_TRY(hipGetLastError());
    return STEGO_OK;
}

}  // extern "C"
