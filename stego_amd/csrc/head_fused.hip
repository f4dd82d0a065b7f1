// Segmentation head of DinoFeaturizer on gfx950 (include/stego_head.h): the producer of `code`, forward and backward.
//
// Reference: src/modules.py:108-116 (Dropout2d x 3, cluster1 = Conv2d(C, K, 1x1), cluster2 = Conv2d(C, C) -> ReLU -> Conv2d(C, K)) and
// what autograd derives for the six parameters.  On the channels-last token matrix the 1x1 convolutions are GEMMs:
//     code = (X * m1) W1^T + b1 + relu((X * m2) W21^T + b21) W22^T + b22          feats_out = X * m3
// with m* the per-(image, channel) scales of nn.Dropout2d.  Two kernels do the arithmetic:
//   * head_gemm_kernel<KVEC, EPI, NW>  C[M, N] = epilogue(A'[M, K] B[N, K]^T): a 128 x 128 NW tile per workgroup, 4 NW waves (64 x 64
//     each, v_mfma_f32_32x32x16_f16).  A (activations) is read as fp32, scaled by the dropout mask of its image (kept in LDS) and a
//     power-of-two prescale, split into fp16 hi + lo and staged in LDS in 32-channel stages; B (weights) arrives pre-split - fp16 hi / lo
//     planes written once per call by head_prep_absmax_kernel / head_prep_split_kernel - so staging it is a copy.  hi*hi + hi*lo + lo*hi,
//     fp32 accumulate: the fp32-class scheme of the loss kernels.  Stages are double-buffered (one barrier per stage); the token operand
//     is fetched two stages ahead, the weight units one.  The MFMAs compute TRANSPOSED 32 x 32 blocks (operands swapped), so that a lane
//     holds runs of four consecutive columns of one output row: the epilogues (+ bias (+ bias), + bias -> ReLU, += C, * 1[aux > 0])
//     use 16-byte global accesses.  The cluster1 GEMM also writes feats_out = X * m3 from the registers it stages (modules.py:116).
//   * head_wgrad_kernel  dW[N, C] = sum_t G'[t, N]^T X'[t, C] over a range of tokens (split over workgroups; partial tiles + a
//     reduction kernel, fixed order: bitwise repeatable): both operands are transposed on their way into LDS (the reduction index -
//     the token - must be contiguous per MFMA lane), the dropout mask rides on X'; the first channel tile also sums the columns of
//     G' (the bias gradients).
// plus small ones: operand magnitudes (head_absmax_kernel; or the producing GEMM's epilogue), the constants of a call (scale words, a
// row of ones / zeros standing in for absent masks / biases: kernels never test a pointer), the cached-token gather.
// Forward: cluster1; cluster2[0] + ReLU -> H (NW = 3: the token operand staged once for all 384 outputs); cluster2[2] accumulated into
// code.  Backward: dHpre = (G W22) * 1[H > 0] (NW = 3), three weight-gradient GEMMs, three reductions.
// What bounds them: the ~11 bytes per cycle a CU is delivered under full-chip load, L2 hits included - not HBM, not
// the matrix cores (22 % busy); and what the compiler does to a staging loop unless told otherwise (comments at the loops).
#include <hip/hip_runtime.h>
#include <type_traits>
#include <algorithm>

#include "../../include/stego_head.h"
#include "corr_common.h"
#include "host_util.h"

namespace stego {

typedef unsigned int hu32x4 __attribute__((ext_vector_type(4)));

constexpr int HT = 128;                          // tile rows / cols
constexpr int HKS = 32;                          // channels (or tokens) per stage
constexpr int HSIDE = 16384;                     // one operand stage: [hi 128 x 64 B][lo 128 x 64 B]
constexpr int HS_WORDS = 8;                      // operand-scale words of a call (enum HS_* below)

__device__ __forceinline__ int hswz(int r, int u) { return r * 64 + ((u ^ ((r >> 2) & 3)) << 4); }

// one 32-deep stage: a.b ~= ah.bh + ah.bl + al.bh   (rows of As = output rows, rows of Bs = output columns).  ALO / BLO: byte offset
// of the lo plane behind the hi plane (64 B per row: 8192 for a 128-row operand stage, 24576 for a 384-row one).
// The MFMA is issued with the operands SWAPPED (it computes the transposed 32 x 32 block): a lane then holds, for ONE output row
// (lane & 31), the columns 8 j + 4 (lane >> 5) + 0..3, j = 0..3 - four runs of four consecutive columns, i.e. 16-byte global accesses in
// the epilogues (4 x fewer store / load instructions than the natural layout's one dword per lane and register: the epilogue of a
// 128 x 384 tile was 768 store instructions per workgroup and store-issue-bound).
template <int ALO, int BLO>
__device__ __forceinline__ void head_mma_stage(const unsigned char* __restrict__ As, const unsigned char* __restrict__ Bs,
                                               f32x16 (&acc)[2][2], int lane, int wr, int wc)
{
    const int r = lane & 31, half = lane >> 5;
    const int ra0 = 64 * wr + r, ra1 = ra0 + 32, rb0 = 64 * wc + r, rb1 = rb0 + 32;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int u = 2 * ks + half;
        const f16x8 ah0 = *reinterpret_cast<const f16x8*>(As + hswz(ra0, u)), al0 = *reinterpret_cast<const f16x8*>(As + ALO + hswz(ra0, u));
        const f16x8 ah1 = *reinterpret_cast<const f16x8*>(As + hswz(ra1, u)), al1 = *reinterpret_cast<const f16x8*>(As + ALO + hswz(ra1, u));
        const f16x8 bh0 = *reinterpret_cast<const f16x8*>(Bs + hswz(rb0, u)), bl0 = *reinterpret_cast<const f16x8*>(Bs + BLO + hswz(rb0, u));
        const f16x8 bh1 = *reinterpret_cast<const f16x8*>(Bs + hswz(rb1, u)), bl1 = *reinterpret_cast<const f16x8*>(Bs + BLO + hswz(rb1, u));
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh0, al0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh1, al0, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh0, al1, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh1, al1, acc[1][1], 0, 0, 0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl0, ah0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl1, ah0, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl0, ah1, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl1, ah1, acc[1][1], 0, 0, 0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh0, ah0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh1, ah0, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh0, ah1, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh1, ah1, acc[1][1], 0, 0, 0);
    }
}

// 16 consecutive values of one row -> its two 16-byte units (8 values each) of the hi and of the lo plane
__device__ __forceinline__ void head_commit16(unsigned char* stage, int row, int h, const float (&v)[16])
{
    hu32x4 hi[2], lo[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            unsigned a, b;
            split_f16_pair(v[8 * u + 2 * e], v[8 * u + 2 * e + 1], a, b);
            hi[u][e] = a;
            lo[u][e] = b;
        }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        *reinterpret_cast<hu32x4*>(stage + hswz(row, 2 * h + u)) = hi[u];
        *reinterpret_cast<hu32x4*>(stage + 8192 + hswz(row, 2 * h + u)) = lo[u];
    }
}

enum { EPI_BIAS = 0, EPI_BIAS_RELU = 1, EPI_ACCUM = 2, EPI_MASK_POS = 3 };

// Power-of-two prescale of an operand from the bits of its largest magnitude (a device word written by head_absmax_kernel or by a
// producing epilogue): max |x| s lands in [2^13, 2^14), so that every value down to 2^-17 of the largest keeps both fp16 halves in
// the normal range (22 bits); smaller ones lose bits only relative to themselves, not to the sums they enter.  Exact to undo.
__device__ __forceinline__ float head_scale(const unsigned* amax)
{
    if (!amax) return 1.f;
    const float mx = __builtin_bit_cast(float, *amax);
    if (!(mx > 0.f) || !(mx < 3.0e38f)) return 1.f;
    return __builtin_ldexpf(1.f, 14 - __builtin_amdgcn_frexp_expf(mx));
}

// Raises *out to the workgroup's largest value: one atomic per WORKGROUP at most, and none when the word already holds a larger
// value (thousands of waves on one address serialise at ~12 ns per atomic: 100 us for a 77 MB pass that reads for 15).
__device__ __forceinline__ void head_publish_max(float mx, unsigned* out, float* scratch /* LDS, >= waves floats */)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) scratch[wave] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        float v = scratch[0];
        for (int w = 1; w < nw; ++w) v = fmaxf(v, scratch[w]);
        const unsigned bits = __builtin_bit_cast(unsigned, v);
        if (bits > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, bits);
    }
}

// largest |x| of a strided [rows, cols] matrix -> *out (float bits of a non-negative value order like unsigned integers)
__global__ void __launch_bounds__(256) head_absmax_kernel(const float* x, long long img_stride, long long row_stride, int rows_per_img,
                                                          long long rows, int cols, unsigned* out)
{
    float mx = 0.f;
    const bool vec = (cols & 3) == 0 && ((img_stride | row_stride) & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
                     rows * (cols >> 2) < (1ll << 31);
    if (vec) {
        // flat index over the 16-byte units of the matrix, four loads in flight per thread
        const int c4 = cols >> 2, total = (int)(rows * c4), step = (int)gridDim.x * 256;
        auto at = [&](int q) {
            const int qq = q < total ? q : total - 1;
            const int r = qq / c4, c = qq - r * c4;
            const int b = r / rows_per_img, t = r - b * rows_per_img;
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + (long long)b * img_stride + (long long)t * row_stride + 4 * c);
            return fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
        };
        for (int q = (int)blockIdx.x * 256 + threadIdx.x; q < total; q += 4 * step) {
            const float a0 = at(q), a1 = at(q + step), a2 = at(q + 2 * step), a3 = at(q + 3 * step);       // (clamped: re-reads the last unit)
            mx = fmaxf(mx, fmaxf(fmaxf(a0, a1), fmaxf(a2, a3)));
        }
    } else {
        const long long total = rows * cols;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
            const long long r = i / cols;
            const int c = (int)(i - r * cols);
            const long long bb = r / rows_per_img, t = r - bb * rows_per_img;
            mx = fmaxf(mx, fabsf(x[bb * img_stride + t * row_stride + c]));
        }
    }
    __shared__ float red[4];
    head_publish_max(mx, out, red);
}

struct HeadGemmParams {
    const float* A;            // row m = (image b, token t): A + b * a_img + t * a_tok, K contiguous
    long long a_img, a_tok;
    int HW;                    // rows per image
    const float* maskA;        // A'[m][k] = A[m][k] * maskA[b * mask_ld + k]: [B, K] (mask_ld = K) or ONE row of ones (mask_ld = 0); never null
    int mask_ld, mask3_ld;
    int n_slot, n_img;         // images a 128-row tile can touch ((HT - 1) / HW + 2, at most B); B
    const half_t* Bh;          // weights, pre-split by head_prep_weights_kernel: [Npad][ldb] fp16 hi / lo planes of B[n][k] * scale,
    const half_t* Bl;          // zero beyond N / K (Npad a multiple of the workgroup's columns, ldb a multiple of 32)
    int ldb;
    float* C;                  // [M, N] row-major (ldc)
    int ldc;
    const float* bias;         // [N]; never null in the bias epilogues (a row of zeros when absent)
    const float* bias2;        // [N], likewise
    const float* aux;          // EPI_MASK_POS: [M, N] (ldaux)
    int ldaux;
    float* feats_out;          // [M, K] dense or null: = A * mask3 (written by the N tile 0)
    const float* mask3;        // like maskA, for feats_out (EPI_BIAS only)
    const unsigned* amax_a;    // largest |A| / |B| (device words, see head_scale) or null
    const unsigned* amax_b;
    unsigned* amax_out;        // or null: atomicMax of |C| (the operand scale of whoever consumes C)
    int M, N, K, epi;
};

// Weights -> the B operand's fp16 hi / lo planes, once per call, two small launches over all weights of the call: the largest |w| of
// each (its bits go to *amax: the GEMMs undo the scale from the same word), then B[n][k] = w * 2^s split into hi + lo.
struct HeadPrepJob {
    const float* src;          // [rows][cols] row-major (ld); transposed: B[n][k] = src[k * ld + n]
    int N, K, ld, transposed;
    half_t* hi;
    half_t* lo;
    int Npad, Kpad;
    unsigned* amax;
};
struct HeadPrepParams { HeadPrepJob job[4]; };

// One launch for the small constants of a call (three memset / copy nodes cost ~5 us each on the stream): the scale words (zeroed, or
// the first n_copy of them copied from the forward's), a row of zeros (may be null) and a row of ones.
__global__ void __launch_bounds__(256) head_consts_kernel(unsigned* words, const unsigned* copy_from, int n_copy, float* zeros, float* ones, int C)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    // (copy_from with n_copy = -1: only word 0 - the tokens' magnitude, handed in by the caller - is copied)
    if (i < HS_WORDS) words[i] = (copy_from && (n_copy < 0 ? i == 0 : i < n_copy)) ? copy_from[i] : 0u;
    if (i < C) {
        if (zeros) zeros[i] = 0.f;
        ones[i] = 1.f;
    }
}

constexpr int PREP_WG = 32;                      // workgroups per job (grid.y)

// pass 1: the largest |w| of every job -> its amax word (zeroed by the caller)
__global__ void __launch_bounds__(256) head_prep_absmax_kernel(const HeadPrepParams prm)
{
    __shared__ float red[4];
    const HeadPrepJob jb = prm.job[blockIdx.x];
    const int rows = jb.transposed ? jb.K : jb.N, cols = jb.transposed ? jb.N : jb.K;
    float mx = 0.f;
    for (int i = blockIdx.y * 256 + threadIdx.x; i < rows * cols; i += PREP_WG * 256) {
        const int r = i / cols, c = i - r * cols;
        mx = fmaxf(mx, fabsf(jb.src[(size_t)r * jb.ld + c]));
    }
    head_publish_max(mx, jb.amax, red);
}

// pass 2: B[n][k] = w * 2^s split into hi + lo, zero-padded to [Npad][Kpad]
__global__ void __launch_bounds__(256) head_prep_split_kernel(const HeadPrepParams prm)
{
    const HeadPrepJob jb = prm.job[blockIdx.x];
    const float sc = head_scale(jb.amax);
    const int kp2 = jb.Kpad / 2;
    for (int i = blockIdx.y * 256 + threadIdx.x; i < jb.Npad * kp2; i += PREP_WG * 256) {
        const int n = i / kp2, k = 2 * (i - n * kp2);
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const bool ok = n < jb.N && k + e < jb.K;
            const int nn = ok ? n : 0, kk = ok ? k + e : 0;
            const float w = jb.src[jb.transposed ? (size_t)kk * jb.ld + nn : (size_t)nn * jb.ld + kk];
            v[e] = w * (ok ? sc : 0.f);
        }
        unsigned h, l;
        split_f16_pair(v[0], v[1], h, l);
        *reinterpret_cast<unsigned*>(jb.hi + (size_t)n * jb.Kpad + k) = h;
        *reinterpret_cast<unsigned*>(jb.lo + (size_t)n * jb.Kpad + k) = l;
    }
}

// KVEC: K % 32 == 0 and 16-byte aligned A rows -> float4 staging; otherwise scalar staging with bounds (the K = cfg.dim GEMM).
// NW: 128-column blocks per workgroup (4 NW waves: a 128 x 128 NW tile).  NW = 3 covers the C = 384 outputs of a ViT-S head with ONE
// staging of the token operand (the activation side is what costs: 16 values to scale and split per thread and stage; the weights
// arrive split).
template <bool KVEC, int EPI, int NW>
__global__ void __launch_bounds__(256 * NW) head_gemm_kernel(const HeadGemmParams prm)
{
    constexpr int BLO = NW * 8192;               // the B stage: [hi 128 NW rows x 64 B][lo ...]
    constexpr int STAGE = HSIDE + 2 * BLO;       // one stage: A, then B; two of them (double buffer: one barrier per stage)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave & 1, wc = wave >> 1;
    const int m0 = blockIdx.x * HT, n0 = blockIdx.y * HT * NW;
    const int row = tid >> 1, h = tid & 1;       // B row (an output column of this workgroup), which 16 channels of the stage
    const bool stage_a = NW == 1 || tid < 256;   // (wave-uniform) the first 256 threads also stage the A rows
    // my A row (rows beyond the matrix are read from the last valid one and zeroed: every load below is unconditional - a load inside
    // a branch gets its own s_waitcnt and the staging becomes a chain of dependent round trips)
    const int arow_i = row & (HT - 1);
    const int m = m0 + arow_i;
    const bool m_ok = m < prm.M;
    const int mc = m_ok ? m : prm.M - 1;
    const int b = mc / prm.HW, t = mc - b * prm.HW;
    const float* arow = prm.A + (long long)b * prm.a_img + (long long)t * prm.a_tok;
    // (maskA / mask3 always point at [B, K] floats - a vector of ones when there is no dropout: a load that happens only when a pointer is
    // set becomes a branch with its own s_waitcnt, and a wait in front of the MFMAs is a stage that no longer overlaps its loads)
    float* frow = (EPI == EPI_BIAS && prm.feats_out && blockIdx.y == 0 && m_ok && stage_a) ? prm.feats_out + (size_t)m * prm.K : nullptr;
    const float a_keep = m_ok ? 1.f : 0.f;
    const half_t* bh_row = prm.Bh + (size_t)(n0 + row) * prm.ldb + 16 * h;
    const half_t* bl_row = prm.Bl + (size_t)(n0 + row) * prm.ldb + 16 * h;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const float sa = head_scale(prm.amax_a), sb = head_scale(prm.amax_b);
    const float unscale = 1.f / (sa * sb);
    // The dropout masks of the images this tile touches live in LDS (a few KB, copied once): the commit reads them from there instead of
    // holding 16 + 16 more registers of prefetched mask values per thread.
    float* mlds = reinterpret_cast<float*>(lds + 2 * STAGE);                 // [n_slot][K] (maskA), then [n_slot][K] (mask3, EPI_BIAS)
    const int b_first = m0 / prm.HW;
    if constexpr (KVEC) {
        const int nm = prm.n_slot * prm.K;
        for (int i = tid; i < nm; i += 256 * NW) {
            const int slot = i / prm.K, k = i - slot * prm.K;
            const int bs = min(b_first + slot, prm.n_img - 1);
            mlds[i] = prm.maskA[(size_t)bs * prm.mask_ld + k];
            if constexpr (EPI == EPI_BIAS) mlds[nm + i] = prm.mask3[(size_t)bs * prm.mask3_ld + k];
        }
    }
    const float* ml_row = mlds + (b - b_first) * prm.K;
    const float* ml3_row = ml_row + prm.n_slot * prm.K;

    // load_a / load_b: nothing but loads into registers (no arithmetic, no stores: whatever uses a loaded value makes the compiler wait
    // for it, and this runs right before the MFMAs of the stage in LDS); commit_stage turns them into a stage's operands.  The token
    // operand streams from HBM (latency ~3 us under load, measured as the stage time with a one-stage look-ahead): it is fetched TWO
    // stages ahead (xa[2]); the weight planes come from L2 and are fetched one stage ahead.
    f32x4 xa[2][4];
    hu32x4 vbh[2], vbl[2];
    auto load_b = [&](int k0) {
        vbh[0] = *reinterpret_cast<const hu32x4*>(bh_row + k0);
        vbh[1] = *reinterpret_cast<const hu32x4*>(bh_row + k0 + 8);
        vbl[0] = *reinterpret_cast<const hu32x4*>(bl_row + k0);
        vbl[1] = *reinterpret_cast<const hu32x4*>(bl_row + k0 + 8);
    };
    auto load_a = [&](int k0, f32x4 (&x)[4]) {
        const int kk = k0 + 16 * h;
        if constexpr (KVEC) {
#pragma unroll
            for (int q = 0; q < 4; ++q) x[q] = *reinterpret_cast<const f32x4*>(arow + kk + 4 * q);
        } else {
            // (the K = cfg.dim operand: no mask on this path; clamped addresses, zeroed by a 0 / 1 factor at commit)
            if ((prm.K & 1) == 0 && ((prm.a_img | prm.a_tok) & 1) == 0) {           // 8-byte aligned rows: pairs
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const int kc = kk + e < prm.K ? kk + e : prm.K - 2;
                    const f32x2 v = *reinterpret_cast<const f32x2*>(arow + kc);
                    x[e >> 2][e & 3] = v[0];
                    x[e >> 2][(e & 3) + 1] = v[1];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) x[e >> 2][e & 3] = arow[kk + e < prm.K ? kk + e : prm.K - 1];
            }
        }
    };

    auto commit_stage = [&](unsigned char* st, int k0, const f32x4 (&x)[4], auto with_a) {
        if constexpr (decltype(with_a)::value) {
            const int kk = k0 + 16 * h;
            float va[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 f = f32x4{1.f, 1.f, 1.f, 1.f};
                if constexpr (KVEC) f = *reinterpret_cast<const f32x4*>(ml_row + kk + 4 * q);
                if constexpr (EPI == EPI_BIAS) {
                    if (frow) *reinterpret_cast<f32x4*>(frow + kk + 4 * q) = x[q] * *reinterpret_cast<const f32x4*>(ml3_row + kk + 4 * q);   // feats_out = x * m3 (modules.py:116)
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float g = f[e] * (sa * a_keep);
                    if constexpr (!KVEC) g *= (kk + 4 * q + e < prm.K) ? 1.f : 0.f;
                    va[4 * q + e] = x[q][e] * g;
                }
            }
            head_commit16(st, arow_i, h, va);
        }
        unsigned char* Bs = st + HSIDE;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            *reinterpret_cast<hu32x4*>(Bs + hswz(row, 2 * h + u)) = vbh[u];
            *reinterpret_cast<hu32x4*>(Bs + BLO + hswz(row, 2 * h + u)) = vbl[u];
        }
    };
    // stage k's MFMAs read buffer k & 1 while stage k + 1 is committed to the other buffer: ONE barrier per stage, and the waves drift
    // apart - one wave's split / LDS writes run under the other waves' MFMAs.  The loop exists twice - for the waves that also stage
    // the token operand and for those that only copy weight units - so that each copy issues a fixed number of loads per iteration: with
    // the A loads inside a branch the compiler has to wait for ALL loads (vmcnt(0)) before the commit, look-ahead included.
    // Every workgroup walks the K stages in the same cyclic order but starts at its own stage (blockIdx.x mod stages): at any time the
    // tiles of the launch read different 128-byte columns of the token rows.  In step, they all hit the same residues of the
    // 1536-byte row pitch - a fraction of the memory channels at a time.
    const int k_end = (prm.K + HKS - 1) / HKS * HKS;
    const int n_st = k_end / HKS;
    const int st0 = (int)(blockIdx.x % (unsigned)n_st);
    auto stage_k = [&](int i) { const int j = st0 + min(i, n_st - 1); return (j >= n_st ? j - n_st : j) * HKS; };      // i-th stage of my order (clamped)
    auto run = [&](auto with_a) {
        constexpr bool WITH_A = decltype(with_a)::value;
        load_b(stage_k(0));
        if constexpr (WITH_A) {
            load_a(stage_k(0), xa[0]);
            load_a(stage_k(1), xa[1]);
        }
        __syncthreads();                             // the masks are in LDS
        commit_stage(lds, stage_k(0), xa[0], with_a);
        __syncthreads();
        // iteration i (its stage is in LDS buffer CUR): fetch B of stage i + 1 and A of stage i + 2, run the MFMAs, commit stage i + 1
        auto iteration = [&](int i, auto cur_c) {
            constexpr int CUR = decltype(cur_c)::value;
            const bool more = i + 1 < n_st;
            // (unconditional, from clamped stage numbers: a load inside a branch makes the number of loads in flight path-dependent,
            // and the compiler then waits for all of them before the commit)
            load_b(stage_k(i + 1));
            __builtin_amdgcn_sched_barrier(0);       // B's loads first: the commit waits for them with the look-ahead A loads still in flight
            if constexpr (WITH_A) load_a(stage_k(i + 2), xa[CUR]);                      // (xa[CUR] held stage i: committed one iteration ago)
            __builtin_amdgcn_sched_barrier(0);
            head_mma_stage<8192, BLO>(lds + CUR * STAGE, lds + CUR * STAGE + HSIDE, acc, lane, wr, wc);
            if (more) commit_stage(lds + (CUR ^ 1) * STAGE, stage_k(i + 1), xa[CUR ^ 1], with_a);
            __syncthreads();
        };
        for (int i = 0; i < n_st; i += 2) {
            iteration(i, std::integral_constant<int, 0>{});
            if (i + 1 < n_st) iteration(i + 1, std::integral_constant<int, 1>{});
        }
    };
    if (NW == 1 || stage_a) run(std::true_type{});
    else run(std::false_type{});

    // epilogue straight from the accumulators (transposed blocks, see head_mma_stage): lane & 31 is the row, register 4 j + e the column
    // 8 j + 4 (lane >> 5) + e of a 32 x 32 block.  Loads (accumulate / mask modes) are unconditional from clamped addresses, issued for
    // a whole block pair before their first use; 16-byte accesses wherever the four columns exist (rows of C need not be 16-byte aligned:
    // the K = 70 code rows are not; global accesses only need dword alignment), scalar ones at the ragged edge.
    float omax = 0.f;
    const int rloc = lane & 31, hsel = lane >> 5;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int rr = m0 + 64 * wr + 32 * mi + rloc;
        const bool r_ok = rr < prm.M;
        const unsigned rrc = (unsigned)(r_ok ? rr : prm.M - 1);
        // (32-bit element offsets from a uniform base: one address register per access instead of two; host-checked range)
        f32x4 old[2][4];
        if (EPI == EPI_ACCUM || EPI == EPI_MASK_POS) {
            const float* src = EPI == EPI_ACCUM ? prm.C : prm.aux;
            const unsigned ld = EPI == EPI_ACCUM ? (unsigned)prm.ldc : (unsigned)prm.ldaux;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int col = n0 + 64 * wc + 32 * ni + 8 * jj + 4 * hsel;
                    if (col + 3 < prm.N) {
                        // (one 16-byte load, alignment 4; it cannot sink below the first store to C, which may alias it)
                        __builtin_memcpy(&old[ni][jj], src + (rrc * ld + (unsigned)col), 16);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) old[ni][jj][e] = *reinterpret_cast<const volatile float*>(src + (rrc * ld + (unsigned)min(col + e, prm.N - 1)));
                    }
                }
        }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int col = n0 + 64 * wc + 32 * ni + 8 * jj + 4 * hsel;
                f32x4 v, bs4 = f32x4{0.f, 0.f, 0.f, 0.f};
                if (EPI == EPI_BIAS || EPI == EPI_BIAS_RELU) {
                    // (bias / bias2 always point at >= N floats, a row of zeros when absent; the host pads nothing: clamp at the edge)
                    if (col + 3 < prm.N) {
                        f32x4 t;
                        __builtin_memcpy(&t, prm.bias + col, 16); bs4 = t;
                        __builtin_memcpy(&t, prm.bias2 + col, 16); bs4 += t;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) bs4[e] = prm.bias[min(col + e, prm.N - 1)] + prm.bias2[min(col + e, prm.N - 1)];
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = acc[mi][ni][4 * jj + e] * unscale;
                    if (EPI == EPI_BIAS || EPI == EPI_BIAS_RELU) {
                        x += bs4[e];
                        if (EPI == EPI_BIAS_RELU) x = fmaxf(x, 0.f);
                    } else if (EPI == EPI_ACCUM) x += old[ni][jj][e];
                    else x *= old[ni][jj][e] > 0.f ? 1.f : 0.f;
                    v[e] = x;
                }
                float* dst = prm.C + (rrc * (unsigned)prm.ldc + (unsigned)col);
                if (r_ok && col + 3 < prm.N) {
                    __builtin_memcpy(dst, &v, 16);                 // (alignment 4: one global_store_dwordx4)
                    omax = fmaxf(omax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
                } else if (r_ok) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col + e < prm.N) { dst[e] = v[e]; omax = fmaxf(omax, fabsf(v[e])); }
                }
            }
    }
    if (prm.amax_out) {                          // (workgroup-uniform)
        __syncthreads();                         // the stage buffers are dead: their first bytes serve as scratch
        head_publish_max(omax, prm.amax_out, reinterpret_cast<float*>(lds));
    }
}

// ------------------------------------------------------------------------------------------------ weight gradients
struct HeadWgradParams {
    const float* G;            // [M, N] row-major (ldg): upstream of the GEMM's output
    int ldg;
    const float* X;            // token operand: (b, t) at X + b * x_img + t * x_tok, Kc contiguous
    long long x_img, x_tok;
    int HW;
    const float* maskX;        // X'[t][c] = X[t][c] * maskX[b * mask_ld + c]: [B, Kc] (mask_ld = Kc) or one row of ones (mask_ld = 0); never null
    int mask_ld;
    float* part;               // [splits][N][Kc] partial sums
    float* part_bias;          // [splits][N] partial column sums of G (written by channel tile 0) or null
    const unsigned* amax_g;    // largest |G| / |X| (see head_scale) or null
    const unsigned* amax_x;
    int M, N, Kc, splits;
};

// Version 2 of the weight-gradient kernel (the default).  The reduction index of dW = G'^T X' is the token, and an MFMA lane wants 8
// CONSECUTIVE tokens of one column - the transpose of how the operands lie in memory ([token][column] rows).  Version 1 (below, kept
// for reference and for token strides that are not 16-byte multiples) transposes at the global load: a thread reads one column of 16
// tokens with sixteen 4-byte loads (coalesced across the wave, 256 bytes per instruction).  This one loads rows as they lie - a
// thread reads 16 bytes (four columns of one token): a quarter of the load instructions - stages the fp16 hi / lo planes ROW-major
// ([token][column], 320-byte rows: 128 columns + padding) and lets the LDS transpose: ds_read_b64_tr_b16 (gfx950) hands lane q of a
// 16-lane group column q of a [4 tokens][16 columns] block, i.e. four consecutive tokens of one column; two of them are an MFMA
// fragment.  The padding makes the eight 8-byte row pieces of the two groups of a half-wave hit disjoint banks (row r of group g at
// word 80 r + 8 g + 2 c).  (Lane p of a group passes the address of token p >> 2, columns 4 (p & 3): measured with
// tools/ubench/tr_read.hip.)
typedef __fp16 fp16x4v __attribute__((__vector_size__(4 * sizeof(__fp16))));
constexpr int WG_SR = 320;                       // bytes per token row of a staged plane
constexpr int WG_PLANE = HKS * WG_SR;            // 10240
constexpr int WG_SIDE = 2 * WG_PLANE;            // hi, lo
constexpr int WG_STAGE = 2 * WG_SIDE;            // G, X

// this lane's A / B fragment of v_mfma_f32_32x32x16_f16: row (output index) colbase + (lane & 31), k = tokens 16 ks + 8 (lane >> 5) + 0..7
__device__ __forceinline__ f16x8 wg_frag(const unsigned char* plane, int ks, int colbase, int lane)
{
    const int grp = lane >> 4, p = lane & 15;
    const unsigned char* a = plane + (16 * ks + 8 * (grp >> 1) + (p >> 2)) * WG_SR + (colbase + 16 * (grp & 1) + 4 * (p & 3)) * 2;
    typedef __attribute__((address_space(3))) fp16x4v* lds_ptr;
    const fp16x4v t0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_ptr)(a));
    const fp16x4v t1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_ptr)(a + 4 * WG_SR));
    f16x8 r;
    __builtin_memcpy(&r, &t0, 8);
    __builtin_memcpy(reinterpret_cast<unsigned char*>(&r) + 8, &t1, 8);
    return r;
}

__global__ void __launch_bounds__(256) head_wgrad_tr_kernel(const HeadWgradParams prm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];      // two stages of [G hi | G lo | X hi | X lo]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int n0 = blockIdx.x * HT, c0 = blockIdx.y * HT, split = blockIdx.z;
    // token range of this split: whole stages of 32 tokens
    const int n_stage = (prm.M + HKS - 1) / HKS;
    const int s_beg = (int)((long long)n_stage * split / prm.splits), s_end = (int)((long long)n_stage * (split + 1) / prm.splits);
    // my four columns (of G: n, of X: channels) and my token rows tr + 8 q of the stage
    const int c4 = tid & 31, trow = tid >> 5;
    const int ng = n0 + 4 * c4, cx = c0 + 4 * c4;
    const bool gvec = (prm.ldg & 3) == 0 && (reinterpret_cast<uintptr_t>(prm.G) & 15) == 0 && prm.N % 4 == 0;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 bsum = f32x4{0.f, 0.f, 0.f, 0.f};
    const float sg = head_scale(prm.amax_g), sx = head_scale(prm.amax_x);
    const float unscale = 1.f / (sg * sx);
    const int Bimg = (prm.M + prm.HW - 1) / prm.HW;
    // column validity as 0 / 1 factors, clamped column numbers for the addresses (every load is unconditional)
    f32x4 nkeep, ckeep;
    int ngc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        nkeep[e] = ng + e < prm.N ? 1.f : 0.f;
        ckeep[e] = cx + e < prm.Kc ? 1.f : 0.f;
        ngc[e] = min(ng + e, prm.N - 1);
    }
    const int cxc = min(cx, prm.Kc - 4);             // (Kc is a multiple of 32: a run of four is inside or outside)
    const int ngv = min(ng, prm.N - 4 < 0 ? 0 : prm.N - 4);

    // load_stage: loads only, into raw registers; commit_stage scales, masks, sums the bias columns, splits and stores row-major
    f32x4 rg[4], rx[4], rk0, rk1;
    auto load_stage = [&](int st) {
        const int t0 = st * HKS + trow;
        // my rows t0 + 8 q span 24 tokens: at most two images when HW >= 32 (the host takes version 1 below that)
        const int b0 = min(t0 / prm.HW, Bimg - 1), b1 = min(b0 + 1, Bimg - 1);
        const int first1 = (b0 + 1) * prm.HW;
        rk0 = *reinterpret_cast<const f32x4*>(prm.maskX + (size_t)b0 * prm.mask_ld + cxc);
        rk1 = *reinterpret_cast<const f32x4*>(prm.maskX + (size_t)b1 * prm.mask_ld + cxc);
        const float* x0 = prm.X + (long long)b0 * prm.x_img + cxc - (long long)b0 * prm.HW * prm.x_tok;
        const float* x1 = prm.X + (long long)b1 * prm.x_img + cxc - (long long)b1 * prm.HW * prm.x_tok;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = t0 + 8 * q;
            const int mcl = m < prm.M ? m : prm.M - 1;
            rx[q] = *reinterpret_cast<const f32x4*>((mcl >= first1 ? x1 : x0) + (long long)mcl * prm.x_tok);
            const float* gr = prm.G + (size_t)mcl * prm.ldg;
            if (gvec) rg[q] = *reinterpret_cast<const f32x4*>(gr + ngv);
            else rg[q] = f32x4{gr[ngc[0]], gr[ngc[1]], gr[ngc[2]], gr[ngc[3]]};
        }
    };
    auto put = [&](unsigned char* plane_hi, int row, const float (&v)[4]) {
        unsigned h0, l0, h1, l1;
        split_f16_pair(v[0], v[1], h0, l0);
        split_f16_pair(v[2], v[3], h1, l1);
        *reinterpret_cast<u32x2*>(plane_hi + row * WG_SR + 8 * c4) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(plane_hi + WG_PLANE + row * WG_SR + 8 * c4) = u32x2{l0, l1};
    };
    auto commit_stage = [&](unsigned char* stg, int st) {
        const int t0 = st * HKS + trow;
        const int first1 = (min(t0 / prm.HW, Bimg - 1) + 1) * prm.HW;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = t0 + 8 * q;
            const int mcl = m < prm.M ? m : prm.M - 1;
            // (zeroing by 0 / 1 factors, not by selects: a value that is only used conditionally has its load sunk into a branch)
            const float okf = m < prm.M ? 1.f : 0.f;
            const f32x4 mk = mcl >= first1 ? rk1 : rk0;
            float vg[4], vx[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gq = rg[q][e] * (okf * nkeep[e]);
                bsum[e] += gq;
                vg[e] = gq * sg;
                vx[e] = rx[q][e] * (mk[e] * (sx * okf * ckeep[e]));
            }
            put(stg, trow + 8 * q, vg);
            put(stg + WG_SIDE, trow + 8 * q, vx);
        }
    };
    auto mma_stage = [&](const unsigned char* stg) {
        const unsigned char* Gh = stg;
        const unsigned char* Xh = stg + WG_SIDE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const f16x8 ah0 = wg_frag(Gh, ks, 64 * wr, lane), al0 = wg_frag(Gh + WG_PLANE, ks, 64 * wr, lane);
            const f16x8 ah1 = wg_frag(Gh, ks, 64 * wr + 32, lane), al1 = wg_frag(Gh + WG_PLANE, ks, 64 * wr + 32, lane);
            const f16x8 bh0 = wg_frag(Xh, ks, 64 * wc, lane), bl0 = wg_frag(Xh + WG_PLANE, ks, 64 * wc, lane);
            const f16x8 bh1 = wg_frag(Xh, ks, 64 * wc + 32, lane), bl1 = wg_frag(Xh + WG_PLANE, ks, 64 * wc + 32, lane);
            // (operands swapped: transposed blocks, see head_mma_stage)
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh0, al0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh1, al0, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh0, al1, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh1, al1, acc[1][1], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl0, ah0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl1, ah0, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl0, ah1, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl1, ah1, acc[1][1], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh0, ah0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh1, ah0, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh0, ah1, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh1, ah1, acc[1][1], 0, 0, 0);
        }
    };
    if (s_beg < s_end) {
        load_stage(s_beg);
        commit_stage(lds, s_beg);
    }
    __syncthreads();
    int cur = 0;
    for (int st = s_beg; st < s_end; ++st) {
        const bool more = st + 1 < s_end;
        load_stage(min(st + 1, s_end - 1));
        __builtin_amdgcn_sched_barrier(0);       // the loads go out BEFORE the MFMAs (left alone, the scheduler puts them behind)
        mma_stage(lds + cur * WG_STAGE);
        __builtin_amdgcn_sched_barrier(0);
        if (more) commit_stage(lds + (cur ^ 1) * WG_STAGE, st + 1);
        __syncthreads();
        cur ^= 1;
    }
    // (transposed blocks, see head_mma_stage: lane & 31 is the row n, registers hold runs of four consecutive channels)
    float* out = prm.part + (size_t)split * prm.N * prm.Kc;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int nn = n0 + 64 * wr + 32 * mi + (lane & 31);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int cc = c0 + 64 * wc + 32 * ni + 8 * jj + 4 * (lane >> 5);
                if (nn < prm.N && cc < prm.Kc) {                   // (Kc is a multiple of 32: a run of four is inside or outside)
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[mi][ni][4 * jj + e] * unscale;
                    *reinterpret_cast<f32x4*>(out + (size_t)nn * prm.Kc + cc) = v;
                }
            }
    }
    if (blockIdx.y == 0 && prm.part_bias) {       // column sums of G': 8 partial rows (one per token-row group) through the dead stages
        f32x4* cs = reinterpret_cast<f32x4*>(lds);
        cs[trow * 32 + c4] = bsum;
        __syncthreads();
        if (tid < 32) {
            f32x4 t = cs[tid];
#pragma unroll
            for (int r = 1; r < 8; ++r) t += cs[r * 32 + tid];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n0 + 4 * tid + e < prm.N) prm.part_bias[(size_t)split * prm.N + n0 + 4 * tid + e] = t[e];
        }
    }
}

// (A 128 x 384 variant of this kernel - the G columns of a tile staged once for all channels, 12 waves, 462 -> 308 MB of L2-level
// traffic for the C x C gradient - was built and measured: 155 us against 142 us for the 3 x 3 tiles of 128 x 128 at two workgroups
// per CU.  Like the GEMMs, this kernel runs at the ~11 bytes per cycle a CU is delivered under full-chip load, not at its MFMA rate.)
__global__ void __launch_bounds__(256) head_wgrad_kernel(const HeadWgradParams prm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];      // two stages of [G 16 KB][X 16 KB] (double buffer)
    __shared__ float colsum[2][HT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int n0 = blockIdx.x * HT, c0 = blockIdx.y * HT, split = blockIdx.z;
    // token range of this split: whole stages of 32 tokens
    const int n_stage = (prm.M + HKS - 1) / HKS;
    const int s_beg = (int)((long long)n_stage * split / prm.splits), s_end = (int)((long long)n_stage * (split + 1) / prm.splits);
    const int col = tid & 127, g = tid >> 7;     // my column (of G: n, of X: channel), my 16 tokens of the stage
    const int n = n0 + col, c = c0 + col;
    const bool n_ok = n < prm.N, c_ok = c < prm.Kc;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float bsum = 0.f;
    const float sg = head_scale(prm.amax_g), sx = head_scale(prm.amax_x);
    const float unscale = 1.f / (sg * sx);

    const int ncl = n_ok ? n : prm.N - 1, ccl = c_ok ? c : prm.Kc - 1;     // (clamped: every load below is unconditional, see the GEMM)
    const float n_keep = n_ok ? 1.f : 0.f, c_keep = c_ok ? 1.f : 0.f;
    const int Bimg = (prm.M + prm.HW - 1) / prm.HW;
    // load_stage: loads only, into raw registers (anything computed from a loaded value here would make the compiler wait for the load
    // before the MFMAs that follow); commit_stage scales, masks, sums the bias column and splits
    float rg[16], rx[16], rk0, rk1;
    auto load_stage = [&](int st) {
        const int t0 = st * HKS + 16 * g;
        // my 16 tokens lie in at most two images (HW >= 16 is checked by the host): their bases and mask values once per stage
        const int b0 = min(t0 / prm.HW, Bimg - 1), b1 = min(b0 + 1, Bimg - 1);
        const int first1 = (b0 + 1) * prm.HW;                 // first token of the next image
        const float* x0 = prm.X + (long long)b0 * prm.x_img + ccl - (long long)b0 * prm.HW * prm.x_tok;
        const float* x1 = prm.X + (long long)b1 * prm.x_img + ccl - (long long)b1 * prm.HW * prm.x_tok;
        rk0 = prm.maskX[(size_t)b0 * prm.mask_ld + ccl];      // (maskX is never null: a row of ones with mask_ld = 0 when there is no mask)
        rk1 = prm.maskX[(size_t)b1 * prm.mask_ld + ccl];
        const float* gp = prm.G + ncl;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = t0 + e;
            const int mcl = m < prm.M ? m : prm.M - 1;
            rg[e] = gp[(size_t)mcl * prm.ldg];
            rx[e] = (mcl >= first1 ? x1 : x0)[(long long)mcl * prm.x_tok];
        }
    };
    auto commit_stage = [&](unsigned char* stg, int st) {
        const int t0 = st * HKS + 16 * g;
        const int first1 = (min(t0 / prm.HW, Bimg - 1) + 1) * prm.HW;
        float vg[16], vx[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = t0 + e;
            const int mcl = m < prm.M ? m : prm.M - 1;
            // (zeroing by a 0 / 1 factor, not by a select: a value that is only used conditionally has its load sunk into a branch)
            const float okf = m < prm.M ? 1.f : 0.f;
            const float gq = rg[e] * (okf * n_keep);
            if (blockIdx.y == 0) bsum += gq;
            vg[e] = gq * sg;
            vx[e] = rx[e] * ((mcl >= first1 ? rk1 : rk0) * (sx * okf * c_keep));
        }
        head_commit16(stg, col, g, vg);           // row = my column, the 32 tokens of the stage are its "channels"
        head_commit16(stg + HSIDE, col, g, vx);
    };
    // double-buffered like the GEMM: one barrier per stage; loads unconditional from a clamped stage number
    if (s_beg < s_end) {
        load_stage(s_beg);
        commit_stage(lds, s_beg);
    }
    __syncthreads();
    int cur = 0;
    for (int st = s_beg; st < s_end; ++st) {
        const bool more = st + 1 < s_end;
        load_stage(min(st + 1, s_end - 1));
        __builtin_amdgcn_sched_barrier(0);       // the loads go out BEFORE the MFMAs (left alone, the scheduler puts them behind)
        head_mma_stage<8192, 8192>(lds + cur * 2 * HSIDE, lds + cur * 2 * HSIDE + HSIDE, acc, lane, wr, wc);
        __builtin_amdgcn_sched_barrier(0);
        if (more) commit_stage(lds + (cur ^ 1) * 2 * HSIDE, st + 1);
        __syncthreads();
        cur ^= 1;
    }
    // (transposed blocks, see head_mma_stage: lane & 31 is the row n, registers hold runs of four consecutive channels)
    float* out = prm.part + (size_t)split * prm.N * prm.Kc;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int nn = n0 + 64 * wr + 32 * mi + (lane & 31);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int cc = c0 + 64 * wc + 32 * ni + 8 * jj + 4 * (lane >> 5);
                if (nn < prm.N && cc < prm.Kc) {                   // (Kc is a multiple of 32: a run of four is inside or outside)
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[mi][ni][4 * jj + e] * unscale;
                    *reinterpret_cast<f32x4*>(out + (size_t)nn * prm.Kc + cc) = v;
                }
            }
    }
    if (blockIdx.y == 0 && prm.part_bias) {
        colsum[g][col] = bsum;
        __syncthreads();
        if (g == 0 && n_ok) prm.part_bias[(size_t)split * prm.N + n] = colsum[0][col] + colsum[1][col];
    }
}

// dW[i] = sum_s part[s][i]  (fixed order);  db[n] = sum_s part_bias[s][n] (optionally into two outputs: b1 and b22 share their gradient).
// numel is a multiple of 4 (Kc is a multiple of 32).  A workgroup owns 64 runs of four consecutive outputs; its four waves sum every
// fourth partial tile each (loads of 16 tiles in flight per lane), then wave 0 adds the four wave sums in wave order: the order of the
// additions is fixed, the result bitwise repeatable.  (One thread per run summing all tiles alone: 144 workgroups, 50 MB in 52 us.)
__global__ void __launch_bounds__(256) head_reduce_kernel(const float* part, int splits, long long numel, float* dW,
                                                          const float* part_bias, int N, float* db, float* db_b)
{
    __shared__ f32x4 wsum[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long i = ((long long)blockIdx.x * 64 + lane) * 4;
    const bool ok = i < numel;
    const long long ic = ok ? i : 0;
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
    int k = wave;
    for (; k + 4 * 15 < splits; k += 4 * 16) {
        f32x4 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = *reinterpret_cast<const f32x4*>(part + (size_t)(k + 4 * j) * numel + ic);
#pragma unroll
        for (int j = 0; j < 16; ++j) s += v[j];
    }
    for (; k < splits; k += 4) s += *reinterpret_cast<const f32x4*>(part + (size_t)k * numel + ic);
    wsum[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && ok) *reinterpret_cast<f32x4*>(dW + i) = ((wsum[0][lane] + wsum[1][lane]) + wsum[2][lane]) + wsum[3][lane];
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (part_bias && t < N) {
        float sb = 0.f;
        for (int kk = 0; kk < splits; ++kk) sb += part_bias[(size_t)kk * N + t];
        if (db) db[t] = sb;
        if (db_b) db_b[t] = sb;
    }
}

// ------------------------------------------------------------------------------------------------ cached tokens
// out[i][t][:] = float(table[index[i]][t][:]); a thread converts 8 halves (16 bytes in, 32 out); the largest magnitude of the rows
// t >= skip_rows goes to *amax (one conditional atomic per workgroup)
__global__ void __launch_bounds__(256) head_tokens_from_cache_kernel(const half_t* table, const long long* index, int n, int ntok, int D,
                                                                     int skip_rows, float* out, unsigned* amax)
{
    __shared__ float red[4];
    const int d8 = D >> 3;
    const long long per_item = (long long)ntok * d8, total = (long long)n * per_item;
    const long long step = (long long)gridDim.x * 256;
    float mx = 0.f;
    for (long long u0 = (long long)blockIdx.x * 256 + threadIdx.x; u0 < total; u0 += 4 * step) {
        f16x8 h[4];
        long long u[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {                 // four 16-byte loads in flight per thread (clamped: the tail re-reads the last unit)
            u[q] = min(u0 + q * step, total - 1);
            const int i = (int)(u[q] / per_item);
            h[q] = *reinterpret_cast<const f16x8*>(table + ((size_t)index[i] * per_item + (u[q] - (long long)i * per_item)) * 8);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (u0 + q * step >= total) break;
            f32x4 a, b;
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[e] = (float)h[q][e]; b[e] = (float)h[q][4 + e]; }
            float* o = out + (size_t)u[q] * 8;
            *reinterpret_cast<f32x4*>(o) = a;
            *reinterpret_cast<f32x4*>(o + 4) = b;
            const long long r = u[q] % per_item;                     // unit inside the item: row t = r / d8
            if (r >= (long long)skip_rows * d8) {
#pragma unroll
                for (int e = 0; e < 4; ++e) mx = fmaxf(mx, fmaxf(fabsf(a[e]), fabsf(b[e])));
            }
        }
    }
    if (amax) head_publish_max(mx, amax, red);
}

// ------------------------------------------------------------------------------------------------ host side
static int head_check(const StegoHeadDesc* d)
{
    if (!d) return STEGO_ERR_NULL;
    if (d->B <= 0 || d->HW <= 0 || d->C <= 0 || d->K <= 0) return STEGO_ERR_SHAPE;
    if (d->C % 32 != 0 || d->K > 128 || d->tok_stride < d->C || d->img_stride < (int64_t)d->HW * d->tok_stride) return STEGO_ERR_UNSUPPORTED;
    if (d->HW < 16) return STEGO_ERR_UNSUPPORTED;                                                   // (a stage of the weight gradient spans <= 2 images)
    if (d->tok_stride % 4 != 0 || d->img_stride % 4 != 0) return STEGO_ERR_UNSUPPORTED;            // 16-byte token rows
    if ((int64_t)d->B * d->HW >= (1ll << 31) / 2 || (int64_t)d->B * d->HW * d->C >= (1ll << 30)) return STEGO_ERR_UNSUPPORTED;   // 32-bit byte offsets
    return STEGO_OK;
}

static size_t round256(size_t v) { return (v + 255) / 256 * 256; }

// channel blocks per workgroup of the weight-gradient GEMM of an N-column upstream: the wide tile for the C x C gradient
static int wgrad_splits(const StegoHeadDesc* d, int N)
{
    const int tiles = ((N + HT - 1) / HT) * ((d->C + HT - 1) / HT);
    const long long stages = ((long long)d->B * d->HW + HKS - 1) / HKS;
    // ~two workgroups per compute unit (one per CU: partial tiles 50 -> 25 MB, reduction 28 -> 16 us, but the GEMMs 81 -> 100 us:
    // measured, not kept)
    long long s = (2 * 256 + tiles - 1) / tiles;
    if (s > stages) s = stages;
    return s < 1 ? 1 : (int)s;
}

// columns of a C-wide output handled per workgroup: 384 (one staging of the token operand) when C is a multiple of it
static int wide_blocks(int N) { return N % (3 * HT) == 0 ? 3 : 1; }
static size_t plane_halves(int N, int K, int nw) { return (size_t)((N + HT * nw - 1) / (HT * nw) * (HT * nw)) * (size_t)((K + HKS - 1) / HKS * HKS); }

template <bool KVEC, int EPI, int NW>
static hipError_t launch_gemm_t(const HeadGemmParams& p, hipStream_t s)
{
    const int lds_bytes = 2 * (HSIDE + 2 * NW * 8192) + (KVEC ? (EPI == EPI_BIAS ? 2 : 1) * p.n_slot * p.K * (int)sizeof(float) : 0);
    if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&head_gemm_kernel<KVEC, EPI, NW>), lds_bytes);
    if (e != hipSuccess) return e;
    const dim3 grid((p.M + HT - 1) / HT, (p.N + HT * NW - 1) / (HT * NW)), block(256 * NW);
    hipLaunchKernelGGL((head_gemm_kernel<KVEC, EPI, NW>), grid, block, lds_bytes, s, p);
    return hipGetLastError();
}

static hipError_t launch_gemm(const HeadGemmParams& p, bool kvec, int nw, hipStream_t s)
{
    if (kvec && p.epi == EPI_BIAS && nw == 1) return launch_gemm_t<true, EPI_BIAS, 1>(p, s);
    if (kvec && p.epi == EPI_BIAS_RELU && nw == 1) return launch_gemm_t<true, EPI_BIAS_RELU, 1>(p, s);
    if (kvec && p.epi == EPI_BIAS_RELU && nw == 3) return launch_gemm_t<true, EPI_BIAS_RELU, 3>(p, s);
    if (kvec && p.epi == EPI_ACCUM && nw == 1) return launch_gemm_t<true, EPI_ACCUM, 1>(p, s);
    if (!kvec && p.epi == EPI_MASK_POS && nw == 1) return launch_gemm_t<false, EPI_MASK_POS, 1>(p, s);
    if (!kvec && p.epi == EPI_MASK_POS && nw == 3) return launch_gemm_t<false, EPI_MASK_POS, 3>(p, s);
    return hipErrorInvalidValue;
}

// one operand of a GEMM as the planes head_prep_weights_kernel writes: carves them from `ws`, fills the job
struct HeadPlanes { half_t* hi; half_t* lo; int ld; };
static HeadPlanes carve_planes(unsigned char*& ws, HeadPrepJob& jb, const float* src, int N, int K, int ld, int transposed, int nw,
                               unsigned* amax)
{
    const size_t n = plane_halves(N, K, nw);
    HeadPlanes pl;
    pl.hi = reinterpret_cast<half_t*>(ws); ws += (n * 2 + 255) / 256 * 256;
    pl.lo = reinterpret_cast<half_t*>(ws); ws += (n * 2 + 255) / 256 * 256;
    pl.ld = (K + HKS - 1) / HKS * HKS;
    jb.src = src; jb.N = N; jb.K = K; jb.ld = ld; jb.transposed = transposed; jb.hi = pl.hi; jb.lo = pl.lo;
    jb.Npad = (N + HT * nw - 1) / (HT * nw) * (HT * nw); jb.Kpad = pl.ld; jb.amax = amax;
    return pl;
}
static size_t planes_bytes(int N, int K, int nw) { return 2 * ((plane_halves(N, K, nw) * 2 + 255) / 256 * 256); }

}  // namespace stego

using namespace stego;

extern "C" {

// operand-scale words (bits of the largest magnitude, see head_scale): the forward fills 0-4 and hands them to the backward
enum { HS_X = 0, HS_W1, HS_W21, HS_W22, HS_H, HS_G, HS_DH, HS_COUNT = HS_WORDS };

int stego_tokens_from_cache(const void* table_f16, const int64_t* index, int32_t n, int32_t ntok, int32_t D, int32_t skip_rows,
                            float* out, uint32_t* amax_bits, stego_stream_t stream)
{
    (void)hipGetLastError();
    if (n < 0 || ntok <= 0 || D <= 0 || (D & 7) || skip_rows < 0) return STEGO_ERR_SHAPE;
    if (n == 0) return STEGO_OK;
    if (!table_f16 || !index || !out) return STEGO_ERR_NULL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e;
    if (amax_bits && (e = hipMemsetAsync(amax_bits, 0, sizeof(uint32_t), s)) != hipSuccess) return STEGO_ERR_HIP + (int)e;
    const long long units = (long long)n * ntok * (D >> 3);
    long long blocks = (units + 256 * 4 - 1) / (256 * 4);
    blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
    hipLaunchKernelGGL(head_tokens_from_cache_kernel, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const half_t*>(table_f16),
                       reinterpret_cast<const long long*>(index), n, ntok, D, skip_rows, out, reinterpret_cast<unsigned*>(amax_bits));
    if ((e = hipGetLastError()) != hipSuccess) return STEGO_ERR_HIP + (int)e;
    return STEGO_OK;
}

size_t stego_head_fwd_workspace_bytes(const StegoHeadDesc* d)
{
    if (head_check(d) != STEGO_OK) return 0;
    size_t total = 256 + 2 * round256((size_t)d->C * sizeof(float)) + planes_bytes(d->K, d->C, 1);     // scale words, a row of ones, a row of zeros, W1 planes
    if (d->nonlinear) total += planes_bytes(d->C, d->C, wide_blocks(d->C)) + planes_bytes(d->K, d->C, 1)     // W21, W22 planes
                               + round256((size_t)d->B * d->HW * d->C * sizeof(float));                      // H (when saved_h is NULL)
    return total;
}

size_t stego_head_bwd_workspace_bytes(const StegoHeadDesc* d)
{
    if (head_check(d) != STEGO_OK) return 0;
    const size_t M = (size_t)d->B * d->HW;
    size_t part = (size_t)wgrad_splits(d, d->K) * d->K * d->C, pbias = (size_t)wgrad_splits(d, d->K) * d->K;
    size_t total = 256 + round256((size_t)d->C * 4) + round256(part * 4) + round256(pbias * 4);
    if (d->nonlinear) {
        total += planes_bytes(d->C, d->K, wide_blocks(d->C));                              // W22^T planes
        total += round256(M * d->C * 4);                                                   // dHpre
        total += round256((size_t)wgrad_splits(d, d->C) * d->C * d->C * 4) + round256((size_t)wgrad_splits(d, d->C) * d->C * 4);
    }
    return total + 256;
}

int stego_head_fwd(const StegoHeadDesc* d, const float* tokens, const float* mask1, const float* mask2, const float* mask3,
                   const float* w1, const float* b1, const float* w21, const float* b21, const float* w22, const float* b22,
                   float* code, float* feats_out, float* saved_h, void* workspace, size_t workspace_bytes, stego_stream_t stream)
{
    (void)hipGetLastError();
    int rc = head_check(d);
    if (rc) return rc;
    if (!tokens || !w1 || !b1 || !code) return STEGO_ERR_NULL;
    if (d->nonlinear && (!w21 || !b21 || !w22 || !b22)) return STEGO_ERR_NULL;
    if (!workspace) return STEGO_ERR_NULL;
    if (workspace_bytes < stego_head_fwd_workspace_bytes(d) - (d->nonlinear && saved_h ? round256((size_t)d->B * d->HW * d->C * sizeof(float)) : 0))
        return STEGO_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int M = d->B * d->HW;
    hipError_t e;
    // operand scales: the first 32 bytes of the workspace (or of saved_scales' home: the tail of saved_h, see the header)
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    unsigned* sc = saved_h ? reinterpret_cast<unsigned*>(saved_h + (size_t)M * d->C) : reinterpret_cast<unsigned*>(ws);
    // [scale words 256 B][a row of zeros: the absent second bias; C >= K][a row of ones: the mask operand of whatever has no dropout
    // mask (kernels never test a mask pointer)], all written by one small launch
    float* zeros = reinterpret_cast<float*>(ws + 256);
    const size_t zbytes = round256((size_t)d->C * sizeof(float));
    ws += 256 + zbytes;
    float* ones = reinterpret_cast<float*>(ws); ws += zbytes;
    static_assert(HS_X == 0, "head_consts_kernel copies word 0");
    hipLaunchKernelGGL(head_consts_kernel, dim3((d->C + 255) / 256), dim3(256), 0, s, sc, reinterpret_cast<const unsigned*>(d->tokens_amax),
                       -1, zeros, ones, d->C);
    if ((e = hipGetLastError()) != hipSuccess) return STEGO_ERR_HIP + (int)e;
    if (!d->tokens_amax) {
        int blocks = (int)(((long long)M * d->C / 4 + 256 * 4 - 1) / (256 * 4));
        blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
        hipLaunchKernelGGL(head_absmax_kernel, dim3(blocks), dim3(256), 0, s, tokens, (long long)d->img_stride, (long long)d->tok_stride,
                           d->HW, (long long)M, d->C, sc + HS_X);
        if ((e = hipGetLastError()) != hipSuccess) return STEGO_ERR_HIP + (int)e;
    }
    // the weights as fp16 hi / lo planes (+ their scale words), two small launches
    const int nw21 = wide_blocks(d->C);
    HeadPrepParams prep{};
    const HeadPlanes pw1 = carve_planes(ws, prep.job[0], w1, d->K, d->C, d->C, 0, 1, sc + HS_W1);
    HeadPlanes pw21{}, pw22{};
    if (d->nonlinear) {
        pw21 = carve_planes(ws, prep.job[1], w21, d->C, d->C, d->C, 0, nw21, sc + HS_W21);
        pw22 = carve_planes(ws, prep.job[2], w22, d->K, d->C, d->C, 0, 1, sc + HS_W22);
    }
    hipLaunchKernelGGL(head_prep_absmax_kernel, dim3(d->nonlinear ? 3 : 1, PREP_WG), dim3(256), 0, s, prep);
    hipLaunchKernelGGL(head_prep_split_kernel, dim3(d->nonlinear ? 3 : 1, PREP_WG), dim3(256), 0, s, prep);
    if ((e = hipGetLastError()) != hipSuccess) return STEGO_ERR_HIP + (int)e;
    HeadGemmParams p{};
    p.n_img = d->B; p.n_slot = std::min(d->B, (HT - 1) / d->HW + 2);
    p.amax_a = sc + HS_X; p.amax_b = sc + HS_W1;
    p.A = tokens; p.a_img = d->img_stride; p.a_tok = d->tok_stride; p.HW = d->HW; p.M = M; p.K = d->C;
    // cluster1 (+ both output biases: cluster2[2]'s is added here so that the third GEMM only accumulates) (+ feats_out)
    p.maskA = mask1 ? mask1 : ones; p.mask_ld = mask1 ? d->C : 0; p.Bh = pw1.hi; p.Bl = pw1.lo; p.ldb = pw1.ld; p.C = code; p.ldc = d->K; p.N = d->K;
    p.bias = b1; p.bias2 = d->nonlinear ? b22 : zeros; p.epi = EPI_BIAS;
    p.feats_out = feats_out; p.mask3 = mask3 ? mask3 : ones; p.mask3_ld = mask3 ? d->C : 0;
    if ((e = launch_gemm(p, true, 1, s)) != hipSuccess) return STEGO_ERR_HIP + (int)e;
    if (!d->nonlinear) return STEGO_OK;
    float* H = saved_h ? saved_h : reinterpret_cast<float*>(ws);
    // H = relu(cluster2[0](x * m2))
    p.maskA = mask2 ? mask2 : ones; p.mask_ld = mask2 ? d->C : 0; p.Bh = pw21.hi; p.Bl = pw21.lo; p.ldb = pw21.ld; p.C = H; p.ldc = d->C; p.N = d->C;
    p.bias = b21; p.bias2 = zeros; p.epi = EPI_BIAS_RELU; p.feats_out = nullptr; p.mask3 = ones; p.mask3_ld = 0;
    p.amax_b = sc + HS_W21; p.amax_out = sc + HS_H;
    if ((e = launch_gemm(p, true, nw21, s)) != hipSuccess) return STEGO_ERR_HIP + (int)e;
    // code += cluster2[2](H)
    HeadGemmParams q{};
    q.n_img = p.n_img; q.n_slot = p.n_slot;
    q.A = H; q.a_img = (long long)d->HW * d->C; q.a_tok = d->C; q.HW = d->HW; q.M = M; q.K = d->C;
    q.maskA = ones; q.mask_ld = 0; q.mask3 = ones; q.mask3_ld = 0;
    q.Bh = pw22.hi; q.Bl = pw22.lo; q.ldb = pw22.ld; q.C = code; q.ldc = d->K; q.N = d->K; q.epi = EPI_ACCUM;
    q.amax_a = sc + HS_H; q.amax_b = sc + HS_W22;
    if ((e = launch_gemm(q, true, 1, s)) != hipSuccess) return STEGO_ERR_HIP + (int)e;
    return STEGO_OK;
}

int stego_head_bwd(const StegoHeadDesc* d, const float* tokens, const float* mask1, const float* mask2, const float* saved_h,
                   const float* w22, const float* d_code, float* dw1, float* db1, float* dw21, float* db21, float* dw22,
                   float* db22, void* workspace, size_t workspace_bytes, stego_stream_t stream)
{
    (void)hipGetLastError();
    int rc = head_check(d);
    if (rc) return rc;
    if (!tokens || !d_code || !dw1 || !db1 || !workspace) return STEGO_ERR_NULL;
    if (d->nonlinear && (!saved_h || !w22 || !dw21 || !db21 || !dw22 || !db22)) return STEGO_ERR_NULL;
    if (workspace_bytes < stego_head_bwd_workspace_bytes(d)) return STEGO_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int M = d->B * d->HW, C = d->C, K = d->K;
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    const int sk = wgrad_splits(d, K), sc = wgrad_splits(d, C);
    // operand scales: the forward's (behind saved_h, or - linear head - recomputed here for X) + G's
    unsigned* scl = reinterpret_cast<unsigned*>(ws); ws += 256;
    float* ones = reinterpret_cast<float*>(ws); ws += round256((size_t)C * 4);       // the mask operand of whatever has no dropout mask
    {
        hipError_t e0;
        // the forward's words X .. H (HS_G onwards are this call's: zeroed; W22's word holds the maximum of the same matrix the prep
        // kernel raises it to again), or - linear head - everything zeroed and X recomputed; + the row of ones: one launch
        hipLaunchKernelGGL(head_consts_kernel, dim3((C + 255) / 256), dim3(256), 0, s, scl,
                           d->nonlinear ? reinterpret_cast<const unsigned*>(saved_h + (size_t)M * C) : (const unsigned*)nullptr, (int)HS_G,
                           (float*)nullptr, ones, C);
        if ((e0 = hipGetLastError()) != hipSuccess) return STEGO_ERR_HIP + (int)e0;
        auto absmax = [&](const float* x, long long img, long long rowst, int rpi, long long rows, int cols, int which) -> hipError_t {
            int blocks = (int)((rows * cols / 4 + 256 * 4 - 1) / (256 * 4));
            blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
            if (rowst == cols && img == (long long)rpi * cols && ((rows * cols) & 3) == 0) {
                // a dense matrix is one flat vector: 16-byte units whatever the row length (the K = 70 upstream took the scalar path)
                const long long units = rows * cols / 4;
                rows = units; rpi = (int)std::min<long long>(units, 1ll << 30); cols = 4; rowst = 4; img = (long long)rpi * 4;
            }
            hipLaunchKernelGGL(head_absmax_kernel, dim3(blocks), dim3(256), 0, s, x, img, rowst, rpi, rows, cols, scl + which);
            return hipGetLastError();
        };
        if (!d->nonlinear && (e0 = absmax(tokens, d->img_stride, d->tok_stride, d->HW, M, C, HS_X)) != hipSuccess) return STEGO_ERR_HIP + (int)e0;
        if ((e0 = absmax(d_code, (long long)d->HW * K, K, d->HW, M, K, HS_G)) != hipSuccess) return STEGO_ERR_HIP + (int)e0;
    }
    float* part_k = reinterpret_cast<float*>(ws); ws += round256((size_t)sk * K * C * 4);
    float* pbias_k = reinterpret_cast<float*>(ws); ws += round256((size_t)sk * K * 4);
    hipError_t e;
    auto wgrad = [&](const float* G, int ldg, int N, const float* X, long long x_img, long long x_tok, const float* maskX,
                     float* part, float* pbias, int splits, float* dW, float* db, float* db_b) -> hipError_t {
        HeadWgradParams w{};
        w.G = G; w.ldg = ldg; w.X = X; w.x_img = x_img; w.x_tok = x_tok; w.HW = d->HW;
        w.maskX = maskX ? maskX : ones; w.mask_ld = maskX ? C : 0;
        w.part = part; w.part_bias = pbias; w.M = M; w.N = N; w.Kc = C; w.splits = splits;
        w.amax_g = G == d_code ? scl + HS_G : scl + HS_DH;
        w.amax_x = X == tokens ? scl + HS_X : scl + HS_H;
        hipError_t er;
        const dim3 grid((N + HT - 1) / HT, (C + HT - 1) / HT, splits);
        if (d->HW >= HKS && (x_tok & 3) == 0 && (x_img & 3) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0 &&
            (reinterpret_cast<uintptr_t>(w.maskX) & 15) == 0) {
            // (HW >= 32: a stage of 32 tokens spans at most two images, which is what the mask selection assumes)
            if ((er = ensure_dynamic_lds(reinterpret_cast<const void*>(&head_wgrad_tr_kernel), 2 * WG_STAGE)) != hipSuccess) return er;
            hipLaunchKernelGGL(head_wgrad_tr_kernel, grid, dim3(256), 2 * WG_STAGE, s, w);
        } else {
            if ((er = ensure_dynamic_lds(reinterpret_cast<const void*>(&head_wgrad_kernel), 4 * HSIDE)) != hipSuccess) return er;
            hipLaunchKernelGGL(head_wgrad_kernel, grid, dim3(256), 4 * HSIDE, s, w);
        }
        er = hipGetLastError();
        if (er != hipSuccess) return er;
        const long long numel = (long long)N * C;
        hipLaunchKernelGGL(head_reduce_kernel, dim3((unsigned)std::max<long long>((numel / 4 + 63) / 64, (N + 255) / 256)), dim3(256), 0, s, part, splits, numel, dW,
                           pbias, N, db, db_b);
        return hipGetLastError();
    };
    // dW1 = G^T (x * m1), db1 = colsum(G) (= db22)
    if ((e = wgrad(d_code, K, K, tokens, d->img_stride, d->tok_stride, mask1, part_k, pbias_k, sk, dw1, db1,
                   d->nonlinear ? db22 : nullptr)) != hipSuccess) return STEGO_ERR_HIP + (int)e;
    if (!d->nonlinear) return STEGO_OK;
    // dW22 = G^T H
    if ((e = wgrad(d_code, K, K, saved_h, (long long)d->HW * C, C, nullptr, part_k, nullptr, sk, dw22, nullptr, nullptr)) != hipSuccess)
        return STEGO_ERR_HIP + (int)e;
    // dHpre = (G W22) * 1[H > 0]
    float* dH = reinterpret_cast<float*>(ws); ws += round256((size_t)M * C * 4);
    float* part_c = reinterpret_cast<float*>(ws); ws += round256((size_t)sc * C * C * 4);
    float* pbias_c = reinterpret_cast<float*>(ws);
    const int nwb = wide_blocks(C);
    HeadPrepParams prep{};
    unsigned char* wsp = reinterpret_cast<unsigned char*>(pbias_c) + round256((size_t)sc * C * 4);
    const HeadPlanes pwt = carve_planes(wsp, prep.job[0], w22, C, K, C, 1, nwb, scl + HS_W22);      // B[n = channel j][k] = w22[k][j]
    // (W22's scale word holds the forward's maximum of the same matrix: raising it to the same value again is a no-op, no zeroing)
    hipLaunchKernelGGL(head_prep_absmax_kernel, dim3(1, PREP_WG), dim3(256), 0, s, prep);
    hipLaunchKernelGGL(head_prep_split_kernel, dim3(1, PREP_WG), dim3(256), 0, s, prep);
    if ((e = hipGetLastError()) != hipSuccess) return STEGO_ERR_HIP + (int)e;
    HeadGemmParams p{};
    p.n_img = d->B; p.n_slot = 1;
    p.A = d_code; p.a_img = (long long)d->HW * K; p.a_tok = K; p.HW = d->HW; p.M = M; p.K = K;
    p.Bh = pwt.hi; p.Bl = pwt.lo; p.ldb = pwt.ld;
    p.C = dH; p.ldc = C; p.N = C; p.epi = EPI_MASK_POS; p.aux = saved_h; p.ldaux = C;
    p.amax_a = scl + HS_G; p.amax_b = scl + HS_W22; p.amax_out = scl + HS_DH;
    if ((e = launch_gemm(p, false, nwb, s)) != hipSuccess) return STEGO_ERR_HIP + (int)e;
    // dW21 = dHpre^T (x * m2), db21 = colsum(dHpre)
    if ((e = wgrad(dH, C, C, tokens, d->img_stride, d->tok_stride, mask2, part_c, pbias_c, sc, dw21, db21, nullptr)) != hipSuccess)
        return STEGO_ERR_HIP + (int)e;
    return STEGO_OK;
}

}  // extern "C"
