// Segmentation head of DinoFeaturizer on gfx950 (include/stego_head.h): the producer of `code`, forward and backward.
//
// Reference: src/modules.py:108-116 (Dropout2d x 3, cluster1 = Conv2d(C, K, 1x1), cluster2 = Conv2d(C, C) -> ReLU -> Conv2d(C, K)) and
// what autograd derives for the six parameters.  On the channels-last token matrix the 1x1 convolutions are GEMMs:
//     code = (X * m1) W1^T + b1 + relu((X * m2) W21^T + b21) W22^T + b22          feats_out = X * m3
// with m* the per-(image, channel) scales of nn.Dropout2d.  Two kernels do all of it:
//   * head_gemm_kernel  C[M, N] = epilogue(A'[M, K] B[N, K]^T): 128 x 128 tile per workgroup, 4 waves (one 64 x 64 quadrant each,
//     v_mfma_f32_32x32x16_f16), operands read as fp32, scaled by the dropout mask of their image (A side), split into fp16 hi + lo and
//     staged in LDS in 32-channel stages (hi*hi + hi*lo + lo*hi, fp32 accumulate: the fp32-class scheme of the loss kernels);
//     epilogues: + bias (+ bias), + bias -> ReLU, += (accumulate into C), * 1[aux > 0]; the first N tile of the cluster1 GEMM also
//     writes feats_out = X * m3 while it has X in registers (the dropped-out feature map of modules.py:116);
//   * head_wgrad_kernel dW[N, C] = sum_t G'[t, N]^T X'[t, C] over a range of tokens (split over workgroups; partial tiles + a
//     reduction kernel, fixed order: bitwise repeatable): both operands are transposed on their way into LDS (the reduction index -
//     the token - must be contiguous per MFMA lane), the dropout mask rides on X'; the first channel tile also sums the columns of
//     G' (the bias gradients).
// Forward: 3 launches (cluster1; cluster2[0] + ReLU -> H; cluster2[2] accumulated into code).  Backward: dHpre = (G W22) * 1[H > 0],
// three weight-gradient GEMMs, one reduction.
#include <hip/hip_runtime.h>

#include "../../include/stego_head.h"
#include "corr_common.h"

namespace stego {

typedef unsigned int hu32x4 __attribute__((ext_vector_type(4)));

constexpr int HT = 128;                          // tile rows / cols
constexpr int HKS = 32;                          // channels (or tokens) per stage
constexpr int HSIDE = 16384;                     // one operand stage: [hi 128 x 64 B][lo 128 x 64 B]

__device__ __forceinline__ int hswz(int r, int u) { return r * 64 + ((u ^ ((r >> 2) & 3)) << 4); }

// one 32-deep stage: a.b ~= ah.bh + ah.bl + al.bh   (rows of As = output rows, rows of Bs = output columns)
__device__ __forceinline__ void head_mma_stage(const unsigned char* __restrict__ As, const unsigned char* __restrict__ Bs,
                                               f32x16 (&acc)[2][2], int lane, int wr, int wc)
{
    const int r = lane & 31, half = lane >> 5;
    const int ra0 = 64 * wr + r, ra1 = ra0 + 32, rb0 = 64 * wc + r, rb1 = rb0 + 32;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int u = 2 * ks + half;
        const f16x8 ah0 = *reinterpret_cast<const f16x8*>(As + hswz(ra0, u)), al0 = *reinterpret_cast<const f16x8*>(As + 8192 + hswz(ra0, u));
        const f16x8 ah1 = *reinterpret_cast<const f16x8*>(As + hswz(ra1, u)), al1 = *reinterpret_cast<const f16x8*>(As + 8192 + hswz(ra1, u));
        const f16x8 bh0 = *reinterpret_cast<const f16x8*>(Bs + hswz(rb0, u)), bl0 = *reinterpret_cast<const f16x8*>(Bs + 8192 + hswz(rb0, u));
        const f16x8 bh1 = *reinterpret_cast<const f16x8*>(Bs + hswz(rb1, u)), bl1 = *reinterpret_cast<const f16x8*>(Bs + 8192 + hswz(rb1, u));
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, bh0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, bh1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1, bh0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1, bh1, acc[1][1], 0, 0, 0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bl0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bl1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bl0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bl1, acc[1][1], 0, 0, 0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bh0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bh1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bh0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bh1, acc[1][1], 0, 0, 0);
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
    const float* maskA;        // [B, K] or null: A'[m][k] = A[m][k] * maskA[b][k]
    const float* B;            // weights [N, K] row-major (ldb), or [K, N] when b_transposed
    int ldb, b_transposed;
    float* C;                  // [M, N] row-major (ldc)
    int ldc;
    const float* bias;         // [N] or null
    const float* bias2;        // [N] or null
    const float* aux;          // EPI_MASK_POS: [M, N] (ldaux)
    int ldaux;
    float* feats_out;          // [M, K] dense or null: = A * mask3 (written by the N tile 0)
    const float* mask3;        // [B, K] or null (null with feats_out: plain copy)
    const unsigned* amax_a;    // largest |A| / |B| (device words, see head_scale) or null
    const unsigned* amax_b;
    unsigned* amax_out;        // or null: atomicMax of |C| (the operand scale of whoever consumes C)
    int M, N, K, epi;
};

// KVEC: K % 32 == 0 and 16-byte aligned rows -> float4 staging; otherwise scalar staging with bounds (the K = cfg.dim GEMM)
template <bool KVEC, int EPI>
__global__ void __launch_bounds__(256) head_gemm_kernel(const HeadGemmParams prm)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * HSIDE];
    unsigned char* As = lds;
    unsigned char* Bs = lds + HSIDE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int m0 = blockIdx.x * HT, n0 = blockIdx.y * HT;
    const int row = tid >> 1, h = tid & 1;
    // my A row (rows / columns beyond the matrix are read from the last valid one and zeroed: every load below is unconditional -
    // a load inside a branch gets its own s_waitcnt and the staging becomes a chain of dependent round trips)
    const int m = m0 + row;
    const bool m_ok = m < prm.M;
    const int mc = m_ok ? m : prm.M - 1;
    const int b = mc / prm.HW, t = mc - b * prm.HW;
    const float* arow = prm.A + (long long)b * prm.a_img + (long long)t * prm.a_tok;
    const float* mrow = prm.maskA ? prm.maskA + (size_t)b * prm.K : nullptr;
    const float* m3row = prm.mask3 ? prm.mask3 + (size_t)b * prm.K : nullptr;
    float* frow = (prm.feats_out && blockIdx.y == 0 && m_ok) ? prm.feats_out + (size_t)m * prm.K : nullptr;
    // my B row (an output column)
    const int n = n0 + row;
    const bool n_ok = n < prm.N;
    const int nc = n_ok ? n : prm.N - 1;
    const float a_keep = m_ok ? 1.f : 0.f, b_keep = n_ok ? 1.f : 0.f;
    const size_t b_kstride = prm.b_transposed ? (size_t)prm.ldb : 1, b_nstride = prm.b_transposed ? 1 : (size_t)prm.ldb;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const float sa = head_scale(prm.amax_a), sb = head_scale(prm.amax_b);
    const float unscale = 1.f / (sa * sb);
    float va[16], vb[16];
    auto load_stage = [&](int k0) {
        const int kk = k0 + 16 * h;
        if constexpr (KVEC) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 x = *reinterpret_cast<const f32x4*>(arow + kk + 4 * q);
                const f32x4 w = *reinterpret_cast<const f32x4*>(prm.B + (size_t)nc * prm.ldb + kk + 4 * q);
                if (frow) {                      // (uniform per thread over the whole K loop)
                    f32x4 y = x;
                    if (m3row) y = y * *reinterpret_cast<const f32x4*>(m3row + kk + 4 * q);
                    *reinterpret_cast<f32x4*>(frow + kk + 4 * q) = y;
                }
                if (mrow) x = x * *reinterpret_cast<const f32x4*>(mrow + kk + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) { va[4 * q + e] = x[e] * (sa * a_keep); vb[4 * q + e] = w[e] * (sb * b_keep); }
            }
        } else {
            float xa[16];
            if ((prm.K & 1) == 0 && ((prm.a_img | prm.a_tok) & 1) == 0) {           // 8-byte aligned rows: pairs
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const int kc = kk + e < prm.K ? kk + e : prm.K - 2;
                    const f32x2 v = *reinterpret_cast<const f32x2*>(arow + kc);
                    xa[e] = v[0];
                    xa[e + 1] = v[1];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) xa[e] = arow[kk + e < prm.K ? kk + e : prm.K - 1];
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = kk + e;
                const bool k_ok = k < prm.K;
                const int kc = k_ok ? k : prm.K - 1;
                float x = xa[e];
                if (mrow) x *= mrow[kc];
                const float w = prm.B[(size_t)kc * b_kstride + (size_t)nc * b_nstride];      // (one load: no select between two)
                const float kf = k_ok ? 1.f : 0.f;
                va[e] = x * (sa * a_keep * kf);
                vb[e] = w * (sb * b_keep * kf);
            }
        }
    };

    load_stage(0);
    for (int k0 = 0; k0 < prm.K; k0 += HKS) {
        __syncthreads();                         // the previous stage's MFMAs are done with the LDS
        head_commit16(As, row, h, va);
        head_commit16(Bs, row, h, vb);
        __syncthreads();
        if (k0 + HKS < prm.K) load_stage(k0 + HKS);          // in flight under the MFMAs
        head_mma_stage(As, Bs, acc, lane, wr, wc);
    }

    // epilogue straight from the accumulators: for a fixed register the lanes 0-31 hold 32 consecutive columns of one row.
    // Loads (accumulate / mask modes) are unconditional from clamped addresses, only the store is predicated.
    float omax = 0.f;
    // (accumulate / mask modes: all 64 old values first, in one round trip - a load per element right before its use is 64 round trips)
    f32x16 old[2][2];
    if (EPI == EPI_ACCUM || EPI == EPI_MASK_POS) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = n0 + 64 * wc + 32 * ni + (lane & 31);
            const int colc = col < prm.N ? col : prm.N - 1;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = m0 + 64 * wr + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const int rrc = rr < prm.M ? rr : prm.M - 1;
                    old[mi][ni][r] = EPI == EPI_ACCUM ? *reinterpret_cast<const volatile float*>(prm.C + (size_t)rrc * prm.ldc + colc)
                                                      : *reinterpret_cast<const volatile float*>(prm.aux + (size_t)rrc * prm.ldaux + colc);
                }
        }
    }
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int col = n0 + 64 * wc + 32 * ni + (lane & 31);
        const bool c_ok = col < prm.N;
        const int colc = c_ok ? col : prm.N - 1;
        float bs = 0.f;
        if (EPI == EPI_BIAS || EPI == EPI_BIAS_RELU) {
            if (prm.bias) bs += prm.bias[colc];
            if (prm.bias2) bs += prm.bias2[colc];
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = m0 + 64 * wr + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                float v = acc[mi][ni][r] * unscale;
                if (EPI == EPI_BIAS) v += bs;
                else if (EPI == EPI_BIAS_RELU) v = fmaxf(v + bs, 0.f);
                else if (EPI == EPI_ACCUM) v += old[mi][ni][r];
                else v *= old[mi][ni][r] > 0.f ? 1.f : 0.f;
                if (c_ok && rr < prm.M) {
                    prm.C[(size_t)rr * prm.ldc + col] = v;
                    omax = fmaxf(omax, fabsf(v));
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
    const float* maskX;        // [B, Kc] or null
    float* part;               // [splits][N][Kc] partial sums
    float* part_bias;          // [splits][N] partial column sums of G (written by channel tile 0) or null
    const unsigned* amax_g;    // largest |G| / |X| (see head_scale) or null
    const unsigned* amax_x;
    int M, N, Kc, splits;
};

__global__ void __launch_bounds__(256) head_wgrad_kernel(const HeadWgradParams prm)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * HSIDE];
    __shared__ float colsum[2][HT];
    unsigned char* Gs = lds;
    unsigned char* Xs = lds + HSIDE;
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
    float vg[16], vx[16];
    auto load_stage = [&](int st) {
        const int t0 = st * HKS + 16 * g;
        // my 16 tokens lie in at most two images (HW >= 16 is checked by the host): their bases and mask values once per stage
        const int b0 = min(t0 / prm.HW, Bimg - 1), b1 = min(b0 + 1, Bimg - 1);
        const int first1 = (b0 + 1) * prm.HW;                 // first token of the next image
        const float* x0 = prm.X + (long long)b0 * prm.x_img + ccl - (long long)b0 * prm.HW * prm.x_tok;
        const float* x1 = prm.X + (long long)b1 * prm.x_img + ccl - (long long)b1 * prm.HW * prm.x_tok;
        const float k0 = prm.maskX ? prm.maskX[(size_t)b0 * prm.Kc + ccl] : 1.f, k1 = prm.maskX ? prm.maskX[(size_t)b1 * prm.Kc + ccl] : 1.f;
        const float* gp = prm.G + ncl;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = t0 + e;
            const int mcl = m < prm.M ? m : prm.M - 1;
            const bool second = mcl >= first1;
            const float gv = gp[(size_t)mcl * prm.ldg];
            const float xv = (second ? x1 : x0)[(long long)mcl * prm.x_tok];
            // (zeroing by a 0 / 1 factor, not by a select: a value that is only used conditionally has its load sunk into a branch)
            const float okf = m < prm.M ? 1.f : 0.f;
            vg[e] = gv * (okf * n_keep);
            vx[e] = xv * ((second ? k1 : k0) * (sx * okf * c_keep));
        }
    };
    if (s_beg < s_end) load_stage(s_beg);
    for (int st = s_beg; st < s_end; ++st) {
        __syncthreads();
        if (blockIdx.y == 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) bsum += vg[e];
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) vg[e] *= sg;
        head_commit16(Gs, col, g, vg);           // row = my column, the 32 tokens of the stage are its "channels"
        head_commit16(Xs, col, g, vx);
        __syncthreads();
        if (st + 1 < s_end) load_stage(st + 1);
        head_mma_stage(Gs, Xs, acc, lane, wr, wc);
    }
    float* out = prm.part + (size_t)split * prm.N * prm.Kc;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int cc = c0 + 64 * wc + 32 * ni + (lane & 31);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nn = n0 + 64 * wr + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (nn < prm.N && cc < prm.Kc) out[(size_t)nn * prm.Kc + cc] = acc[mi][ni][r] * unscale;
            }
    }
    if (blockIdx.y == 0 && prm.part_bias) {
        colsum[g][col] = bsum;
        __syncthreads();
        if (g == 0 && n_ok) prm.part_bias[(size_t)split * prm.N + n] = colsum[0][col] + colsum[1][col];
    }
}

// dW[i] = sum_s part[s][i]  (fixed order);  db[n] = sum_s part_bias[s][n] (optionally into two outputs: b1 and b22 share their gradient)
__global__ void __launch_bounds__(256) head_reduce_kernel(const float* part, int splits, long long numel, float* dW,
                                                          const float* part_bias, int N, float* db, float* db_b)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < numel) {
        float s = 0.f;
        int k = 0;
        for (; k + 8 <= splits; k += 8) {            // eight loads in flight, summed in order
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = part[(size_t)(k + j) * numel + i];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
        }
        for (; k < splits; ++k) s += part[(size_t)k * numel + i];
        dW[i] = s;
    }
    if (part_bias && i < N) {
        float s = 0.f;
        for (int k = 0; k < splits; ++k) s += part_bias[(size_t)k * N + i];
        if (db) db[i] = s;
        if (db_b) db_b[i] = s;
    }
}

// ------------------------------------------------------------------------------------------------ host side
static int head_check(const StegoHeadDesc* d)
{
    if (!d) return STEGO_ERR_NULL;
    if (d->B <= 0 || d->HW <= 0 || d->C <= 0 || d->K <= 0) return STEGO_ERR_SHAPE;
    if (d->C % 32 != 0 || d->K > 128 || d->tok_stride < d->C || d->img_stride < (int64_t)d->HW * d->tok_stride) return STEGO_ERR_UNSUPPORTED;
    if (d->HW < 16) return STEGO_ERR_UNSUPPORTED;                                                   // (a stage of the weight gradient spans <= 2 images)
    if (d->tok_stride % 4 != 0 || d->img_stride % 4 != 0) return STEGO_ERR_UNSUPPORTED;            // 16-byte token rows
    if ((int64_t)d->B * d->HW >= (1ll << 31) / 2) return STEGO_ERR_UNSUPPORTED;
    return STEGO_OK;
}

static size_t round256(size_t v) { return (v + 255) / 256 * 256; }

static int wgrad_splits(const StegoHeadDesc* d, int N)
{
    const int tiles = ((N + HT - 1) / HT) * ((d->C + HT - 1) / HT);
    const long long stages = ((long long)d->B * d->HW + HKS - 1) / HKS;
    long long s = (2 * 256 + tiles - 1) / tiles;                     // ~two workgroups per compute unit
    if (s > stages) s = stages;
    return s < 1 ? 1 : (int)s;
}

static hipError_t launch_gemm(const HeadGemmParams& p, bool kvec, hipStream_t s)
{
    const dim3 grid((p.M + HT - 1) / HT, (p.N + HT - 1) / HT), block(256);
    if (kvec && p.epi == EPI_BIAS) hipLaunchKernelGGL((head_gemm_kernel<true, EPI_BIAS>), grid, block, 0, s, p);
    else if (kvec && p.epi == EPI_BIAS_RELU) hipLaunchKernelGGL((head_gemm_kernel<true, EPI_BIAS_RELU>), grid, block, 0, s, p);
    else if (kvec && p.epi == EPI_ACCUM) hipLaunchKernelGGL((head_gemm_kernel<true, EPI_ACCUM>), grid, block, 0, s, p);
    else if (!kvec && p.epi == EPI_MASK_POS) hipLaunchKernelGGL((head_gemm_kernel<false, EPI_MASK_POS>), grid, block, 0, s, p);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace stego

using namespace stego;

extern "C" {

// operand-scale words (bits of the largest magnitude, see head_scale): the forward fills 0-4 and hands them to the backward
enum { HS_X = 0, HS_W1, HS_W21, HS_W22, HS_H, HS_G, HS_DH, HS_COUNT = 8 };

size_t stego_head_fwd_workspace_bytes(const StegoHeadDesc* d)
{
    if (head_check(d) != STEGO_OK) return 0;
    return 256 + (d->nonlinear ? round256((size_t)d->B * d->HW * d->C * sizeof(float)) : 0);
}

size_t stego_head_bwd_workspace_bytes(const StegoHeadDesc* d)
{
    if (head_check(d) != STEGO_OK) return 0;
    const size_t M = (size_t)d->B * d->HW;
    size_t part = (size_t)wgrad_splits(d, d->K) * d->K * d->C, pbias = (size_t)wgrad_splits(d, d->K) * d->K;
    size_t total = 256 + round256(part * 4) + round256(pbias * 4);
    if (d->nonlinear) {
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
    if (workspace_bytes < (d->nonlinear && !saved_h ? stego_head_fwd_workspace_bytes(d) : 256)) return STEGO_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int M = d->B * d->HW;
    hipError_t e;
    // operand scales: the first 32 bytes of the workspace (or of saved_scales' home: the tail of saved_h, see the header)
    unsigned* sc = saved_h ? reinterpret_cast<unsigned*>(saved_h + (size_t)M * d->C) : static_cast<unsigned*>(workspace);
    if ((e = hipMemsetAsync(sc, 0, HS_COUNT * sizeof(unsigned), s)) != hipSuccess) return STEGO_ERR_HIP + (int)e;
    auto absmax = [&](const float* x, long long img, long long rowst, int rpi, long long rows, int cols, int which) -> hipError_t {
        int blocks = (int)((rows * cols / 4 + 256 * 4 - 1) / (256 * 4));
        blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
        hipLaunchKernelGGL(head_absmax_kernel, dim3(blocks), dim3(256), 0, s, x, img, rowst, rpi, rows, cols, sc + which);
        return hipGetLastError();
    };
    if ((e = absmax(tokens, d->img_stride, d->tok_stride, d->HW, M, d->C, HS_X)) != hipSuccess) return STEGO_ERR_HIP + (int)e;
    if ((e = absmax(w1, 0, d->C, d->K, d->K, d->C, HS_W1)) != hipSuccess) return STEGO_ERR_HIP + (int)e;
    if (d->nonlinear) {
        if ((e = absmax(w21, 0, d->C, d->C, d->C, d->C, HS_W21)) != hipSuccess) return STEGO_ERR_HIP + (int)e;
        if ((e = absmax(w22, 0, d->C, d->K, d->K, d->C, HS_W22)) != hipSuccess) return STEGO_ERR_HIP + (int)e;
    }
    HeadGemmParams p{};
    p.amax_a = sc + HS_X; p.amax_b = sc + HS_W1;
    p.A = tokens; p.a_img = d->img_stride; p.a_tok = d->tok_stride; p.HW = d->HW; p.M = M; p.K = d->C;
    // cluster1 (+ both output biases: cluster2[2]'s is added here so that the third GEMM only accumulates) (+ feats_out)
    p.maskA = mask1; p.B = w1; p.ldb = d->C; p.C = code; p.ldc = d->K; p.N = d->K;
    p.bias = b1; p.bias2 = d->nonlinear ? b22 : nullptr; p.epi = EPI_BIAS;
    p.feats_out = feats_out; p.mask3 = mask3;
    if ((e = launch_gemm(p, true, s)) != hipSuccess) return STEGO_ERR_HIP + (int)e;
    if (!d->nonlinear) return STEGO_OK;
    float* H = saved_h ? saved_h : reinterpret_cast<float*>(static_cast<unsigned char*>(workspace) + 256);
    // H = relu(cluster2[0](x * m2))
    p.maskA = mask2; p.B = w21; p.ldb = d->C; p.C = H; p.ldc = d->C; p.N = d->C;
    p.bias = b21; p.bias2 = nullptr; p.epi = EPI_BIAS_RELU; p.feats_out = nullptr; p.mask3 = nullptr;
    p.amax_b = sc + HS_W21; p.amax_out = sc + HS_H;
    if ((e = launch_gemm(p, true, s)) != hipSuccess) return STEGO_ERR_HIP + (int)e;
    // code += cluster2[2](H)
    HeadGemmParams q{};
    q.A = H; q.a_img = (long long)d->HW * d->C; q.a_tok = d->C; q.HW = d->HW; q.M = M; q.K = d->C;
    q.B = w22; q.ldb = d->C; q.C = code; q.ldc = d->K; q.N = d->K; q.epi = EPI_ACCUM;
    q.amax_a = sc + HS_H; q.amax_b = sc + HS_W22;
    if ((e = launch_gemm(q, true, s)) != hipSuccess) return STEGO_ERR_HIP + (int)e;
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
    {
        hipError_t e0;
        if (d->nonlinear) e0 = hipMemcpyAsync(scl, saved_h + (size_t)M * C, HS_COUNT * sizeof(unsigned), hipMemcpyDeviceToDevice, s);
        else e0 = hipMemsetAsync(scl, 0, HS_COUNT * sizeof(unsigned), s);
        if (e0 != hipSuccess) return STEGO_ERR_HIP + (int)e0;
        if ((e0 = hipMemsetAsync(scl + HS_G, 0, 2 * sizeof(unsigned), s)) != hipSuccess) return STEGO_ERR_HIP + (int)e0;
        auto absmax = [&](const float* x, long long img, long long rowst, int rpi, long long rows, int cols, int which) -> hipError_t {
            int blocks = (int)((rows * cols / 4 + 256 * 4 - 1) / (256 * 4));
            blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
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
        w.G = G; w.ldg = ldg; w.X = X; w.x_img = x_img; w.x_tok = x_tok; w.HW = d->HW; w.maskX = maskX;
        w.part = part; w.part_bias = pbias; w.M = M; w.N = N; w.Kc = C; w.splits = splits;
        w.amax_g = G == d_code ? scl + HS_G : scl + HS_DH;
        w.amax_x = X == tokens ? scl + HS_X : scl + HS_H;
        hipLaunchKernelGGL(head_wgrad_kernel, dim3((N + HT - 1) / HT, (C + HT - 1) / HT, splits), dim3(256), 0, s, w);
        hipError_t er = hipGetLastError();
        if (er != hipSuccess) return er;
        const long long numel = (long long)N * C;
        hipLaunchKernelGGL(head_reduce_kernel, dim3((unsigned)((numel + 255) / 256)), dim3(256), 0, s, part, splits, numel, dW,
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
    HeadGemmParams p{};
    p.A = d_code; p.a_img = (long long)d->HW * K; p.a_tok = K; p.HW = d->HW; p.M = M; p.K = K;
    p.B = w22; p.ldb = C; p.b_transposed = 1;                  // B[n = channel j][k] = w22[k][j]
    p.C = dH; p.ldc = C; p.N = C; p.epi = EPI_MASK_POS; p.aux = saved_h; p.ldaux = C;
    p.amax_a = scl + HS_G; p.amax_b = scl + HS_W22; p.amax_out = scl + HS_DH;
    if ((e = launch_gemm(p, false, s)) != hipSuccess) return STEGO_ERR_HIP + (int)e;
    // dW21 = dHpre^T (x * m2), db21 = colsum(dHpre)
    if ((e = wgrad(dH, C, C, tokens, d->img_stride, d->tok_stride, mask2, part_c, pbias_c, sc, dw21, db21, nullptr)) != hipSuccess)
        return STEGO_ERR_HIP + (int)e;
    return STEGO_OK;
}

}  // extern "C"
