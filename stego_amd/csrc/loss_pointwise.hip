// The elementwise part of helper() for shapes the fused kernels do not take (gfx950): given the two correlation tensors of ALL pair-sets,
//   reference src/modules.py:330-345:   old = fd.mean(); fd -= fd.mean([3, 4]); fd = fd - fd.mean() + old      (cfg.pointwise, per helper call)
//                                       loss = -clamp(cd, min_val[, 0.8]) * (fd - shift)
// in three launches instead of ~25 torch kernels over 59 MB tensors (cfg.feature_samples = 16, B = 32):
//   stego_rowsum              rowsum[n][i] = sum_j fd[n][i][j]            (one wave per row; the per-set old_mean is a [sets, B P] reduction of it)
//   stego_loss_pointwise_fwd  loss (negative sets only: the positives return their mean) + the row sums of the loss of every set
//   stego_loss_pointwise_bwd  d cd = -(fd - rowmean + old_mean - shift) * 1[cmin <= cd <= cmax] * upstream      (fd carries no gradient: no_grad, :326)
// fd_centred = (fd - rowmean) + old_mean: the reference's middle term fd.mean() of the row-centred tensor is zero up to rounding (SURVEY.md 8 a7:
// measured 4.7e-12) and is dropped.  Used by stego_amd.modules.ContrastiveCorrelationLoss.generic_forward.
#include "corr_common.h"
#include "host_util.h"
#include "../../include/stego_corr.h"

namespace stego {

__global__ void __launch_bounds__(256) rowsum_kernel(const float* __restrict__ x, float* __restrict__ out, long long rows, int P)
{
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* p = x + r * P;
    float s = 0.f;
    for (int j = lane; j < P; j += 64) s += p[j];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0) out[r] = s;
}

struct LossPwParams {
    const float* fd;            // [n_sets][B][P][P]
    const float* cd;
    const float* rowsum;        // [n_sets][B][P]
    const float* old_mean;      // [n_sets] (zeros when not pointwise)
    float* neg_loss;            // [n_sets - 2][B][P][P]
    float* loss_rowsum;         // [n_sets][B][P]
    const float* g_neg;         // backward: upstream of neg_loss [n_sets - 2][B][P][P] or null
    const float* g_sums;        // backward: upstream of the per-set loss sums [n_sets]
    float* g_cd;                // backward: [n_sets][B][P][P]
    const float* g_neg_bcast;   // backward: ONE device float added to every negative element's upstream (an expanded scalar, e.g. of .mean()) or null
    float shift[3];
    float cmin, cmax;
    int n_sets, B, P, pointwise;
};

template <bool BWD>
__global__ void __launch_bounds__(256) loss_pointwise_kernel(const LossPwParams p)
{
    const int lane = threadIdx.x & 63;
    const long long rows = (long long)p.n_sets * p.B * p.P;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int s = (int)(r / ((long long)p.B * p.P));
    const float shift = p.shift[s < 2 ? s : 2];
    const float rm = p.pointwise ? p.rowsum[r] / (float)p.P : 0.f;
    const float om = p.pointwise ? p.old_mean[s] : 0.f;
    const float* fd = p.fd + r * p.P;
    const float* cd = p.cd + r * p.P;
    const long long rneg = r - 2ll * p.B * p.P;
    if constexpr (!BWD) {
        float acc = 0.f;
        for (int j = lane; j < p.P; j += 64) {
            const float c = cd[j];
            const float l = -fminf(fmaxf(c, p.cmin), p.cmax) * (((fd[j] - rm) + om) - shift);
            if (s >= 2) p.neg_loss[rneg * p.P + j] = l;
            acc += l;
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
        if (lane == 0) p.loss_rowsum[r] = acc;
    } else {
        const float gs = p.g_sums ? p.g_sums[s] : 0.f;
        for (int j = lane; j < p.P; j += 64) {
            const float c = cd[j];
            float up = gs;
            if (s >= 2) up += (p.g_neg_bcast ? p.g_neg_bcast[0] : 0.f) + (p.g_neg ? p.g_neg[rneg * p.P + j] : 0.f);
            p.g_cd[r * p.P + j] = (c >= p.cmin && c <= p.cmax) ? -(((fd[j] - rm) + om) - shift) * up : 0.f;
        }
    }
}

hipError_t launch_rowsum(const float* x, float* out, long long rows, int P, hipStream_t stream)
{
    hipLaunchKernelGGL(rowsum_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, out, rows, P);
    return hipGetLastError();
}

static int check_pw(const void* a, const void* b, int32_t n_sets, int32_t B, int32_t P)
{
    if (!a || !b) return STEGO_ERR_NULL;
    if (n_sets < 2 || B < 1 || P < 1 || (long long)n_sets * B * P >= (1ll << 31) / 4) return STEGO_ERR_SHAPE;
    return STEGO_OK;
}

}  // namespace stego

using namespace stego;

extern "C" {

int stego_rowsum(const float* x, int64_t rows, int32_t P, float* out, stego_stream_t stream)
{
    if (!x || !out) return STEGO_ERR_NULL;
    if (rows < 0 || P < 1 || rows >= (1ll << 33)) return STEGO_ERR_SHAPE;
    if (rows == 0) return STEGO_OK;
    hipLaunchKernelGGL(rowsum_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), x, out, (long long)rows, P);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? STEGO_OK : STEGO_ERR_HIP + (int)e;
}

int stego_loss_pointwise_fwd(const float* fd, const float* cd, const float* rowsum, const float* old_mean, int32_t n_sets, int32_t B, int32_t P,
                             const float shift[3], float clamp_min, float clamp_max, int32_t pointwise, float* neg_loss, float* loss_rowsum,
                             stego_stream_t stream)
{
    int rc = check_pw(fd, cd, n_sets, B, P);
    if (rc != STEGO_OK) return rc;
    if (!loss_rowsum || !shift || (pointwise && (!rowsum || !old_mean)) || (n_sets > 2 && !neg_loss)) return STEGO_ERR_NULL;
    LossPwParams p{};
    p.fd = fd; p.cd = cd; p.rowsum = rowsum; p.old_mean = old_mean; p.neg_loss = neg_loss; p.loss_rowsum = loss_rowsum;
    p.shift[0] = shift[0]; p.shift[1] = shift[1]; p.shift[2] = shift[2];
    p.cmin = clamp_min; p.cmax = clamp_max; p.n_sets = n_sets; p.B = B; p.P = P; p.pointwise = pointwise;
    const long long rows = (long long)n_sets * B * P;
    hipLaunchKernelGGL(loss_pointwise_kernel<false>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? STEGO_OK : STEGO_ERR_HIP + (int)e;
}

int stego_loss_pointwise_bwd(const float* fd, const float* cd, const float* rowsum, const float* old_mean, int32_t n_sets, int32_t B, int32_t P,
                             const float shift[3], float clamp_min, float clamp_max, int32_t pointwise, const float* g_neg_loss,
                             const float* g_neg_bcast, const float* g_sums, float* g_cd, stego_stream_t stream)
{
    int rc = check_pw(fd, cd, n_sets, B, P);
    if (rc != STEGO_OK) return rc;
    if (!g_cd || !shift || (pointwise && (!rowsum || !old_mean))) return STEGO_ERR_NULL;
    LossPwParams p{};
    p.fd = fd; p.cd = cd; p.rowsum = rowsum; p.old_mean = old_mean; p.g_neg = g_neg_loss; p.g_sums = g_sums; p.g_cd = g_cd;
    p.g_neg_bcast = g_neg_bcast;
    p.shift[0] = shift[0]; p.shift[1] = shift[1]; p.shift[2] = shift[2];
    p.cmin = clamp_min; p.cmax = clamp_max; p.n_sets = n_sets; p.B = B; p.P = P; p.pointwise = pointwise;
    const long long rows = (long long)n_sets * B * P;
    hipLaunchKernelGGL(loss_pointwise_kernel<true>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? STEGO_OK : STEGO_ERR_HIP + (int)e;
}

}
