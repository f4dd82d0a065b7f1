// The RNG draws of ContrastiveCorrelationLoss.forward (modules.py:355-367, 382-385) as the device generator emits
// them -> the tensors the loss consumes, in ONE launch instead of ~9 elementwise kernels:
//   coords = u * 2 - 1              (modules.py:366-367; u = torch.rand: x * 2 is exact, then one rounding: an fma is the same)
//   perm[perm == arange(B)] += 1;  perm % B     (super_perm, modules.py:307-311), for every negative.
#include <hip/hip_runtime.h>

namespace stego {

struct DrawParams {
    const float* u1;
    const float* u2;
    float* c1;
    float* c2;
    long long n_coord;                 // floats per coords tensor
    const long long* raw[16];          // the n_neg randperm results
    long long* perms;                  // [n_neg][B]
    int n_neg, B;
};

__global__ void __launch_bounds__(256) finish_draws_kernel(const DrawParams prm)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < prm.n_coord) {
        prm.c1[i] = __builtin_fmaf(prm.u1[i], 2.f, -1.f);
        prm.c2[i] = __builtin_fmaf(prm.u2[i], 2.f, -1.f);
    }
    if (i < (long long)prm.n_neg * prm.B) {
        const int n = (int)(i / prm.B), b = (int)(i - (long long)n * prm.B);
        long long p = prm.raw[n][b];
        if (p == b) p += 1;
        prm.perms[i] = p % prm.B;
    }
}

hipError_t launch_finish_draws(const float* u1, const float* u2, long long n_coord, const long long* const* raw, int n_neg,
                               int B, float* c1, float* c2, long long* perms, hipStream_t stream)
{
    DrawParams prm{};
    prm.u1 = u1; prm.u2 = u2; prm.c1 = c1; prm.c2 = c2; prm.n_coord = n_coord; prm.perms = perms; prm.n_neg = n_neg; prm.B = B;
    for (int i = 0; i < n_neg; ++i) prm.raw[i] = raw[i];
    long long n = n_coord > (long long)n_neg * B ? n_coord : (long long)n_neg * B;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(finish_draws_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, prm);
    return hipGetLastError();
}

}  // namespace stego
