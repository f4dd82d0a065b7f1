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

// ------------------------------------------------------------------------------------------------ fast draws (opt-in)
// cfg.fast_draws: the SAME distributions as modules.py:366-367, 382-385 (coords uniform on the 2^-24 lattice of torch.rand,
// times 2 minus 1; one uniformly random permutation per negative, then the super_perm fix-up) from ONE kernel with its own
// counter-based generator (Philox-4x32-10, Salmon et al. 2011) keyed by 64 bits the caller drew from the torch generator.
// NOT the reference's random stream: 2 launches instead of ~30 (torch.randperm alone is 5 kernels), for training loops
// where the draws' host cost is visible (cached backbone tokens).  The default path keeps the reference's draws call for call.
namespace stego {

__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

struct FastDrawParams {
    const long long* seed;             // [1] device: 64 random bits
    float* c1;
    float* c2;
    long long n_coord;                 // floats per coords tensor
    long long* perms;                  // [n_neg][B]
    int n_neg, B;
};

// blocks [0, nb_coord): 4 floats of each coords tensor per thread (stream 0);  blocks nb_coord + n: permutation n (stream 1 + n)
__global__ void __launch_bounds__(256) fast_draws_kernel(const FastDrawParams prm, const int nb_coord)
{
    const unsigned long long seed = (unsigned long long)prm.seed[0];
    const unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
    if ((int)blockIdx.x < nb_coord) {
        const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
        if (i >= prm.n_coord) return;
        unsigned a[4] = {(unsigned)(i >> 2), (unsigned)(i >> 34), 0u, 0u};        // counter = (index, stream 0, which tensor)
        unsigned b[4] = {(unsigned)(i >> 2), (unsigned)(i >> 34), 0u, 1u};
        philox4x32_10(a, k0, k1);
        philox4x32_10(b, k0, k1);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (i + e < prm.n_coord) {
                prm.c1[i + e] = __builtin_fmaf((float)(a[e] >> 8) * 0x1p-24f, 2.f, -1.f);
                prm.c2[i + e] = __builtin_fmaf((float)(b[e] >> 8) * 0x1p-24f, 2.f, -1.f);
            }
        return;
    }
    // one permutation: 64-bit key per element, rank by counting (B^2 / 256 comparisons per thread; B is a batch size)
    extern __shared__ unsigned long long keys[];
    const int n = (int)blockIdx.x - nb_coord, B = prm.B;
    for (int i = threadIdx.x; i < B; i += 256) {
        unsigned c[4] = {(unsigned)i, 0u, 1u + (unsigned)n, 2u};
        philox4x32_10(c, k0, k1);
        keys[i] = ((unsigned long long)c[0] << 32) | c[1];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < B; i += 256) {
        const unsigned long long ki = keys[i];
        int rank = 0;
        for (int j = 0; j < B; ++j) rank += (keys[j] < ki || (keys[j] == ki && j < i)) ? 1 : 0;
        // element i goes to position rank: perm[rank] = i is a uniformly random permutation; then super_perm's fix-up
        long long p = i;
        if (p == rank) p += 1;
        prm.perms[(size_t)n * B + rank] = p % B;
    }
}

hipError_t launch_fast_draws(const long long* seed, long long n_coord, int n_neg, int B, float* c1, float* c2, long long* perms,
                             hipStream_t stream)
{
    FastDrawParams prm{};
    prm.seed = seed; prm.c1 = c1; prm.c2 = c2; prm.n_coord = n_coord; prm.perms = perms; prm.n_neg = n_neg; prm.B = B;
    const int nb_coord = (int)((n_coord + 1023) / 1024);
    if (nb_coord + n_neg == 0) return hipSuccess;
    hipLaunchKernelGGL(fast_draws_kernel, dim3(nb_coord + n_neg), dim3(256), (size_t)B * 8, stream, prm, nb_coord);
    return hipGetLastError();
}

}  // namespace stego
