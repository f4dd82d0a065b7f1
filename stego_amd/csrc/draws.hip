// The RNG draws of ContrastiveCorrelationLoss.forward (modules.py:355-367, 382-385) as the device generator emits
// them -> the tensors the loss consumes, in ONE launch instead of ~9 elementwise kernels:
//   coords = u * 2 - 1              (modules.py:366-367; u = torch.rand: x * 2 is exact, then one rounding: an fma is the same)
//   perm[perm == arange(B)] += 1;  perm % B     (super_perm, modules.py:307-311), for every negative.
#include <hip/hip_runtime.h>
#include <cmath>

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

// ------------------------------------------------------------------------------------------------ the reference's draws, one launch
// stego_ref_draws: bit for bit what `torch.rand(shape) x 2` and `torch.randperm(B) x n_neg` (modules.py:366-367, :291-295, :383)
// produce on the DEVICE generator of PyTorch-ROCm from (seed, offset), in ONE kernel instead of ~30 - and the host advances the
// generator by the same amount, so a training run keeps the reference's random stream exactly.
//
// What is reproduced (ATen, aten/src/ATen/native/cuda/{DistributionTemplates.h, Randperm.cu, Randperm.cuh}; the engine is
// rocRAND's Philox-4x32-10 behind hiprand, restated below: key = seed, counter = (offset / 4, subsequence)):
//   * a distribution kernel runs `grid` blocks of 256 threads, thread t = subsequence t draws four 32-bit numbers per visit
//     k (counter offset / 4 + k) and element li = 4 k G + ii G + t (G = 256 grid) gets number ii of them; the generator
//     advances by 4 (ceil(numel / (4 G))) per call;
//   * torch.rand: u = 2^-32 + x 2^-32 in float (rocrand's (0, 1]), 1.0 folded to 0.0 (uniform_kernel's bound reversal);
//   * torch.randperm(n): keys = random_(INT_MIN, INT_MAX) = x % (2^32 - 1) + INT_MIN from one distribution call - on this build the
//     64-bit flavour of it: x = (number 0 << 32) | number 1, two elements per visit (measured: tools/exp/dbg_draws2.py) -, STABLE sort of
//     arange(n) by the low `bits` bits of the keys (bits = ceil(log2(n - (6 n^2 + 1) / (12 ln 0.9))): 13 at n = 32), then
//     randperm_handle_duplicate_keys: every island of equal masked keys is reshuffled (Fisher-Yates from its end) by the thread at
//     its first sorted position tid with subsequence tid of a second call (generator + n rounded up to 4).
// Three details are build-dependent (whether the grid is sized per element or per `unroll` elements; whether the uniform conversion is
// contracted into an fma; whether the keys come from the 32-bit flavour of random_): `variant` bits 0 / 1 / 2.  stego_amd/modules.py finds the variant that matches the installed torch
// by comparing with the real calls once per process, and keeps the torch calls if none does.
namespace stego {

struct RefDrawParams {
    float* c1;
    float* c2;
    long long n_coord;                 // floats per coords tensor
    long long* perms;                  // [n_neg][B]
    unsigned long long seed, offset;   // generator state before the first draw ...
    const long long* seed_ptr;         // ... or (stream capture) where it will be when the graph replays: *seed_ptr,
    const long long* offset_ptr;       //     *offset_ptr + offset  (ATen's PhiloxCudaState, unpacked like at::cuda::philox::unpack)
    int n_neg, B;
    int grid_coord;                    // blocks of a torch.rand call of n_coord elements
    int grid_keys;                     // blocks of the key draw of randperm(B)
    int bits;                          // key bits the sort looks at
    int fma;                           // uniform conversion contracted
    int keys32;                        // keys from the 32-bit flavour of random_ (4 per visit) instead of the 64-bit one (2 per visit)
    unsigned long long off_rand, off_keys, off_dup;     // generator advance per torch.rand / key draw / duplicate pass
};

struct PhiloxState {
    unsigned c[4];
    unsigned k0, k1;
    unsigned r[4];
    int sub;
};

__device__ __forceinline__ void philox_init(PhiloxState& s, unsigned long long seed, unsigned long long subsequence, unsigned long long offset)
{
    s.k0 = (unsigned)seed; s.k1 = (unsigned)(seed >> 32);
    const unsigned long long ctr = offset >> 2;
    s.c[0] = (unsigned)ctr; s.c[1] = (unsigned)(ctr >> 32);
    s.c[2] = (unsigned)subsequence; s.c[3] = (unsigned)(subsequence >> 32);
    s.sub = (int)(offset & 3);
    unsigned t[4] = {s.c[0], s.c[1], s.c[2], s.c[3]};
    philox4x32_10(t, s.k0, s.k1);
    s.r[0] = t[0]; s.r[1] = t[1]; s.r[2] = t[2]; s.r[3] = t[3];
}
__device__ __forceinline__ void philox_bump(PhiloxState& s)
{
    if (++s.c[0] == 0u && ++s.c[1] == 0u && ++s.c[2] == 0u) ++s.c[3];
    unsigned t[4] = {s.c[0], s.c[1], s.c[2], s.c[3]};
    philox4x32_10(t, s.k0, s.k1);
    s.r[0] = t[0]; s.r[1] = t[1]; s.r[2] = t[2]; s.r[3] = t[3];
}
__device__ __forceinline__ unsigned philox_next(PhiloxState& s)
{
    const unsigned v = s.sub == 0 ? s.r[0] : s.sub == 1 ? s.r[1] : s.sub == 2 ? s.r[2] : s.r[3];
    if (++s.sub == 4) { s.sub = 0; philox_bump(s); }
    return v;
}
// element li of a distribution kernel whose generator state was `offset` (a multiple of 4): number ii of visit k of thread t
// (UNROLL = 4: one 32-bit number per element; UNROLL = 2: two of them, (first << 32) | second)
template <int UNROLL>
__device__ __forceinline__ unsigned long long dist_value(unsigned long long seed, unsigned long long offset, long long li, long long G)
{
    const long long k = li / (UNROLL * G), r = li - k * UNROLL * G;
    const int ii = (int)(r / G);
    const long long t = r - (long long)ii * G;
    PhiloxState s;
    philox_init(s, seed, (unsigned long long)t, offset + 4ull * (unsigned long long)k);
    if (UNROLL == 4) return ii == 0 ? s.r[0] : ii == 1 ? s.r[1] : ii == 2 ? s.r[2] : s.r[3];
    return ii == 0 ? ((unsigned long long)s.r[0] << 32) | s.r[1] : ((unsigned long long)s.r[2] << 32) | s.r[3];
}

__device__ __forceinline__ float torch_uniform(unsigned x, int fma)
{
    const float c = 2.3283064e-10f;
    const float u = fma ? __builtin_fmaf((float)x, c, c) : c + __fmul_rn((float)x, c);
    return u == 1.0f ? 0.0f : u;
}

constexpr int REF_MAX_B = 2048;

// blocks [0, nb_coord): one element of each coords tensor per thread; block nb_coord + n: permutation n
__global__ void __launch_bounds__(256) ref_draws_kernel(const RefDrawParams prm_in, const int nb_coord)
{
    RefDrawParams prm = prm_in;
    if (prm.seed_ptr) {                 // graph-safe state: read where CUDAGraph::replay puts it
        prm.seed = (unsigned long long)*prm.seed_ptr;
        prm.offset += (unsigned long long)*prm.offset_ptr;
    }
    if ((int)blockIdx.x < nb_coord) {
        const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
        if (i >= prm.n_coord) return;
        const long long G = 256ll * prm.grid_coord;
        const float u1 = torch_uniform((unsigned)dist_value<4>(prm.seed, prm.offset, i, G), prm.fma);
        const float u2 = torch_uniform((unsigned)dist_value<4>(prm.seed, prm.offset + prm.off_rand, i, G), prm.fma);
        prm.c1[i] = __builtin_fmaf(u1, 2.f, -1.f);
        prm.c2[i] = __builtin_fmaf(u2, 2.f, -1.f);
        return;
    }
    __shared__ unsigned skey[REF_MAX_B];           // masked keys, then (after the sort) in sorted order
    __shared__ unsigned sorted_key[REF_MAX_B];
    __shared__ int sorted_val[REF_MAX_B];
    const int n = (int)blockIdx.x - nb_coord, B = prm.B;
    const unsigned long long off0 = prm.offset + 2 * prm.off_rand + (unsigned long long)n * (prm.off_keys + prm.off_dup);
    const unsigned mask = prm.bits >= 32 ? 0xffffffffu : ((1u << prm.bits) - 1u);
    const long long G = 256ll * prm.grid_keys;
    for (int i = threadIdx.x; i < B; i += 256) {
        const unsigned long long x = prm.keys32 ? dist_value<4>(prm.seed, off0, i, G) : dist_value<2>(prm.seed, off0, i, G);
        // random_(INT_MIN, INT_MAX): (x % (2^32 - 1)) + INT_MIN as an int; the sort only sees its low bits
        const unsigned key = (unsigned)((long long)(x % 4294967295ull) + (long long)(-2147483647 - 1));
        skey[i] = key & mask;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < B; i += 256) {   // stable rank by counting
        const unsigned ki = skey[i];
        int rank = 0;
        for (int j = 0; j < B; ++j) rank += (skey[j] < ki || (skey[j] == ki && j < i)) ? 1 : 0;
        sorted_key[rank] = ki;
        sorted_val[rank] = i;
    }
    __syncthreads();
    // randperm_handle_duplicate_keys_kernel: thread tid at the first position of an island reshuffles it
    for (int tid = threadIdx.x; tid < B - 1; tid += 256) {
        if (sorted_key[tid] != sorted_key[tid + 1]) continue;
        if (tid != 0 && sorted_key[tid] == sorted_key[tid - 1]) continue;
        int island = 0;
        do { ++island; } while (tid + island < B && sorted_key[tid + island] == sorted_key[tid]);
        PhiloxState s;
        philox_init(s, prm.seed, (unsigned long long)tid, off0 + prm.off_keys);
        for (int i = island - 1; i > 0; --i) {
            const unsigned r = philox_next(s) % (unsigned)(i + 1);
            if ((unsigned)i != r) {
                const int tmp = sorted_val[tid + i];
                sorted_val[tid + i] = sorted_val[tid + r];
                sorted_val[tid + r] = tmp;
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < B; i += 256) {   // super_perm's fix-up (modules.py:293-295)
        long long p = sorted_val[i];
        if (p == i) p += 1;
        prm.perms[(size_t)n * B + i] = p % B;
    }
}

// grid of an ATen distribution kernel over `numel` elements (calc_execution_policy) and the generator advance it causes
static void torch_dist_policy(long long numel, int variant, int unroll, int cus, int threads_per_cu, int* grid, unsigned long long* advance)
{
    const long long per_block = (variant & 1) ? 256 * unroll : 256;
    long long g = (numel + per_block - 1) / per_block;
    const long long cap = (long long)cus * (threads_per_cu / 256);
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    *grid = (int)g;
    *advance = (unsigned long long)(((numel - 1) / (256 * g * unroll) + 1) * 4);
}

unsigned long long ref_draws_advance(long long n_coord, int n_neg, int B, int variant, int cus, int threads_per_cu)
{
    int g;
    unsigned long long a_rand = 0, a_keys = 0;
    if (n_coord > 0) torch_dist_policy(n_coord, variant, 4, cus, threads_per_cu, &g, &a_rand);
    torch_dist_policy(B, variant, (variant & 4) ? 4 : 2, cus, threads_per_cu, &g, &a_keys);
    const unsigned long long a_dup = ((unsigned long long)B + 3) / 4 * 4;
    return 2 * a_rand + (unsigned long long)n_neg * (a_keys + a_dup);
}

hipError_t launch_ref_draws(unsigned long long seed, unsigned long long offset, const long long* seed_ptr, const long long* offset_ptr,
                            int variant, long long n_coord, int n_neg, int B,
                            int cus, int threads_per_cu, float* c1, float* c2, long long* perms, hipStream_t stream)
{
    RefDrawParams prm{};
    prm.c1 = c1; prm.c2 = c2; prm.n_coord = n_coord; prm.perms = perms; prm.seed = seed; prm.offset = offset;
    prm.seed_ptr = seed_ptr; prm.offset_ptr = offset_ptr;
    prm.n_neg = n_neg; prm.B = B; prm.fma = (variant >> 1) & 1; prm.keys32 = (variant >> 2) & 1;
    prm.grid_coord = 1;
    if (n_coord > 0) torch_dist_policy(n_coord, variant, 4, cus, threads_per_cu, &prm.grid_coord, &prm.off_rand);
    torch_dist_policy(B, variant, prm.keys32 ? 4 : 2, cus, threads_per_cu, &prm.grid_keys, &prm.off_keys);
    prm.off_dup = ((unsigned long long)B + 3) / 4 * 4;
    {   // Randperm.cu: bits = min(64, ceil(log2(n - (6 n^2 + 1) / (12 ln 0.9))))
        const double nd = (double)B, t12 = std::log(0.9) * 12.0;
        int bits = (int)std::ceil(std::log2(nd - (6.0 * nd * nd + 1.0) / t12));
        prm.bits = bits > 64 ? 64 : bits;
    }
    const int nb_coord = (int)((n_coord + 255) / 256);
    if (nb_coord + n_neg == 0) return hipSuccess;
    hipLaunchKernelGGL(ref_draws_kernel, dim3(nb_coord + n_neg), dim3(256), 0, stream, prm, nb_coord);
    return hipGetLastError();
}

// ---- nn.Dropout2d's channel masks as ATen draws them (feature dropout: noise = x.new_empty(B, C, 1, 1).bernoulli_(1 - p).div_(1 - p)):
// bernoulli_(q) = (curand_uniform4 value < q) per element through the same distribution template as torch.rand (so the same
// grid policy / counter layout as above, one Philox block per 4 elements and visit), WITHOUT torch.rand's fold of 1.0 to 0.0;
// div_(q) multiplies by the float 1 / q.  n_masks consecutive calls, each advancing the generator by the policy's amount.
struct RefMaskParams {
    float* out;                        // [n_masks][numel]
    long long numel;
    unsigned long long seed, offset;
    const long long* seed_ptr;
    const long long* offset_ptr;
    int n_masks, grid, fma;
    unsigned long long adv;            // generator advance per mask
    float q, inv_q;
};

__global__ void __launch_bounds__(256) ref_masks_kernel(const RefMaskParams prm)
{
    unsigned long long seed = prm.seed, offset = prm.offset;
    if (prm.seed_ptr) {
        seed = (unsigned long long)*prm.seed_ptr;
        offset += (unsigned long long)*prm.offset_ptr;
    }
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= prm.numel) return;
    const int mk = blockIdx.y;
    const unsigned x = (unsigned)dist_value<4>(seed, offset + (unsigned long long)mk * prm.adv, i, 256ll * prm.grid);
    const float c = 2.3283064e-10f;
    const float u = prm.fma ? __builtin_fmaf((float)x, c, c) : c + __fmul_rn((float)x, c);
    prm.out[(size_t)mk * prm.numel + i] = u < prm.q ? prm.inv_q : 0.f;
}

unsigned long long ref_masks_advance(long long numel, int n_masks, int variant, int cus, int threads_per_cu)
{
    int g;
    unsigned long long a = 0;
    if (numel > 0) torch_dist_policy(numel, variant, 4, cus, threads_per_cu, &g, &a);
    return a * (unsigned long long)n_masks;
}

hipError_t launch_ref_masks(unsigned long long seed, unsigned long long offset, const long long* seed_ptr, const long long* offset_ptr,
                            int variant, int n_masks, long long numel, float keep_prob, int cus, int threads_per_cu, float* out,
                            hipStream_t stream)
{
    if (n_masks <= 0 || numel <= 0) return hipSuccess;
    RefMaskParams prm{};
    prm.out = out; prm.numel = numel; prm.seed = seed; prm.offset = offset; prm.seed_ptr = seed_ptr; prm.offset_ptr = offset_ptr;
    prm.n_masks = n_masks; prm.fma = (variant >> 1) & 1; prm.q = keep_prob; prm.inv_q = 1.0f / keep_prob;
    torch_dist_policy(numel, variant, 4, cus, threads_per_cu, &prm.grid, &prm.adv);
    hipLaunchKernelGGL(ref_masks_kernel, dim3((unsigned)((numel + 255) / 256), n_masks), dim3(256), 0, stream, prm);
    return hipGetLastError();
}

}  // namespace stego
