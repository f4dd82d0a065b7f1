// Micro-benchmark behind DESIGN.md's forward restructuring (round 2): what one CU can pull with the tile kernel's
// access pattern (4-tap bilinear gathers of 256-byte channel slices out of channels-last 28x28x384 fp32 maps) and with
// linear LDS-DMA copies, by where the bytes live (one XCD's L2 / Infinity Cache / HBM), by waves per CU, by chunk order.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/gather_bw.hip -o tools/ubench/gather_bw
// Prints one JSON object per configuration.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <functional>
#include <string.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int HW = 28, C = 384, NPIX = HW * HW;
constexpr size_t IMG_FLOATS = (size_t)NPIX * C;          // 1.2 MB per image

struct GParams {
    const float* pool;       // n_img images
    const int* img_of_wg;    // [grid] image index of each workgroup
    const int* taps;         // [grid][128][4] pixel indices
    float* sink;             // [grid]
    int n_chunks;            // 6 (64 channels each)
    int rotate;              // chunk order rotated by the image index
    int iters;               // passes over the image set (same taps)
};

// Each wave: 4 points per load instruction (16 lanes x 16 B = one 256-byte channel slice of one tap), 8 "items" per
// chunk as in corr_tile_kernel; W waves split the 128 points of the set.  Two chunks in flight (double buffer).
template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64) gather_kernel(GParams p)
{
    __shared__ int4 tapo[128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int img = p.img_of_wg[blockIdx.x];
    const float* base = p.pool + (size_t)img * IMG_FLOATS;
    if (tid < 128) {
        const int* t = p.taps + ((size_t)blockIdx.x * 128 + tid) * 4;
        tapo[tid] = make_int4(t[0] * C, t[1] * C, t[2] * C, t[3] * C);
    }
    __syncthreads();
    constexpr int PTS_PER_WAVE_INSTR = 4;
    constexpr int ITEMS = 128 / (PTS_PER_WAVE_INSTR * WAVES);     // items per chunk per wave
    const int slot = lane & 15, prow = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int rot = p.rotate ? img % p.n_chunks : 0;
    for (int it = 0; it < p.iters; ++it) {
        f32x4 cur[ITEMS][4], nxt[ITEMS][4];
        auto issue = [&](f32x4 (&dst)[ITEMS][4], int t) {
            int tt = t + rot; if (tt >= p.n_chunks) tt -= p.n_chunks;
            const float* cb = base + tt * 64 + slot * 4;
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const int4 o = tapo[(j * WAVES + wave) * PTS_PER_WAVE_INSTR + prow];
                dst[j][0] = *reinterpret_cast<const f32x4*>(cb + o.x);
                dst[j][1] = *reinterpret_cast<const f32x4*>(cb + o.y);
                dst[j][2] = *reinterpret_cast<const f32x4*>(cb + o.z);
                dst[j][3] = *reinterpret_cast<const f32x4*>(cb + o.w);
            }
        };
        issue(cur, 0);
        for (int t = 0; t < p.n_chunks; ++t) {
            if (t + 1 < p.n_chunks) issue(nxt, t + 1);
#pragma unroll
            for (int j = 0; j < ITEMS; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc += cur[j][k];
#pragma unroll
            for (int j = 0; j < ITEMS; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) cur[j][k] = nxt[j][k];
        }
    }
    float s = acc[0] + acc[1] + acc[2] + acc[3];
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0) atomicAdd(p.sink + blockIdx.x, s);
}

// Linear LDS-DMA stream: every workgroup copies `bytes_per_wg` of its image into a 2 x 32 KB LDS ring.
template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64) ldsdma_kernel(GParams p, int kb_per_wg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int img = p.img_of_wg[blockIdx.x];
    const unsigned char* base = reinterpret_cast<const unsigned char*>(p.pool + (size_t)img * IMG_FLOATS);
    constexpr int STAGE_KB = 32;
    float s = 0.f;
    const int nstage = kb_per_wg / STAGE_KB;
    auto issue = [&](int st) {
        unsigned char* dst = smem + (st & 1) * STAGE_KB * 1024;
        const unsigned char* src = base + (size_t)st * STAGE_KB * 1024;
        for (int pc = wave; pc < STAGE_KB; pc += WAVES)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)pc * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(dst + pc * 1024), 16, 0, 0);
    };
    issue(0);
    for (int st = 0; st < nstage; ++st) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (st + 1 < nstage) issue(st + 1);
        s += reinterpret_cast<const float*>(smem + (st & 1) * STAGE_KB * 1024)[tid];
        __syncthreads();
    }
    if (tid == 0) p.sink[blockIdx.x] = s;
}

// Rendezvous probe: groups of `gsz` workgroups; each does `work_us`-ish of spinning on the clock, publishes an 8-byte
// {tag, value} granule (sc1 store) and waits until the granules of its whole group carry the tag; the group sum is
// written out.  Measures the exposed hand-off time (stamps on the 100 MHz clock).
__global__ void __launch_bounds__(256) rendezvous_kernel(unsigned long long* gran, float* out, unsigned long long* stamps,
                                                         int gsz, unsigned epoch, int work_ticks, int skew)
{
    const int tid = threadIdx.x;
    const int g = blockIdx.x / gsz;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const int mywork = work_ticks + (skew ? (int)((blockIdx.x * 2654435761u) % (unsigned)skew) : 0);
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < mywork) {}
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) {
        const float v = 1.0f + (float)blockIdx.x;
        __hip_atomic_store(gran + blockIdx.x, ((unsigned long long)epoch << 32) | __builtin_bit_cast(unsigned, v),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    float sum = 0.f;
    if (tid < 64) {
        const int lane = tid;
        bool ok;
        unsigned spins = 0;
        unsigned long long x = 0;
        do {
            ok = true;
            if (lane < gsz) {
                x = __hip_atomic_load(gran + g * gsz + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = (unsigned)(x >> 32) == epoch;
            }
            if (!__all(ok)) { __builtin_amdgcn_s_sleep(2); ++spins; }
        } while (!__all(ok) && spins < (1u << 20));
        float v = lane < gsz ? __builtin_bit_cast(float, (unsigned)x) : 0.f;
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        sum = v;
    }
    const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) { out[blockIdx.x] = sum; stamps[blockIdx.x * 4] = t0; stamps[blockIdx.x * 4 + 1] = t1; stamps[blockIdx.x * 4 + 2] = t2; }
}

static float time_launches(hipStream_t s, int reps, const std::function<void(int)>& launch)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(0); launch(1);
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s));
    for (int i = 0; i < reps; ++i) launch(i);
    CK(hipEventRecord(b, s));
    CK(hipStreamSynchronize(s));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1000.f / reps;
}

int main(int argc, char** argv)
{
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_khz\": %d}\n", prop.name, prop.multiProcessorCount, prop.clockRate);

    // image pool: 4 rings x 64 images (4 x 77 MB = 308 MB > Infinity Cache)
    const int RING = 4, IMGS = 64;
    float* pool;
    CK(hipMalloc(&pool, (size_t)RING * IMGS * IMG_FLOATS * 4));
    CK(hipMemset(pool, 0, (size_t)RING * IMGS * IMG_FLOATS * 4));
    const int GRID = 224;
    std::vector<int> taps((size_t)GRID * 128 * 4);
    srand(1);
    for (int w = 0; w < GRID; ++w)
        for (int q = 0; q < 128; ++q) {
            const int x0 = rand() % (HW - 1), y0 = rand() % (HW - 1);
            int* t = &taps[((size_t)w * 128 + q) * 4];
            t[0] = y0 * HW + x0; t[1] = y0 * HW + x0 + 1; t[2] = (y0 + 1) * HW + x0; t[3] = (y0 + 1) * HW + x0 + 1;
        }
    int *d_taps, *d_map;
    float* d_sink;
    CK(hipMalloc(&d_taps, taps.size() * 4));
    CK(hipMemcpy(d_taps, taps.data(), taps.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_map, GRID * 4 * RING));
    CK(hipMalloc(&d_sink, GRID * 4));
    CK(hipMemset(d_sink, 0, GRID * 4));

    // mappings workgroup -> image
    //  "xcd6"   : the 28 workgroups of an XCD (blockIdx % 8) share 5 images (L2-local reuse, ~6 sets per image)
    //  "spread" : workgroup w reads image (w * 7) % 32 (each image read from ~7 different XCDs, as negatives are now)
    //  "own"    : workgroup w reads image w % 64 ... every workgroup (almost) its own image: no reuse
    auto make_map = [&](const char* kind, int ring) {
        std::vector<int> m(GRID);
        for (int w = 0; w < GRID; ++w) {
            int img;
            if (!strcmp(kind, "xcd6")) img = (w % 8) + 8 * ((w / 8) % 5);
            else if (!strcmp(kind, "spread")) img = (w * 7 + w / 32) % 32;
            else img = w % IMGS;
            m[w] = ring * IMGS + img;
        }
        return m;
    };
    const char* kinds[3] = {"xcd6", "spread", "own"};
    for (int kind = 0; kind < 3; ++kind) {
        std::vector<int> all;
        for (int r = 0; r < RING; ++r) { auto m = make_map(kinds[kind], r); all.insert(all.end(), m.begin(), m.end()); }
        CK(hipMemcpy(d_map, all.data(), all.size() * 4, hipMemcpyHostToDevice));
        for (int cold = 0; cold < 2; ++cold)
            for (int rotate = 0; rotate < 2; ++rotate)
                for (int waves = 4; waves <= 16; waves *= 2) {
                    GParams p{pool, d_map, d_taps, d_sink, 6, rotate, 1};
                    auto launch = [&](int i) {
                        GParams q = p;
                        q.img_of_wg = d_map + (cold ? (i % RING) : 0) * GRID;
                        if (waves == 4) hipLaunchKernelGGL(gather_kernel<4>, dim3(GRID), dim3(256), 0, s, q);
                        else if (waves == 8) hipLaunchKernelGGL(gather_kernel<8>, dim3(GRID), dim3(512), 0, s, q);
                        else hipLaunchKernelGGL(gather_kernel<16>, dim3(GRID), dim3(1024), 0, s, q);
                    };
                    const float us = time_launches(s, 40, launch);
                    const double bytes = (double)GRID * 128 * 4 * C * 4;
                    printf("{\"test\": \"gather\", \"map\": \"%s\", \"ring_rotated\": %d, \"rotate_chunks\": %d, \"waves\": %d, \"us\": %.2f, "
                           "\"l1_GBps_total\": %.0f, \"GBps_per_cu\": %.1f}\n",
                           kinds[kind], cold, rotate, waves, us, bytes / us * 1e-3, bytes / us * 1e-3 / GRID);
                    fflush(stdout);
                }
        // linear LDS-DMA stream of 768 KB per workgroup (same bytes as one gathered side)
        for (int cold = 0; cold < 2; ++cold)
            for (int waves = 4; waves <= 8; waves *= 2) {
                GParams p{pool, d_map, d_taps, d_sink, 6, 0, 1};
                const int kb = 768;
                auto launch = [&](int i) {
                    GParams q = p;
                    q.img_of_wg = d_map + (cold ? (i % RING) : 0) * GRID;
                    if (waves == 4) hipLaunchKernelGGL(ldsdma_kernel<4>, dim3(GRID), dim3(256), 65536, s, q, kb);
                    else hipLaunchKernelGGL(ldsdma_kernel<8>, dim3(GRID), dim3(512), 65536, s, q, kb);
                };
                const float us = time_launches(s, 40, launch);
                const double bytes = (double)GRID * kb * 1024;
                printf("{\"test\": \"ldsdma\", \"map\": \"%s\", \"ring_rotated\": %d, \"waves\": %d, \"us\": %.2f, \"GBps_total\": %.0f, "
                       "\"GBps_per_cu\": %.1f}\n", kinds[kind], cold, waves, us, bytes / us * 1e-3, bytes / us * 1e-3 / GRID);
                fflush(stdout);
            }
    }

    // rendezvous: 224 workgroups in 7 groups of 32
    {
        unsigned long long *gran, *stamps;
        float* out;
        CK(hipMalloc(&gran, 256 * 8)); CK(hipMalloc(&stamps, 256 * 4 * 8)); CK(hipMalloc(&out, 256 * 4));
        std::vector<unsigned long long> hs(256 * 4);
        std::vector<float> ho(256);
        for (int skew = 0; skew <= 200; skew += 100) {        // ticks of 10 ns
            double worst = 0, mean = 0;
            int bad = 0;
            const int reps = 20;
            for (int r = 0; r < reps; ++r) {
                CK(hipMemsetAsync(gran, 0, 256 * 8, s));
                hipLaunchKernelGGL(rendezvous_kernel, dim3(224), dim3(256), 0, s, gran, out, stamps, 32, (unsigned)(r + 1), 500, skew);
                CK(hipStreamSynchronize(s));
                CK(hipMemcpy(hs.data(), stamps, 224 * 4 * 8, hipMemcpyDeviceToHost));
                CK(hipMemcpy(ho.data(), out, 224 * 4, hipMemcpyDeviceToHost));
                for (int g = 0; g < 7; ++g) {
                    unsigned long long last_pub = 0;
                    float expect = 0;
                    for (int i = 0; i < 32; ++i) { last_pub = std::max(last_pub, hs[(g * 32 + i) * 4 + 1]); expect += 1.0f + g * 32 + i; }
                    for (int i = 0; i < 32; ++i) {
                        const double d = (double)((long long)(hs[(g * 32 + i) * 4 + 2] - last_pub)) * 0.01;   // us after the last publisher
                        worst = std::max(worst, d); mean += d / (7 * 32 * reps);
                        if (ho[g * 32 + i] != expect) ++bad;
                    }
                }
            }
            printf("{\"test\": \"rendezvous\", \"groups\": 7, \"group_size\": 32, \"skew_us\": %.1f, \"after_last_publish_us_mean\": %.2f, "
                   "\"after_last_publish_us_worst\": %.2f, \"wrong_sums\": %d}\n", skew * 0.01, mean, worst, bad);
        }
    }
    return 0;
}
