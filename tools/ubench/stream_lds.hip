// Round-5 micro-benchmark: how fast can ONE compute unit stream operand chunks into LDS, by mechanism?
// Every streaming kernel of this repo that stages through LDS with global_load_lds (the dense / KNN row-block kernels, the ViT GEMM, the loss's backward
// tiles) ends up at ~20-28 GB/s per CU (5-7 TB/s chip), while the fused forward's gather waves pull 71 GB/s per CU into REGISTERS with buffer loads.
// Is the LDS-DMA path the cap?  One workgroup per CU (256 threads) streams `chunks` chunks of 36 KB of cold data:
//   mode 0  global_load_lds_dwordx4 into a ring of `depth` slots (1 KB per wave instruction), one barrier per chunk   (what the kernels do)
//   mode 1  global_load_dwordx4 into registers, ds_write_b128 into the slot, one barrier per chunk, `depth` - 1 chunks of registers in flight
//   mode 2  global_load_dwordx4 into registers only (xor-folded), `depth` chunks in flight: the load path without LDS
// 2 workgroups per CU (wg2 = 1) halves the LDS per workgroup.  Reported: GB/s per CU and chip TB/s.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/stream_lds.hip -o tools/ubench/bin/stream_lds
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int CHUNK = 36864;              // 36 pieces of 1 KB: 9 per wave
constexpr int PIECES = CHUNK / 1024;

struct P {
    const unsigned char* src;     // [workgroup][chunks][CHUNK]
    unsigned* sink;
    int chunks, depth, mode;
    int streams;                  // > 0: workgroup w streams stream w % streams (data shared by the workgroups of an XCD: L2 / Infinity-Cache hits)
};

__device__ __forceinline__ void dma_piece(const unsigned char* gsrc_lane, unsigned lds_addr)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc_lane), "s"(lds_addr) : "memory");
}

template <int DEPTH>
__global__ void __launch_bounds__(256) stream_kernel(P p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned char* src = p.src + (size_t)(p.streams > 0 ? blockIdx.x % p.streams : blockIdx.x) * p.chunks * CHUNK;
    const unsigned smem_addr = (unsigned)(uintptr_t)((__attribute__((address_space(3))) const unsigned char*)smem);
    unsigned acc = 0;
    if (p.mode == 0) {
        auto issue = [&](int g) {
            const unsigned char* s = src + (size_t)g * CHUNK + lane * 16;
            const unsigned dst = smem_addr + (g % DEPTH) * CHUNK;
#pragma unroll
            for (int i = 0; i < PIECES / 4; ++i) dma_piece(s + (wave + 4 * i) * 1024, dst + (wave + 4 * i) * 1024);
        };
        for (int g = 0; g < DEPTH - 1 && g < p.chunks; ++g) issue(g);
        for (int g = 0; g < p.chunks; ++g) {
            const int ahead = min(p.chunks - 1 - g, DEPTH - 2);
            if (ahead >= 3) asm volatile("s_waitcnt vmcnt(27)" ::: "memory");
            else if (ahead == 2) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (g + DEPTH - 1 < p.chunks) issue(g + DEPTH - 1);
            acc ^= reinterpret_cast<const unsigned*>(smem + (g % DEPTH) * CHUNK)[tid];        // somebody reads the slot
        }
    } else if (p.mode == 1) {
        // registers hold ONE chunk share (9 x 16 B per lane) in flight; the slot ring decouples write and read
        u32x4 r[PIECES / 4];
        auto load = [&](int g) {
            const unsigned char* s = src + (size_t)g * CHUNK + lane * 16;
#pragma unroll
            for (int i = 0; i < PIECES / 4; ++i) r[i] = *reinterpret_cast<const u32x4*>(s + (wave + 4 * i) * 1024);
        };
        load(0);
        for (int g = 0; g < p.chunks; ++g) {
            unsigned char* dst = smem + (g & 1) * CHUNK + lane * 16;
#pragma unroll
            for (int i = 0; i < PIECES / 4; ++i) *reinterpret_cast<u32x4*>(dst + (wave + 4 * i) * 1024) = r[i];      // (waits for the loads)
            if (g + 1 < p.chunks) load(g + 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            acc ^= reinterpret_cast<const unsigned*>(smem + (g & 1) * CHUNK)[tid];
        }
    } else {
        u32x4 r[DEPTH][PIECES / 4];
        auto load = [&](int g, int slot) {
            const unsigned char* s = src + (size_t)g * CHUNK + lane * 16;
#pragma unroll
            for (int i = 0; i < PIECES / 4; ++i) r[slot][i] = *reinterpret_cast<const u32x4*>(s + (wave + 4 * i) * 1024);
        };
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) if (d < p.chunks) load(d, d);
        for (int g0 = 0; g0 < p.chunks; g0 += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
                for (int i = 0; i < PIECES / 4; ++i) acc ^= r[d][i][0] ^ r[d][i][3];
                if (g0 + DEPTH + d < p.chunks) load(g0 + DEPTH + d, d);
            }
        }
    }
    if (acc == 0x12345678u) p.sink[blockIdx.x] = acc;
}

template <int DEPTH>
static double run(P p, int wgs, int lds, hipStream_t s)
{
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_kernel<DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(stream_kernel<DEPTH>, dim3(wgs), dim3(256), lds, s, p);
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    return best;
}

int main()
{
    hipStream_t s;
    CK(hipStreamCreate(&s));
    const int chunks = 48;
    const int max_wgs = 512;
    const size_t bytes = (size_t)max_wgs * chunks * CHUNK;            // 906 MB: cold for every launch (> L2 + Infinity Cache)
    unsigned char* buf;
    CK(hipMalloc(&buf, bytes));
    CK(hipMemset(buf, 1, bytes));
    unsigned* sink;
    CK(hipMalloc(&sink, max_wgs * 4));
    const char* names[3] = {"global_load_lds (DMA)", "load -> registers -> ds_write", "load -> registers only"};
    for (int streams = 0; streams <= 64; streams = streams ? streams * 8 : 8)
    for (int wg2 = 0; wg2 < 2; ++wg2)
        for (int mode = 0; mode < 3; ++mode)
            for (int depth = 2; depth <= 4; ++depth) {
                if (mode == 1 && depth != 2) continue;
                if (wg2 && mode == 0 && depth > 2) continue;
                P p{buf, sink, chunks, depth, mode, streams};
                const int wgs = wg2 ? 512 : 256;
                const int lds = mode == 2 ? 1024 : (mode == 1 ? 2 : depth) * CHUNK;
                double ms = depth == 2 ? run<2>(p, wgs, lds, s) : depth == 3 ? run<3>(p, wgs, lds, s) : run<4>(p, wgs, lds, s);
                const double total = (double)wgs * chunks * CHUNK;
                printf("{\"streams\": %d, \"mode\": \"%s\", \"workgroups_per_cu\": %d, \"depth\": %d, \"us\": %.1f, \"chip_TBps\": %.2f, \"GBps_per_cu\": %.1f}\n", streams, names[mode], wg2 + 1,
                       depth, ms * 1e3, total / (ms * 1e-3) * 1e-12, total / (ms * 1e-3) * 1e-9 / 256.0);
                fflush(stdout);
            }
    return 0;
}
