// Round-6 micro-benchmark: how long does ONE v_mfma_f32_32x32x16_f16 take in the shapes the fused forward issues it?
//   one MFMA wave per SIMD (4 per workgroup, + 8 waves parked at a barrier as the gather team of a self-correlation tile is),
//   NACC independent accumulator chains, operands from registers (no LDS), N instructions per wave, on G workgroups (1 .. every CU).
// Prints ns per MFMA per SIMD from s_memrealtime (100 MHz) inside the kernel and from HIP events around it.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/mfma_rate.hip -o tools/ubench/bin/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NACC, bool LDSREAD>
__global__ void __launch_bounds__(768) mfma_kernel(float* sink, unsigned long long* stamps, int iters)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[32768];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 8192; i += 768) reinterpret_cast<float*>(lds)[i] = (float)(i & 7) * 0.01f;
    __syncthreads();
    if (wave >= 4) { __syncthreads(); return; }          // parked, as a gather team without a gather
    f32x16 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    f16x8 av, bv;
#pragma unroll
    for (int e = 0; e < 8; ++e) { av[e] = (_Float16)(0.001f * (lane + e)); bv[e] = (_Float16)(0.002f * (lane - e)); }
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (LDSREAD) {
            // 12 fragment reads + 12 MFMAs: one feature stage of the half kernel
            f16x8 f[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) f[k] = *reinterpret_cast<const f16x8*>(lds + ((k * 1024 + (lane & 31) * 64 + ((lane >> 5) + 2 * (it & 1)) * 16) & 32767));
#pragma unroll
            for (int k = 0; k < 12; ++k) acc[k % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[k], f[(k + 5) % 12], acc[k % NACC], 0, 0, 0);
        } else {
#pragma unroll
            for (int k = 0; k < 12; ++k) acc[k % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[k % NACC], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < NACC; ++a) s += acc[a][lane & 15];
    if (s == 123.456f) sink[0] = s;
    if (lane == 0) stamps[blockIdx.x * 4 + wave] = t1 - t0;
    __syncthreads();
}

template <int NACC, bool L>
static void run(const char* name, int grid, int iters, float* sink, unsigned long long* stamps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ms;
    std::vector<unsigned long long> h((size_t)grid * 4);
    for (int r = 0; r < 6; ++r) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((mfma_kernel<NACC, L>), dim3(grid), dim3(768), 0, 0, sink, stamps, iters);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        if (r > 1) ms.push_back(t);
    }
    CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    std::sort(ms.begin(), ms.end());
    const double n = 12.0 * iters;
    printf("%-44s grid %3d: in-kernel %6.1f ns per MFMA (median wave; slowest %6.1f), kernel by events %7.1f us\n", name, grid,
           h[h.size() / 2] * 10.0 / n, h.back() * 10.0 / n, ms[ms.size() / 2] * 1e3);
}

int main()
{
    float* sink; unsigned long long* stamps;
    CK(hipMalloc(&sink, 64)); CK(hipMalloc(&stamps, 1024 * 4 * 8));
    for (int iters : {15, 150, 1500}) {
        printf("---- %d stages of 12 MFMAs per wave\n", iters);
        for (int grid : {1, 8, 256}) {
            run<2, false>("registers, 2 chains", grid, iters, sink, stamps);
            run<4, false>("registers, 4 chains", grid, iters, sink, stamps);
            run<2, true>("12 ds_read_b128 + 12 MFMA per stage, 2 chains", grid, iters, sink, stamps);
            run<4, true>("12 ds_read_b128 + 12 MFMA per stage, 4 chains", grid, iters, sink, stamps);
        }
    }
    return 0;
}
