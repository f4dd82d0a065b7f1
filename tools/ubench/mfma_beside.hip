// Round-6 micro-benchmark: what does work BESIDE the MFMAs cost when every compute unit multiplies?  One wave per SIMD (4 per workgroup, one
// workgroup per CU, every CU) runs 12-MFMA groups of v_mfma_f32_32x32x16_f16 on four accumulator chains; behind every MFMA it issues
//   V packed FMAs (v_pk_fma_f32) and / or one 16-byte LDS read;
// a second form puts the VALU work on a SECOND wave of each SIMD instead (8 waves per workgroup).  Reports ns per MFMA (events) for the whole
// chip and for ONE workgroup (the same code on one CU: the unthrottled rate).  If issue slots were the cost, up to ~7 VALU instructions per
// MFMA would be free (an MFMA holds the matrix core for 32 cycles, a VALU instruction issues in 4); if the chip's power budget is, everything adds.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/mfma_beside.hip -o tools/ubench/bin/mfma_beside
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int V, bool LDS, bool SPLIT, bool SCALAR = false>
__global__ void __launch_bounds__(512) k(float* sink, int iters)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[36864];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 9216; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = (float)(i & 7) * 0.01f;
    __syncthreads();
    const bool mfma_wave = wave < 4;
    f32x16 acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    f16x8 av, bv;
#pragma unroll
    for (int e = 0; e < 8; ++e) { av[e] = (_Float16)(0.001f * (lane + e)); bv[e] = (_Float16)(0.002f * (lane - e)); }
    f32x2 x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = f32x2{0.5f + 0.01f * (lane + j), 0.25f};
    const f32x2 m = {0.999f, 1.001f}, c = {1e-3f, -1e-3f};
    float y[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) y[j] = 0.5f + 0.01f * (lane + j);
    const _Float16* bp = reinterpret_cast<const _Float16*>(lds) + (lane & 31) * 72 + 8 * (lane >> 5);
    f16x8 fr = {};
    if (mfma_wave) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[q & 3], 0, 0, 0);
                if (!SPLIT) {
#pragma unroll
                    for (int j = 0; j < V; ++j) {
                        if (SCALAR) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y[j & 7]) : "v"(m[0]), "v"(c[0])); }
                        else x[j & 7] = __builtin_elementwise_fma(x[j & 7], m, c);
                    }
                }
                if (LDS) { fr = *reinterpret_cast<const f16x8*>(bp + (q & 3) * 32 * 72 + 16 * ((q >> 2) & 3)); asm volatile("" :: "v"(fr)); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if (SPLIT) {                             // the second wave of the SIMD: the same number of packed FMAs, no MFMA
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 12; ++q) {
#pragma unroll
                for (int j = 0; j < V; ++j) x[j & 7] = __builtin_elementwise_fma(x[j & 7], m, c);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) s += acc[a][lane & 15];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += x[j][0] + x[j][1] + y[j];
    if (s == 123.456f) sink[0] = s;
}

template <int V, bool LDS, bool SPLIT, bool SCALAR = false>
static void run(const char* name, int cus, float* sink)
{
    const int iters = 4000, threads = SPLIT ? 512 : 256;
    for (int grid : {1, cus}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k<V, LDS, SPLIT, SCALAR>), dim3(grid), dim3(threads), 0, 0, sink, iters);
        hipDeviceSynchronize();
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL((k<V, LDS, SPLIT, SCALAR>), dim3(grid), dim3(threads), 0, 0, sink, iters);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        printf("%-58s %3d workgroup(s): %6.1f ns per MFMA\n", name, grid, best * 1e6 / (iters * 12.0));
    }
}
int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float* sink; hipMalloc(&sink, 4);
    run<0, false, false>("MFMAs alone", cus, sink);
    run<2, false, false>("+ 2 v_pk_fma_f32 behind every MFMA (same wave)", cus, sink);
    run<4, false, false>("+ 4 v_pk_fma_f32 behind every MFMA (same wave)", cus, sink);
    run<6, false, false>("+ 6 v_pk_fma_f32 behind every MFMA (same wave)", cus, sink);
    run<2, false, false, true>("+ 2 v_fma_f32 (32-bit) behind every MFMA (same wave)", cus, sink);
    run<4, false, false, true>("+ 4 v_fma_f32 (32-bit) behind every MFMA (same wave)", cus, sink);
    run<6, false, false, true>("+ 6 v_fma_f32 (32-bit) behind every MFMA (same wave)", cus, sink);
    run<8, false, false, true>("+ 8 v_fma_f32 (32-bit) behind every MFMA (same wave)", cus, sink);
    run<0, true, false>("+ one 16-byte LDS read behind every MFMA (waited for at once)", cus, sink);
    run<4, true, false>("+ 4 v_pk_fma_f32 + one LDS read (waited for) per MFMA", cus, sink);
    run<4, false, true>("+ 4 v_pk_fma_f32 per MFMA on a SECOND wave of the SIMD", cus, sink);
    run<6, false, true>("+ 6 v_pk_fma_f32 per MFMA on a SECOND wave of the SIMD", cus, sink);
    return 0;
}
