// Unit check of the DPP / permlane-swap reductions used by corr_fused.hip (run on gfx950): prints mismatches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true)); }
__device__ __forceinline__ float sum8(float x) { x += dpp_f<0xB1>(x); x += dpp_f<0x4E>(x); x += dpp_f<0x141>(x); return x; }
__device__ __forceinline__ float sum32(float x)
{
    x = sum8(x);
    x += dpp_f<0x140>(x);
    // (inline asm: with the builtin, hipcc 7.2 adds the first result to itself - v_add v1, v1, v1 after v_permlane16_swap v1, v2)
    float a = x, b = x;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));      // rows (0,1) and (2,3) exchanged
    return a + b;
}
__device__ __forceinline__ float sum64(float x)
{
    x = sum32(x);
    float a = x, b = x;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}
__global__ void k(const float* in, float* o8, float* o32, float* o64)
{
    const float x = in[threadIdx.x];
    o8[threadIdx.x] = sum8(x); o32[threadIdx.x] = sum32(x); o64[threadIdx.x] = sum64(x);
}
int main()
{
    float h[64], r8[64], r32[64], r64[64], *d, *a, *b, *c;
    for (int i = 0; i < 64; ++i) h[i] = (float)(1 << (i % 20)) + i;
    hipMalloc(&d, 256); hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&c, 256);
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, a, b, c);
    hipMemcpy(r8, a, 256, hipMemcpyDeviceToHost); hipMemcpy(r32, b, 256, hipMemcpyDeviceToHost); hipMemcpy(r64, c, 256, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        double e8 = 0, e32 = 0, e64 = 0;
        for (int j = 0; j < 64; ++j) { if (j / 8 == i / 8) e8 += h[j]; if (j / 32 == i / 32) e32 += h[j]; e64 += h[j]; }
        if (fabs(r8[i] - e8) > 1e-3 * e8 || fabs(r32[i] - e32) > 1e-3 * e32 || fabs(r64[i] - e64) > 1e-3 * e64) {
            if (bad++ < 8) printf("lane %d: sum8 %g (%g) sum32 %g (%g) sum64 %g (%g)\n", i, r8[i], e8, r32[i], e32, r64[i], e64);
        }
    }
    printf("dpp_sums: %d bad lanes\n", bad);
    return bad != 0;
}
