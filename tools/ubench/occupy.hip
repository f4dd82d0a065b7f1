// Stand-in for a collective kernel that holds compute units while the loss runs (tools/exp/shared_contention.py):
// n_wg workgroups of 256 threads with lds_bytes of LDS each spin for `micros` on the given stream.
#include <hip/hip_runtime.h>
#include <cstdint>
__global__ void occupy_kernel(long long ticks, int* sink)
{
    extern __shared__ int lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    lds[threadIdx.x] = threadIdx.x;
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(32);
    if (lds[threadIdx.x] == -1) sink[0] = 1;
}
extern "C" int occupy_launch(int n_wg, int lds_bytes, int micros, void* sink, void* stream)
{
    static bool set = false;
    if (!set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; }
    hipLaunchKernelGGL(occupy_kernel, dim3(n_wg), dim3(256), lds_bytes, (hipStream_t)stream, (long long)micros * 100, (int*)sink);
    return (int)hipGetLastError();
}
