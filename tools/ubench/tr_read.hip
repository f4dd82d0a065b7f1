// What ds_read_b64_tr_b16 returns: a 16 x 16 tile of halves in LDS holds value 16 * row + col; lane l of each 16-lane group passes the
// address of row (l >> 2) + 4 * pass, columns 4 * (l & 3) .. + 3 (8 bytes); prints the four halves every lane receives.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half_t;
__global__ void k(float* out, int variant)
{
    __shared__ __attribute__((aligned(16))) half_t t[16 * 16];
    const int l = threadIdx.x;
    for (int i = l; i < 256; i += 64) t[i] = (half_t)(float)i;
    __syncthreads();
    const int g = l & 15;
    int row, col;
    if (variant == 0) { row = g >> 2; col = 4 * (g & 3); }
    else { row = g & 3; col = 4 * (g >> 2); }
    const unsigned addr = (unsigned)(reinterpret_cast<uintptr_t>(&t[row * 16 + col]));
    unsigned long long v = 0;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    half_t h[4];
    __builtin_memcpy(h, &v, 8);
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = (float)h[e];
}
int main()
{
    float* d; hipMalloc(&d, 64 * 4 * 4);
    float h[256];
    for (int variant = 0; variant < 2; ++variant) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, variant);
        hipError_t e = hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        if (e != hipSuccess) printf("error %s\n", hipGetErrorString(e));
        printf("variant %d (lane: 4 values as row.col)\n", variant);
        for (int l = 0; l < 20; ++l) {
            printf("  lane %2d:", l);
            for (int e = 0; e < 4; ++e) printf(" %2d.%-2d", (int)h[l * 4 + e] / 16, (int)h[l * 4 + e] % 16);
            printf("\n");
        }
    }
    return 0;
}
