// Stand-alone timing harness of dense_stream_kernel (csrc/dense_stream.hip is included as it is; compile-time ablations by -DDS_ABL=bits:
// 1 no fragment re-reads, 2 no conversion / loads, 4 no MFMAs, 8 no output stores, 16 no park).  Results are NOT checked here (tests do that).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I stego_amd/csrc -I include tools/ubench/dense_stream_bench.hip stego_amd/csrc/host_util.hip -o tools/ubench/bin/dense_stream_bench
//   dense_stream_bench [B] [C] [H] [iters]
#include "../../stego_amd/csrc/dense_stream.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace stego;
int main(int argc, char** argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 32, C = argc > 2 ? atoi(argv[2]) : 384, H = argc > 3 ? atoi(argv[3]) : 28, iters = argc > 4 ? atoi(argv[4]) : 30;
    const size_t P = (size_t)H * H, nmap = (size_t)B * P * C;
    std::vector<float> ha(nmap), hb(nmap);
    unsigned x = 12345u;
    for (size_t i = 0; i < nmap; ++i) { x = x * 1664525u + 1013904223u; ha[i] = ((x >> 8) & 0xffff) / 32768.f - 1.f; x = x * 1664525u + 1013904223u; hb[i] = ((x >> 8) & 0xffff) / 32768.f - 1.f; }
    float *a, *b, *out; void* ws;
    hipMalloc(&a, nmap * 4); hipMalloc(&b, nmap * 4); hipMalloc(&out, (size_t)B * P * P * 4); hipMalloc(&ws, dense_stream_workspace_bytes(B, (int)P) + 4096);
    hipMemcpy(a, ha.data(), nmap * 4, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), nmap * 4, hipMemcpyHostToDevice);
    MapV ma{a, (long long)(P * C), 1, H * C, C}, mb{b, (long long)(P * C), 1, H * C, C};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) launch_dense_stream(ma, mb, B, C, H, H, H, H, 1, out, ws, 0);
    hipDeviceSynchronize();
    std::vector<float> t(iters);
    for (int i = 0; i < iters; ++i) {
        hipEventRecord(e0, 0);
        launch_dense_stream(ma, mb, B, C, H, H, H, H, 1, out, ws, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        hipEventElapsedTime(&t[i], e0, e1);
    }
    std::sort(t.begin(), t.end());
    float s = 0.f; hipMemcpy(&s, out + 5, 4, hipMemcpyDeviceToHost);
    printf("B=%d C=%d %dx%d: stats + stream launches, us by events: p10 %.1f p50 %.1f p90 %.1f  (out[5] = %g)\n", B, C, H, H, 1e3 * t[iters / 10], 1e3 * t[iters / 2], 1e3 * t[iters * 9 / 10], s);
    return 0;
}
