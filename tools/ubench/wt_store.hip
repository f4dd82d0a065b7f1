// Round-5 micro-benchmark: how fast do phase 1's write-through (sc1) stores of the anchor operand leave a compute unit, by store pattern?
// 64 "light" workgroups (8 per XCD, 12 waves) each write what corr_fused_kernel's phase 1 writes per light workgroup - 64 rows x
// (12 feature stages x 128 B + 3 code stages x 128 B) = 120 KB - while (optionally) the other 192 workgroups stream cold data like the gather heads:
//   mode 0  "runs":      from an LDS staging area, every wave whole planes of rows x 64 B contiguous (what the kernel does today, after a barrier)
//   mode 1  "lines":     straight from registers, 16 B per lane, every group of 8 lanes one FULL 128-byte line ([stage][row][hi 64 | lo 64]); lines 16 KB apart
//   mode 2  "halves":    straight from registers, 16 B per lane, every group of 4 lanes a 64-byte half line ([stage][plane][row 64 B]: today's layout)
// Stamps per writer workgroup: start, all stores issued, all stores acknowledged (s_waitcnt vmcnt(0)); reported p50 / p100 over the 64 writers.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/wt_store.hip -o tools/ubench/bin/wt_store
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NST = 12, NKC = 3, ROWS = 64;

struct SP {
    unsigned char* fs;            // [32 anchors][NST][16 KB]
    unsigned char* csf;           // [32][NKC][16 KB]
    const float* pool;            // background stream
    float* sink;
    unsigned long long* stamps;   // [256][4]
    int mode;
    int background;               // 1: the 192 other workgroups stream 384 KB each
    int bytes_fs, bytes_csf;
};

__global__ void __launch_bounds__(768) wt_store_kernel(SP p)
{
    __shared__ __attribute__((aligned(16))) unsigned char stage[ROWS * (NST + NKC) * 128];      // 120 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int me = blockIdx.x, xcd = me & 7, slot = me >> 3;
    const __amdgpu_buffer_rsrc_t fs_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.fs, 0, p.bytes_fs, 0x00020000);
    const __amdgpu_buffer_rsrc_t csf_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.csf, 0, p.bytes_csf, 0x00020000);
    if (slot >= 8) {
        if (!p.background) return;
        // a cold stream: 8 x 16 B per lane per round, 8 rounds = 384 KB per workgroup
        const f32x4* src = reinterpret_cast<const f32x4*>(p.pool) + (size_t)me * 32768;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < 4; ++r) {
            f32x4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = src[(r * 8 + i) * 768 + tid];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += v[i];
        }
        if (acc[0] == 123.456f) p.sink[me] = acc[1] + acc[2] + acc[3];
        return;
    }
    // writer: rows [64 * (slot & 1), + 64) of anchor xcd + 8 * (slot >> 1)
    const int anchor = xcd + 8 * (slot >> 1), q0 = 64 * (slot & 1);
    const u32x4 d = {(unsigned)tid, (unsigned)me, 0x3c003c00u, 0x3c003c00u};
    if (p.mode == 0) {
        for (int i = tid; i < (int)sizeof(stage) / 16; i += 768) reinterpret_cast<u32x4*>(stage)[i] = d;
    }
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), t1;
    if (p.mode == 0) {
        // planes: 2 * NST feature planes of ROWS x 64 B, NKC code stages of ROWS x 128 B; plane pl by wave pl % 12
        for (int pl = wave; pl < 2 * NST + NKC; pl += 12) {
            if (pl < 2 * NST) {
                const int s2 = pl >> 1, pp = pl & 1;
                const unsigned base = (unsigned)(((size_t)anchor * NST + s2) * 16384 + pp * 8192 + q0 * 64);
                const unsigned char* src = stage + pl * ROWS * 64;
                for (int u = lane; u < ROWS * 4; u += 64)
                    __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(src + u * 16), fs_rsrc, base + u * 16, 0, 16);
            } else {
                const unsigned base = (unsigned)(((size_t)anchor * NKC + (pl - 2 * NST)) * 16384 + q0 * 128);
                const unsigned char* src = stage + 2 * NST * ROWS * 64 + (pl - 2 * NST) * ROWS * 128;
                for (int u = lane; u < ROWS * 8; u += 64)
                    __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(src + u * 16), csf_rsrc, base + u * 16, 0, 16);
            }
        }
    } else {
        // rows: chunk 1 = 4 rows per wave (two per half-wave), chunk 2 = 2 rows per wave of waves 4..11
        const int hl = lane & 31, hw = lane >> 5;
        for (int chunk = 0; chunk < 2; ++chunk) {
            if (chunk == 1 && wave < 4) break;
            const int G = chunk == 0 ? 2 : 1;
            for (int g = 0; g < G; ++g) {
                const int row = chunk == 0 ? 4 * wave + 2 * g + hw : 48 + 2 * (wave - 4) + hw;
                const int q = q0 + row;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int s2 = 4 * j + (hl >> 3);
                    unsigned off;
                    if (p.mode == 1) off = (unsigned)((((size_t)anchor * NST + s2) * 128 + q) * 128 + (((hl & 7)) ^ ((q >> 1) & 7)) * 16);
                    else off = (unsigned)(((size_t)anchor * NST + s2) * 16384 + (hl & 1) * 8192 + q * 64 + ((((hl & 7) >> 1)) ^ ((q >> 2) & 3)) * 16);
                    __builtin_amdgcn_raw_buffer_store_b128(d, fs_rsrc, off, 0, 16);
                }
                // codes: 24 lanes x 16 B = 96 B of the row's 128-byte line in each... one chunk per 8 lanes: lanes 0..23 -> chunk hl / 8 (full 128 B when 8 lanes)
                if (hl < 24) {
                    const unsigned off = (unsigned)((((size_t)anchor * NKC + (hl >> 3)) * 128 + q) * 128 + (((hl & 7)) ^ ((q >> 1) & 7)) * 16);
                    __builtin_amdgcn_raw_buffer_store_b128(d, csf_rsrc, off, 0, 16);
                }
            }
        }
    }
    t1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
    __shared__ unsigned long long tmax[2];
    if (tid == 0) { tmax[0] = 0; tmax[1] = 0; }
    __syncthreads();
    atomicMax(&tmax[0], t1);
    atomicMax(&tmax[1], t2);
    __syncthreads();
    if (tid == 0) {
        p.stamps[me * 4 + 0] = t0;
        p.stamps[me * 4 + 1] = tmax[0];
        p.stamps[me * 4 + 2] = tmax[1];
    }
}

int main()
{
    hipStream_t s;
    CK(hipStreamCreate(&s));
    SP p{};
    p.bytes_fs = 32 * NST * 16384;
    p.bytes_csf = 32 * NKC * 16384;
    CK(hipMalloc(&p.fs, p.bytes_fs));
    CK(hipMalloc(&p.csf, p.bytes_csf));
    float* pool;
    const size_t pool_bytes = (size_t)256 * 32768 * 16 * 4;     // 4 rotating copies so that the background stream stays cold
    CK(hipMalloc(&pool, pool_bytes));
    CK(hipMemset(pool, 0, pool_bytes));
    CK(hipMalloc(&p.sink, 256 * 4));
    CK(hipMalloc(&p.stamps, 256 * 4 * 8));
    std::vector<unsigned long long> hs(256 * 4);
    const char* names[3] = {"runs_from_lds", "full_lines_from_registers", "half_lines_from_registers"};
    for (int bg = 0; bg < 2; ++bg)
        for (int mode = 0; mode < 3; ++mode) {
            p.mode = mode;
            p.background = bg;
            double b_issue50 = 1e9, b_issue100 = 1e9, b_ack50 = 1e9, b_ack100 = 1e9, b_span = 1e9;
            for (int rep = 0; rep < 6; ++rep) {
                p.pool = pool + (size_t)(rep & 3) * 256 * 32768 * 4;
                hipLaunchKernelGGL(wt_store_kernel, dim3(256), dim3(768), 0, s, p);
                CK(hipStreamSynchronize(s));
                CK(hipMemcpy(hs.data(), p.stamps, 256 * 4 * 8, hipMemcpyDeviceToHost));
                std::vector<double> issue, ack;
                unsigned long long first = ~0ull, last = 0;
                for (int w = 0; w < 64; ++w) {
                    issue.push_back((double)(hs[w * 4 + 1] - hs[w * 4 + 0]) * 0.01);
                    ack.push_back((double)(hs[w * 4 + 2] - hs[w * 4 + 0]) * 0.01);
                    first = std::min(first, hs[w * 4 + 0]);
                    last = std::max(last, hs[w * 4 + 2]);
                }
                std::sort(issue.begin(), issue.end());
                std::sort(ack.begin(), ack.end());
                if (rep == 0) continue;
                b_issue50 = std::min(b_issue50, issue[32]); b_issue100 = std::min(b_issue100, issue[63]);
                b_ack50 = std::min(b_ack50, ack[32]); b_ack100 = std::min(b_ack100, ack[63]);
                b_span = std::min(b_span, (double)(last - first) * 0.01);
            }
            printf("{\"test\": \"%s\", \"background_stream\": %d, \"KB_per_workgroup\": %.1f, \"issued_us_p50\": %.2f, \"issued_us_p100\": %.2f, "
                   "\"acked_us_p50\": %.2f, \"acked_us_p100\": %.2f, \"first_start_to_last_ack_us\": %.2f, \"TBps_chip\": %.2f}\n",
                   names[mode], bg, ROWS * (NST + NKC) * 128 / 1024.0, b_issue50, b_issue100, b_ack50, b_ack100, b_span,
                   64.0 * ROWS * (NST + NKC) * 128 / b_span * 1e-6);
            fflush(stdout);
        }
    return 0;
}
