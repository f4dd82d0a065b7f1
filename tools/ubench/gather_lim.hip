// Round-4 micro-benchmark: what bounds the ring loop's B-side gather of corr_fused_kernel - the CU's own port or a chip-level
// resource (L2 / fabric)?  The loop's gather is reproduced (8 gather waves, lane = (point row, 16-byte slot of a 128-byte line), two
// items x four taps per lane and stage, two stages in flight, one workgroup barrier per stage, 12 feature stages per pass over a
// 28 x 28 x 384 fp32 channels-last map with the ViT's token stride) and run
//   * on a varying number of workgroups per XCD (per-CU rate flat -> CU-bound; rising as CUs drop out -> chip-bound),
//   * with the real pixel stride (1536 B) and a padded one (1664 B: an odd number of 128-byte lines),
//   * with the stage order rotated per image,
//   * with the points sorted by pixel (duplicate lines adjacent: L1 hits),
//   * as a linear stream of the same byte count (the ceiling of this pipeline shape),
//   * with / without a 16 KB-per-stage LDS-DMA "anchor" stream by four more waves.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/gather_lim.hip -o tools/ubench/bin/gather_lim
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int HW = 28, NPIX = HW * HW, C = 384, NST = C / 32;

struct GP {
    const float* pool;
    const int* taps;             // [256][128][4] pixel indices
    const unsigned char* astream;// [256][NST][16 KB]
    float* sink;
    unsigned long long* stamps;  // [256][4]
    long long img_stride;        // floats
    int pix_stride;              // floats
    int passes;
    int active_per_xcd;          // workgroups per XCD that take part (slots 0 .. n-1)
    int rotate;                  // stage order rotated by the image index
    int linear;                  // 1: the same number of bytes as a linear stream
    int with_a;                  // 1: four more waves stream 16 KB per stage with LDS-DMA
    int barrier;                 // 1: one __syncthreads per stage (as the ring)
    int pipe3;                   // 1 (round 5): three register sets, stage s + 2 issued BEFORE stage s is consumed
    int heavy;                   // n: n extra dependent FMAs per consumed value pair (stands for the blend / split / LDS work of a commit)
};

__device__ __forceinline__ void dma_piece(const unsigned char* gsrc_lane, unsigned lds_addr)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc_lane), "s"(lds_addr) : "memory");
}

template <bool BARRIER>
__global__ void __launch_bounds__(768) gather_lim_kernel(GP p)
{
    __shared__ __attribute__((aligned(16))) unsigned char abuf[4][16384];
    __shared__ float dump[768];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int me = blockIdx.x, xcd = me & 7, slot = me >> 3;
    if (slot >= p.active_per_xcd) return;
    // images of XCD x: 4 "feats" images shared by 5 workgroups each (slots 0..19), 4 "pos" images with one reader (slots 20..23)
    const int img = slot < 20 ? xcd + 8 * (slot & 3) : 32 + xcd + 8 * (slot & 3);
    const float* base = p.pool + (long long)img * p.img_stride;
    const int rot = p.rotate ? (img >> 3) * 3 + (img & 7) : 0;
    unsigned long long t_start = 0, t_mid = 0;
    float acc = 0.f;
    if (wave < 8) {
        const int gt = tid, g8 = gt & 7, prow = gt >> 3;
        unsigned fo[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = 64 * j + prow;
            const int* t = p.taps + ((size_t)me * 128 + q) * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) fo[j][k] = p.linear ? (unsigned)(((j * 4 + k) * 512 + gt) * 16) : (unsigned)(t[k] * p.pix_stride + 4 * g8) * 4u;
        }
        f32x4 ga[2][4], gb[2][4];
        const unsigned stage_stride = p.linear ? 65536u : 128u;
        const char* bb = reinterpret_cast<const char*>(base);
        auto issue = [&](f32x4 (&g)[2][4], int s) {
            int f = s % NST + rot;
            f = f >= NST ? f - NST : f;
            const char* cb = bb + (size_t)((unsigned)f * stage_stride);      // wave-uniform
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) g[j][k] = *reinterpret_cast<const f32x4*>(cb + fo[j][k]);
        };
        auto consume = [&](const f32x4 (&g)[2][4]) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = 0.25f * g[j][0][e] + 0.5f * g[j][1][e] + 0.125f * g[j][2][e] + 0.0625f * g[j][3][e];
                    for (int h = 0; h < p.heavy; ++h) v = __builtin_fmaf(v, 1.0001f, 0.5f);
                    acc += v;
                }
        };
        const int total = p.passes * NST;
        issue(ga, 0);
        issue(gb, 1);
        __syncthreads();
        t_start = __builtin_amdgcn_s_memrealtime();
        if (p.pipe3) {
            // three sets: after the barrier of stage s the loads of stage s + 2 go out FIRST (into the set stage s - 1 left), then stage s is consumed
            f32x4 gc[2][4];
            for (int s = 0; s < total; s += 3) {
                if (BARRIER) __syncthreads();
                issue(gc, s + 2);
                __builtin_amdgcn_sched_barrier(0);
                consume(ga);
                __builtin_amdgcn_sched_barrier(0);
                if (BARRIER) __syncthreads();
                issue(ga, s + 3);
                __builtin_amdgcn_sched_barrier(0);
                consume(gb);
                __builtin_amdgcn_sched_barrier(0);
                if (BARRIER) __syncthreads();
                issue(gb, s + 4);
                __builtin_amdgcn_sched_barrier(0);
                consume(gc);
                __builtin_amdgcn_sched_barrier(0);
                if (s + 3 == NST) t_mid = __builtin_amdgcn_s_memrealtime();
            }
        } else
        for (int s = 0; s < total; s += 2) {          // (always issues: the last two land on stages 0 / 1 again and are drained below)
            if (BARRIER) __syncthreads();
            consume(ga);
            __builtin_amdgcn_sched_barrier(0);
            issue(ga, s + 2);
            if (BARRIER) __syncthreads();
            consume(gb);
            __builtin_amdgcn_sched_barrier(0);
            issue(gb, s + 3);
            if (s + 2 == NST) t_mid = __builtin_amdgcn_s_memrealtime();
        }
        consume(ga);
        consume(gb);
    } else {
        // the "anchor" stream: 16 KB per stage, LDS-DMA, three stages ahead
        const int w4 = wave - 8;
        const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)&abuf[0][0]);
        const unsigned char* src = p.astream + (size_t)me * NST * 16384;
        const int total = p.passes * NST;
        auto fill = [&](int s) {
            if (!p.with_a) return;
            const unsigned char* ss = src + (size_t)(s % NST) * 16384;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int pc = w4 + 4 * i;
                dma_piece(ss + pc * 1024 + lane * 16, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (s & 3) * 16384 + pc * 1024)));
            }
        };
        fill(0); fill(1); fill(2);
        __syncthreads();
        for (int s = 0; s < total; ++s) {
            if (p.with_a) { if (s + 2 < total) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            if (BARRIER) __syncthreads();
            if (s + 3 < total) fill(s + 3);
            acc += reinterpret_cast<const float*>(&abuf[s & 3][0])[tid & 255];
        }
    }
    const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
    dump[tid] = acc;
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int i = 0; i < 768; ++i) s += dump[i];
        p.sink[me] = s;
        p.stamps[me * 4 + 0] = t_start;
        p.stamps[me * 4 + 1] = t_mid;
        p.stamps[me * 4 + 2] = t_end;
    }
}

int main(int argc, char** argv)
{
    hipStream_t s;
    CK(hipStreamCreate(&s));
    const int NIMG = 64;
    const long long max_img = 785LL * 416;                       // padded pixel stride 416 floats = 1664 B
    float* pool;
    CK(hipMalloc(&pool, (size_t)NIMG * max_img * 4 + (1 << 20)));
    CK(hipMemset(pool, 0, (size_t)NIMG * max_img * 4 + (1 << 20)));
    unsigned char* astream;
    CK(hipMalloc(&astream, (size_t)256 * NST * 16384));
    CK(hipMemset(astream, 0, (size_t)256 * NST * 16384));
    std::vector<int> taps((size_t)256 * 128 * 4), taps_sorted(taps.size());
    srand(7);
    for (int w = 0; w < 256; ++w) {
        std::vector<std::pair<int, int>> pts;
        for (int q = 0; q < 128; ++q) {
            const float fx = (float)rand() / RAND_MAX * (HW - 1), fy = (float)rand() / RAND_MAX * (HW - 1);
            const int x0 = std::min((int)fx, HW - 2), y0 = std::min((int)fy, HW - 2);
            pts.push_back({y0, x0});
        }
        auto put = [&](std::vector<int>& dst, const std::vector<std::pair<int, int>>& v) {
            for (int q = 0; q < 128; ++q) {
                int* t = &dst[((size_t)w * 128 + q) * 4];
                const int y0 = v[q].first, x0 = v[q].second;
                t[0] = 1 + y0 * HW + x0; t[1] = 1 + y0 * HW + x0 + 1; t[2] = 1 + (y0 + 1) * HW + x0; t[3] = 1 + (y0 + 1) * HW + x0 + 1;   // (+1: the CLS token)
            }
        };
        put(taps, pts);
        std::sort(pts.begin(), pts.end());
        // sorted by pixel, then dealt to the lanes so that the 8 points of one wave instruction (prow = 8 w + i, items j) are neighbours:
        // point index q = 64 j + prow; instruction (wave w, item j) covers q = 64 j + 8 w .. + 7
        put(taps_sorted, pts);
    }
    int *d_taps, *d_taps_sorted;
    CK(hipMalloc(&d_taps, taps.size() * 4));
    CK(hipMalloc(&d_taps_sorted, taps.size() * 4));
    CK(hipMemcpy(d_taps, taps.data(), taps.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_taps_sorted, taps_sorted.data(), taps.size() * 4, hipMemcpyHostToDevice));
    float* sink;
    unsigned long long* stamps;
    CK(hipMalloc(&sink, 256 * 4));
    CK(hipMalloc(&stamps, 256 * 4 * 8));
    std::vector<unsigned long long> hs(256 * 4);
    // distinct lines per stage (dedup potential)
    {
        double distinct = 0;
        for (int w = 0; w < 256; ++w) {
            std::vector<int> v(taps.begin() + (size_t)w * 512, taps.begin() + (size_t)(w + 1) * 512);
            std::sort(v.begin(), v.end());
            distinct += std::unique(v.begin(), v.end()) - v.begin();
        }
        printf("{\"distinct_lines_per_stage_of_512\": %.1f}\n", distinct / 256);
    }
    auto run = [&](const char* name, GP p) {
        p.pool = pool; p.astream = astream; p.sink = sink; p.stamps = stamps;
        double best_mid = 1e9, best_all = 1e9, best_max = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            if (p.barrier) hipLaunchKernelGGL(gather_lim_kernel<true>, dim3(256), dim3(768), 0, s, p);
            else hipLaunchKernelGGL(gather_lim_kernel<false>, dim3(256), dim3(768), 0, s, p);
            CK(hipStreamSynchronize(s));
            CK(hipMemcpy(hs.data(), stamps, 256 * 4 * 8, hipMemcpyDeviceToHost));
            std::vector<double> steady, all;
            for (int w = 0; w < 256; ++w) {
                if ((w >> 3) >= p.active_per_xcd) continue;
                steady.push_back((double)(hs[w * 4 + 2] - hs[w * 4 + 1]) * 0.01 / ((p.passes - 1) * NST));
                all.push_back((double)(hs[w * 4 + 2] - hs[w * 4 + 0]) * 0.01 / (p.passes * NST));
            }
            std::sort(steady.begin(), steady.end());
            std::sort(all.begin(), all.end());
            if (rep == 0) continue;
            best_mid = std::min(best_mid, steady[steady.size() / 2]);
            best_max = std::min(best_max, steady.back());
            best_all = std::min(best_all, all[all.size() / 2]);
        }
        const double kb = 64.0 + (p.with_a ? 16.0 : 0.0);
        printf("{\"test\": \"%s\", \"wg_per_xcd\": %d, \"pix_stride_B\": %d, \"rotate\": %d, \"linear\": %d, \"with_a\": %d, \"barrier\": %d, "
               "\"us_per_stage_steady_p50\": %.3f, \"p100\": %.3f, \"incl_first_pass_p50\": %.3f, \"GBps_per_cu\": %.1f, \"TBps_chip\": %.2f}\n",
               name, p.active_per_xcd, p.pix_stride * 4, p.rotate, p.linear, p.with_a, p.barrier, best_mid, best_max, best_all,
               kb * 1.024 / best_mid, kb * 1.024 / best_mid * 8 * p.active_per_xcd * 1e-3);
        fflush(stdout);
    };
    GP p{};
    p.passes = 5;
    p.barrier = 1;
    const int counts[] = {24, 16, 12, 8, 4, 2, 1};
    for (int n : counts) {
        p.taps = d_taps; p.pix_stride = 384; p.img_stride = 785LL * 384; p.active_per_xcd = n; p.rotate = 0; p.linear = 0; p.with_a = 0;
        run("gather", p);
    }
    for (int n : {24, 8}) {
        p.taps = d_taps; p.pix_stride = 416; p.img_stride = 785LL * 416; p.active_per_xcd = n; p.rotate = 0; p.linear = 0; p.with_a = 0;
        run("gather_padded_stride", p);
        p.pix_stride = 384; p.img_stride = 785LL * 384; p.rotate = 1;
        run("gather_rotated", p);
        p.rotate = 0; p.taps = d_taps_sorted;
        run("gather_sorted_points", p);
        p.taps = d_taps; p.linear = 1;
        run("linear_stream", p);
        p.linear = 0; p.with_a = 1;
        run("gather_plus_anchor_stream", p);
        p.with_a = 0; p.barrier = 0;
        run("gather_no_barrier", p);
        p.barrier = 1;
        // round 5: the commit's work between barrier and issue (heavy), and the issue moved in front of it (pipe3)
        p.heavy = 12; run("gather_heavy_commit", p);
        p.pipe3 = 1; run("gather_heavy_commit_pipe3", p);
        p.with_a = 1; run("gather_heavy_commit_pipe3_plus_anchor", p);
        p.pipe3 = 0; run("gather_heavy_commit_plus_anchor", p);
        p.with_a = 0; p.heavy = 0; p.pipe3 = 1; run("gather_pipe3", p);
        p.pipe3 = 0;
    }
    return 0;
}
