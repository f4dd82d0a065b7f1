// Round-6 micro-benchmark ("next" item 2): the TRAFFIC SKELETON of the fused forward - what the memory system alone
// charges for corr_fused_kernel's access pattern, and what it would charge for the same bytes cut into column-half sub-tiles with two
// workgroups per compute unit.
//
// Same grid, XCD placement, gather addresses (4 bilinear taps per point, 128-byte pieces of channels-last pixels, codes as 8-byte pieces),
// LDS-DMA copies of the anchor operand (16 KB per 32-channel stage, sc1), ring stores, output / context stores as the real kernel;
// NO normalisation, NO rendezvous (old_mean, tail), no tile assignment in the kernel (the host builds the table).  Optional parts,
// switched per run:  p1 (the anchors are sampled in the launch and tiles wait for their counter; off = the operand of the previous launch
// is streamed, nobody waits), mfma (the ring's fragment reads + the 3 x fp16 MFMAs of every stage, operands are whatever the ring holds),
// gather / astream / out (each stream on its own).
//
//   layout FULL: one workgroup of 768 threads (4 MFMA + 8 gather waves, 136 KB of LDS) per 128 x 128 tile, 256 workgroups
//                (tiles + helpers), phase 1 on the workgroups without a gather stream - the structure of corr_fused_kernel;
//   layout HALF: one workgroup of 512 threads (4 + 4 waves, 76 KB of LDS) per 128 x 64 column half, two per compute unit; the anchors
//                by 2 B workgroups of their own at the front of the grid.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/fused_skeleton.hip -o tools/ubench/bin/fused_skeleton
//   tools/ubench/bin/fused_skeleton [B=32] [C=384] [HW=28] [launches=40]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int TP = 128, K = 70, KPER = 24, NKC = 3, LDK = 76, NEG = 5, NSETS = 7, S = 11, P = S * S, P2 = P * P;
constexpr int RS_SIDE = 16384;

enum { F_P1 = 1, F_GATHER = 2, F_ASTREAM = 4, F_MFMA = 8, F_OUT = 16, F_ALL = 31 };

struct SK {
    const float* feats; const float* feats_pos; const float* code; const float* code_pos;     // channels-last, this launch's input set
    long long f_sn, c_sn;        // image strides (floats)
    int f_sp, c_sp;              // pixel strides (floats)
    const int4* taps;            // [NSETS*B][128] pixel indices of the 4 taps (set s = role * B + b; role 0 anchors @coords1)
    const float4* tapw;          // [NSETS*B][128]
    const int* tile_src;         // [NSETS*B] source image of the tile's B side
    const int* tile_rank;        // [NSETS*B] rank of the tile among the tiles that gather from the same image
    const int* wg_item;          // [grid] work item of workgroup w: FULL: tile or -1; HALF: 2 * tile + half, or -1
    unsigned char* fs;           // anchor feature operand [B][NCH2][16 KB]
    unsigned char* csf;          // anchor code operand    [B][NKC][16 KB]
    float* cs;                   // context rows [NSETS*B][128][LDK]
    int4* o_taps; float4* o_tapw;
    unsigned* anchor_cnt;        // [B] * 64 words apart
    float* cd; float* w; float* loss;     // [NSETS*B][P2] each (loss: only sets >= 2B are written)
    unsigned long long* stamps;  // [grid][4]
    int B, NCH2, flags, n_anchor_wg;
    const float* cold;           // experiment pool: one feature image per workgroup that nobody else reads
    int src_mode;                // 0 the real sources; 1 every tile gathers from ONE image per XCD (L2-resident); 2 inter tiles cold-unique, others hot;
                                 // 3 every tile: points 0..63 hot, 64..127 cold-unique (hits and misses mixed in every wave); 4 everything cold-unique
    int rot_mul;                 // feature stages start at (rank * rot_mul) % NCH2: tiles that share an image do not ask for the same lines at the same time
};

__device__ __forceinline__ void dma_piece_sc1(const unsigned char* gsrc_lane, unsigned lds_addr)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc_lane), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ unsigned lds_address(const void* p)
{
    return (unsigned)(uintptr_t)((__attribute__((address_space(3))) const unsigned char*)p);
}
__device__ __forceinline__ int swz_h(int r, int u) { return r * 64 + ((u ^ ((r >> 2) & 3)) << 4); }
__device__ __forceinline__ void split_f16_pair(float x, float y, unsigned& hi, unsigned& lo)
{
    const f16x2 h = __builtin_convertvector(f32x2{x, y}, f16x2);
    hi = __builtin_bit_cast(unsigned, h);
    const f32x2 hf = __builtin_convertvector(h, f32x2);
    const f16x2 l = __builtin_convertvector(f32x2{x - hf[0], y - hf[1]}, f16x2);
    lo = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ void ring_barrier()
{
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// ---- anchor sampler: rows [beg, end) of the list "anchors x, x + 8, ..." (128 rows per anchor), 4 rows per wave and pass
// (one point per half-wave and G = 2 points in flight), straight 16-byte sc1 stores of the operand pieces (64-byte runs)
template <int NJ, int G>
__device__ void sample_anchor_rows(const SK& p, int x, int beg, int end, int wave, int nwaves, int lane)
{
    const int hl = lane & 31, hw = lane >> 5;
    const int NCH2 = p.NCH2;
    const __amdgpu_buffer_rsrc_t fs_r = __builtin_amdgcn_make_buffer_rsrc(p.fs, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t csf_r = __builtin_amdgcn_make_buffer_rsrc(p.csf, 0, 0x7fffffff, 0x00020000);
    for (int r0 = beg + 2 * G * wave; r0 < end; r0 += 2 * G * nwaves) {
        f32x4 v[G][NJ];
        f32x2 ca[G][4];
        float cb[G][4];
        int b[G], q[G];
        float4 w[G];
        bool act[G];
        int tt[G][4];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int idx = r0 + 2 * g + hw;
            act[g] = idx < end;
            b[g] = min(x + 8 * (idx >> 7), p.B - 1);
            q[g] = idx & 127;
            const int4 tp = p.taps[(size_t)b[g] * TP + q[g]];
            w[g] = p.tapw[(size_t)b[g] * TP + q[g]];
            tt[g][0] = tp.x; tt[g][1] = tp.y; tt[g][2] = tp.z; tt[g][3] = tp.w;
            const float* cbp = p.code + (long long)b[g] * p.c_sn;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                ca[g][k] = *reinterpret_cast<const f32x2*>(cbp + (size_t)tt[g][k] * p.c_sp + 2 * hl);
                cb[g][k] = cbp[(size_t)tt[g][k] * p.c_sp + (64 + hl < K ? 64 + hl : 0)];
            }
        }
        // the feature taps in chunks of three 128-channel groups (48 registers per point in flight)
#pragma unroll
        for (int j0 = 0; j0 < NJ; j0 += 3) {
            f32x4 t[G][3][4];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float* fb = p.feats + (long long)b[g] * p.f_sn + 4 * hl;
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int j = 0; j < 3; ++j) t[g][j][k] = *reinterpret_cast<const f32x4*>(fb + (size_t)tt[g][k] * p.f_sp + 128 * (j0 + j));
            }
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        v[g][j0 + j][e] = w[g].x * t[g][j][0][e] + w[g].y * t[g][j][1][e] + w[g].z * t[g][j][2][e] + w[g].w * t[g][j][3][e];
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (!act[g]) continue;
            const float4 wg = w[g];
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) ss += v[g][j][e] * v[g][j][e];
#pragma unroll
            for (int m = 16; m >= 1; m >>= 1) ss += __shfl_xor(ss, m, 64);
            const float inv = 1.f / fmaxf(sqrtf(ss), 1e-10f);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                unsigned h0, l0, h1, l1;
                split_f16_pair(v[g][j][0] * inv, v[g][j][1] * inv, h0, l0);
                split_f16_pair(v[g][j][2] * inv, v[g][j][3] * inv, h1, l1);
                const bool odd = hl & 1;
                const unsigned r0_ = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(odd ? h0 : l0), 0xB1, 0xF, 0xF, true);
                const unsigned r1_ = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(odd ? h1 : l1), 0xB1, 0xF, 0xF, true);
                const u32x4 d = odd ? u32x4{r0_, r1_, l0, l1} : u32x4{h0, h1, r0_, r1_};
                const int c = 128 * j + 4 * (hl & ~1);
                const int s2 = c >> 5, u = ((c & 31) >> 3) ^ ((q[g] >> 2) & 3);
                if (s2 < NCH2) {
                    const unsigned off = (unsigned)(((size_t)b[g] * NCH2 + s2) * RS_SIDE + (odd ? 8192 : 0) + q[g] * 64 + u * 16);
                    __builtin_amdgcn_raw_buffer_store_b128(d, fs_r, off, 0, 16);
                }
            }
            // codes: operand stages (format F: 128-byte rows of KPER floats) + the context row
            f32x2 c0 = f32x2{wg.x * ca[g][0][0] + wg.y * ca[g][1][0] + wg.z * ca[g][2][0] + wg.w * ca[g][3][0],
                             wg.x * ca[g][0][1] + wg.y * ca[g][1][1] + wg.z * ca[g][2][1] + wg.w * ca[g][3][1]};
            float c1 = wg.x * cb[g][0] + wg.y * cb[g][1] + wg.z * cb[g][2] + wg.w * cb[g][3];
            float cs2 = c0[0] * c0[0] + c0[1] * c0[1] + c1 * c1;
#pragma unroll
            for (int m = 16; m >= 1; m >>= 1) cs2 += __shfl_xor(cs2, m, 64);
            const float cinv = 1.f / fmaxf(sqrtf(cs2), 1e-10f);
            c0 = c0 * cinv;
            c1 *= cinv;
            if (2 * hl < NKC * KPER) {
                const int k = 2 * hl, sc = k / KPER, col = k - sc * KPER;
                const unsigned off = (unsigned)(((size_t)b[g] * NKC + sc) * RS_SIDE + q[g] * 128 + col * 4);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, c0[0]), csf_r, off, 0, 16);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, c0[1]), csf_r, off + 4, 0, 16);
            }
            if (64 + hl < NKC * KPER) {
                const int k = 64 + hl, sc = k / KPER, col = k - sc * KPER;
                const unsigned off = (unsigned)(((size_t)b[g] * NKC + sc) * RS_SIDE + q[g] * 128 + col * 4);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, c1), csf_r, off, 0, 16);
            }
            float* cx = p.cs + ((size_t)b[g] * TP + q[g]) * LDK;
            if (2 * hl < 72) *reinterpret_cast<f32x2*>(cx + 2 * hl) = c0;
            if (64 + hl < 72) cx[64 + hl] = c1;
            if (hl == 0) { p.o_taps[(size_t)b[g] * TP + q[g]] = p.taps[(size_t)b[g] * TP + q[g]]; p.o_tapw[(size_t)b[g] * TP + q[g]] = wg; }
        }
    }
}

__device__ __forceinline__ void publish_rows(const SK& p, int x, int first, int n)
{
    while (n > 0) {
        const int a = first >> 7;
        const int m = min(n, ((a + 1) << 7) - first);
        __hip_atomic_fetch_add(p.anchor_cnt + (size_t)(x + 8 * a) * 64, (unsigned)m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        first += m;
        n -= m;
    }
}

// ---- the kernel.  HALF = 0: 128 x 128 tiles, 768 threads; HALF = 1: 128 x 64 column halves, 512 threads
template <int HALF, int NJ>
__global__ void __launch_bounds__(HALF ? 512 : 768, HALF ? 4 : 3) skeleton_kernel(const SK p)
{
    constexpr int GW = HALF ? 4 : 8;                 // gather waves
    constexpr int NB = HALF ? 64 : 128;              // B points of the work item
    constexpr int NS = HALF ? 3 : 4;                 // ring stages
    constexpr int BSIDE = NB * 128;                  // bytes of the B side of a stage (hi plane + lo plane)
    constexpr int STAGE = RS_SIDE + BSIDE;
    constexpr int NCH2 = 4 * NJ;
    constexpr int NT = NKC + NCH2;
    constexpr int GPW = 8 * GW;                      // points per item across the team
    constexpr int GI = NB / GPW;                     // 2
    constexpr int NTHR = 64 * (4 + GW);
    constexpr int LDT = NB + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ring = smem + 1024;
    float* Tfd = reinterpret_cast<float*>(ring);
    float* Tcd = Tfd + TP * LDT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool mfma_team = wave8 < 4;
    const int wave = wave8 & 3;
    const int me = blockIdx.x;
    const int B = p.B, n_tiles = NSETS * B;
    unsigned long long* st = p.stamps + (size_t)me * 4;
    if (tid == 0) st[0] = __builtin_amdgcn_s_memrealtime();
    const int item = p.wg_item[me];

    // ---------------- phase 1
    if (p.flags & F_P1) {
        if (!HALF) {
            // the light workgroups (self-correlation tiles and helpers: 8 per XCD at B = 32) sample 64 rows each with all twelve waves
            const bool light = item < 0 || item < B;
            const int x = me & 7, r = me >> 3;
            const int nb = (B - x + 7) >> 3;                       // anchors of this XCD
            const int tile_slots = (n_tiles - x + 7) >> 3, nslot = (gridDim.x - x + 7) >> 3;
            const int n_light = nb + (nslot - tile_slots);
            const int lr = r < nb ? r : nb + (r - tile_slots);
            if (light && n_light > 0) {
                const int R = nb * TP;
                const int beg = R * lr / n_light, end = R * (lr + 1) / n_light;
                sample_anchor_rows<NJ, (NJ <= 3 ? 2 : 1)>(p, x, beg, end, wave8, 4 + GW, lane);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) publish_rows(p, x, beg, end - beg);
            }
        } else {
            // the samplers of XCD x: its anchor workgroups (front of the grid) and the workgroups that hold a self-correlation half
            // (no gather stream of their own; the placement puts them on the first 2 nb tile slots of the XCD)
            const int x = me & 7;
            const int nb = (B - x + 7) >> 3;
            const int na = p.n_anchor_wg / 8;
            const int r = me < p.n_anchor_wg ? me >> 3 : na + ((me - p.n_anchor_wg) >> 3);
            const int ns = na + 2 * nb;
            if (r < ns && nb > 0) {
                const int R = nb * TP;
                const int beg = R * r / ns, end = R * (r + 1) / ns;
                sample_anchor_rows<NJ, 1>(p, x, beg, end, wave8, 4 + GW, lane);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) publish_rows(p, x, beg, end - beg);
            }
        }
    }
    if (item < 0) { if (tid == 0) { st[1] = st[2] = st[3] = __builtin_amdgcn_s_memrealtime(); } return; }

    const int tile = HALF ? item >> 1 : item, half = HALF ? item & 1 : 0;
    const int b = tile % B, pset = tile / B;
    const bool sameAB = pset == 0;
    const int sB = tile;
    const int src = p.tile_src[tile];
    const float* imgB = (pset == 1 ? p.feats_pos : p.feats) + (long long)src * p.f_sn;
    const float* cimgB = (pset == 1 ? p.code_pos : p.code) + (long long)src * p.c_sn;
    const unsigned char* fsA = p.fs + (size_t)b * NCH2 * RS_SIDE;
    const unsigned char* csfA = p.csf + (size_t)b * NKC * RS_SIDE;
    const int q0 = half * 64;                                      // first B point of my half
    const int rot = __builtin_amdgcn_readfirstlane((p.tile_rank[tile] * p.rot_mul + (HALF ? half * (p.rot_mul ? NCH2 / 2 : 0) : 0)) % NCH2);
    auto fstage = [&](int f) { const int g = f + rot; return g >= NCH2 ? g - NCH2 : g; };     // feature stage visited at step f

    f32x16 acc[2][HALF ? 1 : 2], accc[2][HALF ? 1 : 2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < (HALF ? 1 : 2); ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.f; accc[i][j][e] = 0.f; }

    if (mfma_team) {
        auto stage_src = [&](int n) { return n < NKC ? csfA + (size_t)n * RS_SIDE : fsA + (size_t)fstage(n - NKC) * RS_SIDE; };
        const unsigned ring_addr = __builtin_amdgcn_readfirstlane(lds_address(ring));
        if (wave == 0 && (p.flags & F_P1)) {
            for (;;) {
                const unsigned c = __hip_atomic_load(p.anchor_cnt + (size_t)b * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__builtin_amdgcn_readfirstlane(c) >= (unsigned)TP) break;
                __builtin_amdgcn_s_sleep(10);
            }
        }
        if (tid == 0) st[1] = __builtin_amdgcn_s_memrealtime();
        const bool astream = p.flags & F_ASTREAM;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (wave == 0 && astream)
            for (int n = 0; n < NS - 1; ++n)
                for (int pc = 0; pc < 16; ++pc) dma_piece_sc1(stage_src(n) + pc * 1024 + lane * 16, ring_addr + n * STAGE + pc * 1024);
        const int wr = wave >> 1, wc = wave & 1;
        const int r = lane & 31, hf = lane >> 5;
        for (int n = 0; n < NT; ++n) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the skeleton drains; the product counts)
            ring_barrier();
            if (n + NS - 1 < NT && astream) {
                const unsigned char* s3 = stage_src(n + NS - 1);
                const unsigned dst = ring_addr + ((n + NS - 1) % NS) * STAGE;
#pragma unroll
                for (int i = 0; i < 4; ++i) dma_piece_sc1(s3 + (wave + 4 * i) * 1024 + lane * 16, dst + (wave + 4 * i) * 1024);
            }
            if (p.flags & F_MFMA) {
                const unsigned char* As = ring + (n % NS) * STAGE;
                const unsigned char* Bs = sameAB ? As + (HALF ? 0 : 0) : As + RS_SIDE;
                const int lob = sameAB ? 8192 : NB * 64;
                const int rb0 = sameAB ? q0 : 0;
                auto& A_ = n < NKC ? accc : acc;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int u = 2 * ks + hf;
                    if constexpr (!HALF) {
                        const int ra0 = 64 * wr + r, ra1 = ra0 + 32, rb_0 = 64 * wc + r, rb_1 = rb_0 + 32;
                        const f16x8 ah0 = *reinterpret_cast<const f16x8*>(As + swz_h(ra0, u)), al0 = *reinterpret_cast<const f16x8*>(As + 8192 + swz_h(ra0, u));
                        const f16x8 ah1 = *reinterpret_cast<const f16x8*>(As + swz_h(ra1, u)), al1 = *reinterpret_cast<const f16x8*>(As + 8192 + swz_h(ra1, u));
                        const f16x8 bh0 = *reinterpret_cast<const f16x8*>(Bs + swz_h(rb_0, u)), bl0 = *reinterpret_cast<const f16x8*>(Bs + lob + swz_h(rb_0, u));
                        const f16x8 bh1 = *reinterpret_cast<const f16x8*>(Bs + swz_h(rb_1, u)), bl1 = *reinterpret_cast<const f16x8*>(Bs + lob + swz_h(rb_1, u));
                        A_[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, bh0, A_[0][0], 0, 0, 0);
                        A_[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, bh1, A_[0][1], 0, 0, 0);
                        A_[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1, bh0, A_[1][0], 0, 0, 0);
                        A_[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1, bh1, A_[1][1], 0, 0, 0);
                        A_[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bl0, A_[0][0], 0, 0, 0);
                        A_[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bl1, A_[0][1], 0, 0, 0);
                        A_[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bl0, A_[1][0], 0, 0, 0);
                        A_[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bl1, A_[1][1], 0, 0, 0);
                        A_[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bh0, A_[0][0], 0, 0, 0);
                        A_[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bh1, A_[0][1], 0, 0, 0);
                        A_[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bh0, A_[1][0], 0, 0, 0);
                        A_[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bh1, A_[1][1], 0, 0, 0);
                    } else {
                        const int ra0 = 64 * wr + r, ra1 = ra0 + 32, rb_0 = rb0 + 32 * wc + r;
                        const f16x8 ah0 = *reinterpret_cast<const f16x8*>(As + swz_h(ra0, u)), al0 = *reinterpret_cast<const f16x8*>(As + 8192 + swz_h(ra0, u));
                        const f16x8 ah1 = *reinterpret_cast<const f16x8*>(As + swz_h(ra1, u)), al1 = *reinterpret_cast<const f16x8*>(As + 8192 + swz_h(ra1, u));
                        const f16x8 bh0 = *reinterpret_cast<const f16x8*>(Bs + swz_h(rb_0, u)), bl0 = *reinterpret_cast<const f16x8*>(Bs + lob + swz_h(rb_0, u));
                        A_[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, bh0, A_[0][0], 0, 0, 0);
                        A_[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1, bh0, A_[1][0], 0, 0, 0);
                        A_[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bl0, A_[0][0], 0, 0, 0);
                        A_[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bl0, A_[1][0], 0, 0, 0);
                        A_[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bh0, A_[0][0], 0, 0, 0);
                        A_[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bh0, A_[1][0], 0, 0, 0);
                    }
                }
            }
        }
    } else if (sameAB || !(p.flags & F_GATHER)) {
        for (int n = 0; n < NT; ++n) ring_barrier();
    } else {
        // ---------------- gather team
        const int gt = tid - 256, g8 = gt & 7, prow = gt >> 3;
        unsigned fo[GI][4], co[GI][4];
        float4 tw[GI];
#pragma unroll
        for (int j = 0; j < GI; ++j) {
            const int q = q0 + GPW * j + prow;
            const int4 tp = p.taps[(size_t)sB * TP + q];
            tw[j] = p.tapw[(size_t)sB * TP + q];
            fo[j][0] = (unsigned)(tp.x * p.f_sp + 4 * g8) * 4u; fo[j][1] = (unsigned)(tp.y * p.f_sp + 4 * g8) * 4u;
            fo[j][2] = (unsigned)(tp.z * p.f_sp + 4 * g8) * 4u; fo[j][3] = (unsigned)(tp.w * p.f_sp + 4 * g8) * 4u;
            co[j][0] = (unsigned)(tp.x * p.c_sp) * 4u; co[j][1] = (unsigned)(tp.y * p.c_sp) * 4u;
            co[j][2] = (unsigned)(tp.z * p.c_sp) * 4u; co[j][3] = (unsigned)(tp.w * p.c_sp) * 4u;
            if (g8 == 0) { p.o_taps[(size_t)sB * TP + q] = tp; p.o_tapw[(size_t)sB * TP + q] = tw[j]; }
        }
        const float* hot = p.feats + (long long)(me & 7) * p.f_sn;
        const float* uniq = p.cold + (long long)me * p.f_sn;
        const float* i0 = p.src_mode == 0 ? imgB : (p.src_mode == 1 || p.src_mode == 3) ? hot : p.src_mode == 2 ? (pset == 1 ? uniq : hot) : uniq;
        const float* i1 = p.src_mode == 0 ? imgB : p.src_mode == 1 ? hot : p.src_mode == 2 ? (pset == 1 ? uniq : hot) : uniq;
        const __amdgpu_buffer_rsrc_t imgB_r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(i0), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t imgB_r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(i1), 0, 0x7fffffff, 0x00020000);
        struct GSet { f32x4 tv[GI][4]; };
        auto issue = [&](GSet& g, int m) {
            if (m < NKC) {
                const int k = m * KPER + 4 * g8;
                const bool in = 4 * g8 < KPER;
                const bool v0 = in && k + 1 < K, v1 = in && k + 3 < K;
                const unsigned k0 = v0 ? 4u * k : 0u, k1 = v1 ? 4u * (k + 2) : 0u;
                const char* cb = reinterpret_cast<const char*>(cimgB);
#pragma unroll
                for (int j = 0; j < GI; ++j)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const f32x2 lo = *reinterpret_cast<const f32x2*>(cb + (co[j][t] + k0));
                        const f32x2 hi = *reinterpret_cast<const f32x2*>(cb + (co[j][t] + k1));
                        g.tv[j][t] = f32x4{v0 ? lo[0] : 0.f, v0 ? lo[1] : 0.f, v1 ? hi[0] : 0.f, v1 ? hi[1] : 0.f};
                    }
            } else {
                const int so = __builtin_amdgcn_readfirstlane(fstage(m - NKC) * 128);
#pragma unroll
                for (int j = 0; j < GI; ++j)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        g.tv[j][t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(j == 0 ? imgB_r0 : imgB_r1, (int)fo[j][t], so, 0));
            }
        };
        float ss[GI] = {0.f, 0.f};
        auto commit = [&](const GSet& g, int m) {
            unsigned char* dst = ring + (m % NS) * STAGE + RS_SIDE;
#pragma unroll
            for (int j = 0; j < GI; ++j) {
                const int q = GPW * j + prow;                      // row inside my B side
                const float4 w = tw[j];
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = w.x * g.tv[j][0][e] + w.y * g.tv[j][1][e] + w.z * g.tv[j][2][e] + w.w * g.tv[j][3][e];
                ss[j] += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
                unsigned h0, l0, h1, l1;
                split_f16_pair(v[0], v[1], h0, l0);
                split_f16_pair(v[2], v[3], h1, l1);
                const bool odd = g8 & 1;
                const unsigned r0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(odd ? h0 : l0), 0xB1, 0xF, 0xF, true);
                const unsigned r1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(odd ? h1 : l1), 0xB1, 0xF, 0xF, true);
                const u32x4 d = odd ? u32x4{r0, r1, l0, l1} : u32x4{h0, h1, r0, r1};
                *reinterpret_cast<u32x4*>(dst + (odd ? NB * 64 : 0) + swz_h(q, g8 >> 1)) = d;
            }
            if (m == NKC - 1) {
                // context rows of the B set (normalised codes), from the registers
#pragma unroll
                for (int j = 0; j < GI; ++j) {
                    const int q = q0 + GPW * j + prow;
                    float* crow = p.cs + ((size_t)sB * TP + q) * LDK;
                    for (int mm = 0; mm < NKC; ++mm)
                        if (4 * g8 < KPER) *reinterpret_cast<f32x4*>(crow + mm * KPER + 4 * g8) = g.tv[j][mm];
                }
            }
        };
        GSet ga, gb;
        // head: the first NS - 1 stages go into the ring before the first barrier
        issue(ga, 0);
        issue(gb, 1);
        commit(ga, 0);
        issue(ga, 2);
        commit(gb, 1);
        issue(gb, 3);
        if (NS == 4) { commit(ga, 2); issue(ga, 4); }
        // after B(n): commit stage n + NS - 1, re-issue its registers two stages further
        for (int n = 0; n < NT; n += 2) {
            ring_barrier();
            if (n + NS - 1 < NT) { if (NS == 4) commit(gb, n + 3); else commit(ga, n + 2); }
            __builtin_amdgcn_sched_barrier(0);
            if (n + NS + 1 < NT) { if (NS == 4) issue(gb, n + 5); else issue(ga, n + 4); }
            if (n + 1 >= NT) break;
            ring_barrier();
            if (n + NS < NT) { if (NS == 4) commit(ga, n + 4); else commit(gb, n + 3); }
            __builtin_amdgcn_sched_barrier(0);
            if (n + NS + 2 < NT) { if (NS == 4) issue(ga, n + 6); else issue(gb, n + 5); }
        }
        if (ss[0] + ss[1] == 123.456f) p.cs[0] = ss[0];
    }
    __syncthreads();
    if (tid == 0) st[2] = __builtin_amdgcn_s_memrealtime();
    if (!(p.flags & F_OUT)) { if (tid == 0) st[3] = __builtin_amdgcn_s_memrealtime(); return; }

    // ---------------- way out: park the accumulators, sweep them out (cd, w, and the loss for the negatives)
    if (mfma_team) {
        const int wr = wave >> 1, wc = wave & 1, r = lane & 31, hf = lane >> 5;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < (HALF ? 1 : 2); ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = 64 * wr + 32 * i + 8 * (e >> 2) + 4 * hf + (e & 3);
                    const int col = (HALF ? 32 * wc : 64 * wc + 32 * j) + r;
                    Tfd[row * LDT + col] = acc[i][j][e];
                    Tcd[row * LDT + col] = accc[i][j][e];
                }
    }
    __syncthreads();
    float* cd_out = p.cd + (size_t)tile * P2;
    float* w_out = p.w + (size_t)tile * P2;
    float* loss_out = pset >= 2 ? p.loss + (size_t)tile * P2 : nullptr;
    if constexpr (!HALF) {
        for (int v = tid; v < (P2 + 3) / 4; v += NTHR) {
            f32x4 c4, f4;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = min(4 * v + k, P2 - 1), rr = e / P, cc = e - rr * P;
                c4[k] = Tcd[rr * LDT + cc];
                f4[k] = Tfd[rr * LDT + cc];
            }
            if (4 * v + 3 < P2) {
                __builtin_nontemporal_store(c4, reinterpret_cast<f32x4*>(cd_out + 4 * v));
                __builtin_nontemporal_store(f4, reinterpret_cast<f32x4*>(w_out + 4 * v));
                if (loss_out) __builtin_nontemporal_store(c4 * f4, reinterpret_cast<f32x4*>(loss_out + 4 * v));
            } else {
                for (int k = 0; 4 * v + k < P2; ++k) { cd_out[4 * v + k] = c4[k]; w_out[4 * v + k] = f4[k]; if (loss_out) loss_out[4 * v + k] = c4[k] * f4[k]; }
            }
        }
    } else {
        // my columns [c0, c1) of every row: 16-byte groups of the flat [P][P] layout, whole groups as vectors, the two straddling ones by element
        const int c0 = q0, c1 = min(q0 + 64, P);
        constexpr int GPR = 18;                                    // groups a row's run can touch (64 floats: <= 17 + 1)
        for (int it = tid; it < P * GPR; it += NTHR) {
            const int rr = it / GPR, gi = it - rr * GPR;
            const int e0 = rr * P + c0, e1 = rr * P + c1;          // my run of this row
            const int g = (e0 >> 2) + gi;
            if (4 * g >= e1) continue;
            f32x4 c4, f4;
            bool in[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = 4 * g + k;
                in[k] = e >= e0 && e < e1;
                const int cc = in[k] ? e - rr * P - c0 : 0;
                c4[k] = Tcd[rr * LDT + cc];
                f4[k] = Tfd[rr * LDT + cc];
            }
            if (in[0] && in[3]) {
                __builtin_nontemporal_store(c4, reinterpret_cast<f32x4*>(cd_out + 4 * g));
                __builtin_nontemporal_store(f4, reinterpret_cast<f32x4*>(w_out + 4 * g));
                if (loss_out) __builtin_nontemporal_store(c4 * f4, reinterpret_cast<f32x4*>(loss_out + 4 * g));
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (in[k]) { cd_out[4 * g + k] = c4[k]; w_out[4 * g + k] = f4[k]; if (loss_out) loss_out[4 * g + k] = c4[k] * f4[k]; }
            }
        }
    }
    if (tid == 0) st[3] = __builtin_amdgcn_s_memrealtime();
}

// ------------------------------------------------------------------------------------------------ host
static void make_taps(float x, float y, int H, int W, int* t, float* w)
{
    float ix = ((x + 1.f) * 0.5f) * (float)(W - 1), iy = ((y + 1.f) * 0.5f) * (float)(H - 1);
    ix = fminf((float)(W - 1), fmaxf(ix, 0.f));
    iy = fminf((float)(H - 1), fmaxf(iy, 0.f));
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - fx0, wx0 = 1.f - wx1, wy1 = iy - fy0, wy0 = 1.f - wy1;
    float nw = wx0 * wy0, ne = wx1 * wy0, sw = wx0 * wy1, se = wx1 * wy1;
    if (x1 > W - 1) { ne = se = 0.f; x1 = x0; }
    if (y1 > H - 1) { sw = se = 0.f; y1 = y0; }
    t[0] = y0 * W + x0; t[1] = y0 * W + x1; t[2] = y1 * W + x0; t[3] = y1 * W + x1;
    w[0] = nw; w[1] = ne; w[2] = sw; w[3] = se;
}

// workgroups [w0, w0 + n) <-> items [0, n): the r-th item preferring XCD x takes slot r of x while x has slots, the rest fill what is left
static void assign(const std::vector<int>& pref, int w0, std::vector<int>& wg_item)
{
    const int n = (int)pref.size();
    std::vector<std::vector<int>> slots(8);
    for (int w = w0; w < w0 + n; ++w) slots[w & 7].push_back(w);
    std::vector<size_t> used(8, 0);
    std::vector<int> overflow;
    for (int i = 0; i < n; ++i) {
        const int x = pref[i];
        if (used[x] < slots[x].size()) wg_item[slots[x][used[x]++]] = i;
        else overflow.push_back(i);
    }
    size_t o = 0;
    for (int x = 0; x < 8; ++x)
        while (used[x] < slots[x].size()) wg_item[slots[x][used[x]++]] = overflow[o++];
}

int main(int argc, char** argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 32;
    const int C = argc > 2 ? atoi(argv[2]) : 384;
    const int HW = argc > 3 ? atoi(argv[3]) : 28;
    const int launches = argc > 4 ? atoi(argv[4]) : 40;
    const bool json = argc > 5 && !strcmp(argv[5], "--json");      // bench.py: one JSON line with the two "all" variants only
    const int NJ = C / 128, NCH2 = C / 32, NPIX = HW * HW, NSET_IN = getenv("SKEL_SETS") ? atoi(getenv("SKEL_SETS")) : 4;
    if (!(C == 384 || C == 768) || B % 8 != 0 || NSETS * B > 256) { printf("C in {384, 768}, B a multiple of 8, 7 B <= 256\n"); return 1; }
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int n_tiles = NSETS * B;
    const size_t f_img = (size_t)(NPIX + 1) * C, c_img = (size_t)NPIX * K;       // features: token stride of the ViT (CLS token in front)
    const size_t f_elems = f_img * B, c_elems = c_img * B;
    std::vector<float*> d_f(NSET_IN), d_fp(NSET_IN), d_c(NSET_IN), d_cp(NSET_IN), d_cold(NSET_IN);
    const bool want_cold = getenv("SKEL_MODES") != nullptr;
    {
        std::vector<float> h(f_elems);
        srand(1);
        for (int s = 0; s < NSET_IN; ++s) {
            for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
            CK(hipMalloc(&d_f[s], f_elems * 4)); CK(hipMalloc(&d_fp[s], f_elems * 4));
            CK(hipMalloc(&d_c[s], c_elems * 4 + 64)); CK(hipMalloc(&d_cp[s], c_elems * 4 + 64));
            CK(hipMemcpy(d_f[s], h.data(), f_elems * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(d_fp[s], h.data(), f_elems * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(d_c[s], h.data(), c_elems * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(d_cp[s], h.data(), c_elems * 4, hipMemcpyHostToDevice));
            d_cold[s] = nullptr;
            if (want_cold) { CK(hipMalloc(&d_cold[s], f_img * 512 * 4)); CK(hipMemset(d_cold[s], 0, f_img * 512 * 4)); }
        }
    }
    // draws: coords1 / coords2 per image, 5 permutations without fixed points
    std::vector<int> h_taps((size_t)n_tiles * TP * 4, 0);
    std::vector<float> h_tapw((size_t)n_tiles * TP * 4, 0.f);
    std::vector<int> h_src(n_tiles);
    std::vector<std::vector<int>> perms(NEG, std::vector<int>(B));
    srand(7);
    for (int i = 0; i < NEG; ++i) {
        std::vector<int> pm(B);
        for (int k = 0; k < B; ++k) pm[k] = k;
        for (int k = B - 1; k > 0; --k) std::swap(pm[k], pm[rand() % (k + 1)]);
        for (int k = 0; k < B; ++k) perms[i][k] = pm[k] == k ? (pm[k] + 1) % B : pm[k];
    }
    std::vector<float> c1((size_t)B * P * 2), c2((size_t)B * P * 2);
    for (auto& v : c1) v = 2.f * rand() / RAND_MAX - 1.f;
    for (auto& v : c2) v = 2.f * rand() / RAND_MAX - 1.f;
    for (int t = 0; t < n_tiles; ++t) {
        const int pset = t / B, b = t % B;
        h_src[t] = pset < 2 ? b : perms[pset - 2][b];
        const float* cc = (pset == 0 ? c1.data() : c2.data()) + (size_t)b * P * 2;
        for (int q = 0; q < P; ++q) make_taps(cc[2 * q], cc[2 * q + 1], HW, HW, &h_taps[((size_t)t * TP + q) * 4], &h_tapw[((size_t)t * TP + q) * 4]);
    }
    std::vector<int> h_rank(n_tiles, 0);
    {
        std::vector<int> cnt(2 * B, 0);
        for (int t = B; t < n_tiles; ++t) { const int key = (t / B == 1 ? B : 0) + h_src[t]; h_rank[t] = cnt[key]++; }
    }
    int* d_rank;
    CK(hipMalloc(&d_rank, n_tiles * 4));
    CK(hipMemcpy(d_rank, h_rank.data(), n_tiles * 4, hipMemcpyHostToDevice));
    int4* d_taps; float4* d_tapw; int* d_src;
    CK(hipMalloc(&d_taps, h_taps.size() * 4)); CK(hipMalloc(&d_tapw, h_tapw.size() * 4)); CK(hipMalloc(&d_src, n_tiles * 4));
    CK(hipMemcpy(d_taps, h_taps.data(), h_taps.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tapw, h_tapw.data(), h_tapw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_src, h_src.data(), n_tiles * 4, hipMemcpyHostToDevice));

    SK p{};
    p.B = B; p.NCH2 = NCH2;
    p.f_sn = (long long)f_img; p.c_sn = (long long)c_img; p.f_sp = C; p.c_sp = K;
    p.taps = d_taps; p.tapw = d_tapw; p.tile_src = d_src; p.tile_rank = d_rank;
    CK(hipMalloc(&p.fs, (size_t)B * NCH2 * RS_SIDE)); CK(hipMalloc(&p.csf, (size_t)B * NKC * RS_SIDE));
    CK(hipMemset(p.fs, 0, (size_t)B * NCH2 * RS_SIDE)); CK(hipMemset(p.csf, 0, (size_t)B * NKC * RS_SIDE));
    CK(hipMalloc(&p.cs, (size_t)n_tiles * TP * LDK * 4));
    CK(hipMalloc(&p.o_taps, (size_t)n_tiles * TP * 16)); CK(hipMalloc(&p.o_tapw, (size_t)n_tiles * TP * 16));
    CK(hipMalloc(&p.anchor_cnt, (size_t)B * 64 * 4));
    CK(hipMalloc(&p.cd, (size_t)n_tiles * P2 * 4)); CK(hipMalloc(&p.w, (size_t)n_tiles * P2 * 4)); CK(hipMalloc(&p.loss, (size_t)n_tiles * P2 * 4));
    const int grid_full = std::max(n_tiles, cus & ~7), n_anchor_wg = 2 * B, grid_half = n_anchor_wg + 2 * n_tiles;
    CK(hipMalloc(&p.stamps, (size_t)std::max(grid_full, grid_half) * 4 * 8));
    // work items
    std::vector<int> wg_full(grid_full, -1), wg_half(grid_half, -1);
    {
        std::vector<int> pref(n_tiles);
        for (int t = 0; t < n_tiles; ++t) pref[t] = h_src[t] & 7;
        assign(pref, 0, wg_full);
        std::vector<int> pref2(2 * n_tiles);
        for (int t = 0; t < 2 * n_tiles; ++t) pref2[t] = h_src[t >> 1] & 7;
        assign(pref2, n_anchor_wg, wg_half);
    }
    int *d_wg_full, *d_wg_half;
    CK(hipMalloc(&d_wg_full, grid_full * 4)); CK(hipMalloc(&d_wg_half, grid_half * 4));
    CK(hipMemcpy(d_wg_full, wg_full.data(), grid_full * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_wg_half, wg_half.data(), grid_half * 4, hipMemcpyHostToDevice));

    const int lds_full = 1024 + 4 * (RS_SIDE + 128 * 128), lds_half = 1024 + 3 * (RS_SIDE + 64 * 128);
    auto kfull = NJ == 3 ? skeleton_kernel<0, 3> : skeleton_kernel<0, 6>;
    auto khalf = NJ == 3 ? skeleton_kernel<1, 3> : skeleton_kernel<1, 6>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfull), hipFuncAttributeMaxDynamicSharedMemorySize, lds_full));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(khalf), hipFuncAttributeMaxDynamicSharedMemorySize, lds_half));
    {
        int nb = 0;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, khalf, 512, lds_half));
        hipFuncAttributes fa;
        CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(khalf)));
        if (!json) printf("# %d CUs; HALF kernel: %d workgroups per CU (occupancy API), %d VGPRs, %zu B spill; grids: full %d x 768, half %d x 512\n", cus, nb,
               fa.numRegs, (size_t)fa.localSizeBytes, grid_full, grid_half);
        CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kfull)));
        if (!json) printf("# FULL kernel: %d VGPRs, %zu B spill\n", fa.numRegs, (size_t)fa.localSizeBytes);
    }
    const double alg = 4.0 * (2.0 * B * C * NPIX + 2.0 * B * K * NPIX + 2.0 * B * P * 2) + 8.0 * NEG * B + 4.0 * ((double)NSETS * B * P2 + (double)NEG * B * P2) + 12;
    if (!json) printf("# B=%d C=%d %dx%d: algorithmic bytes %.2f MB (roofline 8 TB/s: %.1f us)\n", B, C, HW, HW, alg / 1e6, alg / 8e6);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Var { const char* name; int half; int flags; int rot; int mode; };
    const Var vars[] = {
        {"FULL all", 0, F_ALL, 0}, {"FULL all rot1", 0, F_ALL, 1}, {"FULL all rot2", 0, F_ALL, 2}, {"FULL all rot5", 0, F_ALL, 5},
        {"HALF all", 1, F_ALL, 0}, {"HALF all rot2", 1, F_ALL, 2},
        {"FULL no-wait (p1 off)", 0, F_ALL & ~F_P1, 0}, {"FULL no-wait rot2", 0, F_ALL & ~F_P1, 2}, {"HALF no-wait (p1 off)", 1, F_ALL & ~F_P1, 0}, {"HALF no-wait rot2", 1, F_ALL & ~F_P1, 2},
        {"FULL no mfma", 0, F_ALL & ~F_MFMA, 0}, {"HALF no mfma", 1, F_ALL & ~F_MFMA, 0},
        {"FULL gather only", 0, F_GATHER, 0}, {"FULL gather only rot1", 0, F_GATHER, 1}, {"FULL gather only rot2", 0, F_GATHER, 2}, {"FULL gather only rot5", 0, F_GATHER, 5},
        {"HALF gather only", 1, F_GATHER, 0}, {"HALF gather only rot2", 1, F_GATHER, 2},
        {"FULL gather + astream", 0, F_GATHER | F_ASTREAM, 0}, {"FULL gather + astream rot2", 0, F_GATHER | F_ASTREAM, 2}, {"HALF gather + astream", 1, F_GATHER | F_ASTREAM, 0},
        {"FULL out only", 0, F_OUT, 0}, {"HALF out only", 1, F_OUT, 0},
        {"FULL p1 only", 0, F_P1, 0}, {"HALF p1 only", 1, F_P1, 0},
        {"FULL gather mode1 all-hot", 0, F_GATHER, 0, 1}, {"FULL gather mode2 inter cold", 0, F_GATHER, 0, 2}, {"FULL gather mode3 mixed waves", 0, F_GATHER, 0, 3}, {"FULL gather mode4 all cold", 0, F_GATHER, 0, 4},
        {"HALF gather mode1 all-hot", 1, F_GATHER, 0, 1}, {"HALF gather mode3 mixed waves", 1, F_GATHER, 0, 3}, {"HALF gather mode4 all cold", 1, F_GATHER, 0, 4},
        {"FULL all but out", 0, F_ALL & ~F_OUT, 0}, {"FULL all but out rot2", 0, F_ALL & ~F_OUT, 2}, {"HALF all but out", 1, F_ALL & ~F_OUT, 0},
    };
    if (json) printf("{\"B\": %d, \"C\": %d, \"HW\": %d, ", B, C, HW);
    for (int rep = 0; rep < 2; ++rep)
        for (const Var& v : vars) {
            p.flags = v.flags;
            p.rot_mul = v.rot;
            p.src_mode = v.mode;
            if (v.mode && !want_cold) continue;
            if (json && !(v.flags == F_ALL && v.rot == 0 && v.mode == 0)) continue;
            if (json && v.half && grid_half > 2 * cus) continue;
            if (!v.mode && want_cold && !(v.flags == F_GATHER && v.rot == 0)) continue;
            p.wg_item = v.half ? d_wg_half : d_wg_full;
            p.n_anchor_wg = n_anchor_wg;
            const int grid = v.half ? grid_half : grid_full;
            std::vector<float> ms;
            std::vector<unsigned long long> hs((size_t)grid * 4);
            double ph[3] = {0, 0, 0}, ph_max[3] = {0, 0, 0};
            for (int it = 0; it < launches + 4; ++it) {
                const int s = it % NSET_IN;
                p.feats = d_f[s] + C; p.feats_pos = d_fp[s] + C; p.code = d_c[s]; p.code_pos = d_cp[s]; p.cold = d_cold[s] ? d_cold[s] + C : nullptr;
                CK(hipMemsetAsync(p.anchor_cnt, 0, (size_t)B * 64 * 4, 0));
                CK(hipEventRecord(e0, 0));
                if (v.half) hipLaunchKernelGGL(khalf, dim3(grid), dim3(512), lds_half, 0, p);
                else hipLaunchKernelGGL(kfull, dim3(grid), dim3(768), lds_full, 0, p);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                CK(hipGetLastError());
                float t;
                CK(hipEventElapsedTime(&t, e0, e1));
                if (it >= 4) ms.push_back(t);
                if (it == launches + 3) {
                    CK(hipMemcpy(hs.data(), p.stamps, hs.size() * 8, hipMemcpyDeviceToHost));
                    unsigned long long t0 = ~0ull;
                    for (int w = 0; w < grid; ++w) t0 = std::min(t0, hs[(size_t)w * 4]);
                    std::vector<double> a[3];
                    const int* items = v.half ? wg_half.data() : wg_full.data();
                    for (int w = 0; w < grid; ++w) {
                        if (items[w] < 0) continue;
                        for (int k = 0; k < 3; ++k) a[k].push_back((double)(hs[(size_t)w * 4 + 1 + k] - t0) / 100.0);
                    }
                    if (rep == 1 && getenv("SKEL_VERBOSE") && strstr(getenv("SKEL_VERBOSE"), v.name)) {
                        printf("  [%s] loop end per XCD (median / max, n; tiles whose source image lives on another XCD):", v.name);
                        for (int x = 0; x < 8; ++x) {
                            std::vector<double> e;
                            int foreign = 0;
                            for (int w = x; w < grid; w += 8) {
                                if (items[w] < 0) continue;
                                const int t = v.half ? items[w] >> 1 : items[w];
                                if (t < B) continue;
                                e.push_back((double)(hs[(size_t)w * 4 + 2] - t0) / 100.0);
                                foreign += (h_src[t] & 7) != x;
                            }
                            std::sort(e.begin(), e.end());
                            if (!e.empty()) printf("  x%d %.1f/%.1f n%zu f%d", x, e[e.size() / 2], e.back(), e.size(), foreign);
                        }
                        printf("\n  slowest:");
                        std::vector<std::pair<double, int>> sl;
                        for (int w = 0; w < grid; ++w) if (items[w] >= 0) sl.push_back({(double)(hs[(size_t)w * 4 + 2] - t0) / 100.0, w});
                        std::sort(sl.begin(), sl.end());
                        for (size_t i = sl.size() - std::min<size_t>(12, sl.size()); i < sl.size(); ++i) {
                            const int w = sl[i].second, t = v.half ? items[w] >> 1 : items[w];
                            printf(" [%.1f wg%d x%d p%d src%d(x%d) rank%d]", sl[i].first, w, w & 7, t / B, h_src[t], h_src[t] & 7, h_rank[t]);
                        }
                        printf("\n");
                    }
                    for (int k = 0; k < 3; ++k) {
                        std::sort(a[k].begin(), a[k].end());
                        ph[k] = a[k][a[k].size() / 2];
                        ph_max[k] = a[k].back();
                    }
                }
            }
            std::sort(ms.begin(), ms.end());
            if (rep == 1 && json)
                printf("\"%s\": {\"us\": %.2f, \"p10\": %.2f, \"p90\": %.2f, \"last_item_out_us\": %.2f}, ", v.half ? "half" : "full",
                       ms[ms.size() / 2] * 1e3, ms[ms.size() / 10] * 1e3, ms[ms.size() * 9 / 10] * 1e3, ph_max[2]);
            else if (rep == 1)
                printf("%-28s  %6.1f us (p10 %.1f, p90 %.1f)   anchors ready %5.1f / %5.1f   loop end %5.1f / %5.1f   out %5.1f / %5.1f  (median / slowest work item, us from the first start)\n",
                       v.name, ms[ms.size() / 2] * 1e3, ms[ms.size() / 10] * 1e3, ms[ms.size() * 9 / 10] * 1e3, ph[0], ph_max[0], ph[1], ph_max[1], ph[2], ph_max[2]);
        }
    if (json) printf("\"cus\": %d, \"launches\": %d, \"input_sets\": %d}\n", cus, launches, NSET_IN);
    return 0;
}
