// All-pairs cosine top-k (STEGO's KNN precompute) for gfx950: fused GEMM + running top-k, the [N,N] similarity
// matrix is never materialised.
//
//   reference: src/precompute_knns.py:86-96  -  16 row blocks of  sims = blk . X^T ;  topk(sims, 30)[1]
//              (:15-21 get_feats: rows are F.normalize'd mean-pooled DINO features; the consumer skips rank 0 = self,
//               data.py:524)
//
// Three launches:
//   knn_prep_kernel  - (optional F.normalize, eps 1e-12) + fp32 -> split-fp16 (hi, lo) in the LDS-image layout
//                      [row block of 128][64-column chunk][hi|lo][128][72], zero padded: every later tile load is a
//                      linear global_load_lds copy, no VALU.
//   knn_tile_kernel  - persistent workgroups, each an even share of the (128-query block, 128-column tile) units,
//                      walked query block by query block (a block is cut at most where two workgroups meet).
//                      sims on v_mfma_f32_32x32x16_f16 as hi*hi + hi*lo + lo*hi (fp32 accumulate, ~2e-7 abs on a
//                      cosine: the order of neighbours matches fp32 except for ties at that level; plain fp32 MFMA
//                      runs at the VALU rate on gfx950 and would take 3x longer).  Wave w owns query rows
//                      32w..32w+31 and all 128 columns, so in the MFMA C layout every register index r holds two
//                      complete query rows (lanes 0-31 / 32-63 = 32 of the row's columns): the running top-k of a row
//                      is ONE VGPR pair (value, index) spread over the 32 lanes of its half-wave, sorted descending.
//                      Per tile: one compare-and-ballot per row pair against the current k-th value; the (rare:
//                      ~k ln(N/k) per row in total) hits are inserted with a ballot-popcount position and a
//                      one-lane shift - registers only, no LDS, no atomics, deterministic.
//   knn_merge_kernel - merges the per-segment lists of a row (k-way, by repeated arg-max over the list heads).
// Ties: torch.topk leaves the order of equal values unspecified; so does this.
#include <cstdlib>
#include "corr_common.h"
#include "host_util.h"

namespace stego {

constexpr int KNN_TQ = 128;                         // queries per workgroup = rows of an image block
constexpr int KNN_SIDE = 2 * TP * LDH * 2;          // bytes of one chunk image: hi[128][72] + lo[128][72] fp16
constexpr int KNN_MAXK = 32;                        // a list lives in the 32 lanes of a half-wave

struct KnnParams {
    const float* X;             // [N][ldx] fp32
    void* img;                  // [nblk][NCH][2][128][LDH] fp16
    float* part_val;            // [NS][Nq_pad][k]
    int* part_idx;
    long long* out_idx;         // [q_count][k]
    float* out_val;             // optional
    long long N, ldx;
    int D, NCH, k, normalize;
    long long q_begin, q_count; // query rows [q_begin, q_begin + q_count)
    int nblk, NS;               // NS = max partial lists (segments) per query block
    long long units_total, units_per_wg;      // work units = (query block, database tile) pairs, split evenly over the WGs
    int debug;                  // STEGO_DEBUG_KNN: 1 skip the selection, 2 skip the MFMAs, 4 count slow-path entries
    unsigned long long* counters;   // [2] (debug 4): row pairs that took the slow path, insertion iterations
};

// ---------------------------------------------------------------------------------------------- prep
// grid = nblk, block = 256: two threads per row.
__global__ void __launch_bounds__(NTHREADS) knn_prep_kernel(const KnnParams prm)
{
    const int tid = threadIdx.x;
    const int rl = tid >> 1, half = tid & 1;
    const long long row = (long long)blockIdx.x * TP + rl;
    const bool rv = row < prm.N;
    const float* x = prm.X + (rv ? row : 0) * prm.ldx;
    const int D = prm.D;
    float inv = 1.f;
    if (prm.normalize) {
        float ss = 0.f;
        if (rv) for (int c = half; c < D; c += 2) ss += x[c] * x[c];
        ss += __shfl_xor(ss, 1, 64);
        inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);                 // F.normalize default eps (precompute_knns.py:19)
    }
    half_t* base = static_cast<half_t*>(prm.img) + (size_t)blockIdx.x * prm.NCH * (2 * TP * LDH);
    for (int ch = 0; ch < prm.NCH; ++ch) {
        half_t* dh = base + (size_t)ch * (2 * TP * LDH) + rl * LDH;
        half_t* dl = dh + TP * LDH;
        for (int c2 = half * 2; c2 < LDH; c2 += 4) {             // pairs of columns, interleaved between the two threads
            const int c = ch * KC + c2;
            const float v0 = (rv && c2 < KC && c < D) ? x[c] * inv : 0.f;
            const float v1 = (rv && c2 + 1 < KC && c + 1 < D) ? x[c + 1] * inv : 0.f;
            unsigned h, l;
            split_f16_pair(v0, v1, h, l);
            *reinterpret_cast<unsigned*>(dh + c2) = h;
            *reinterpret_cast<unsigned*>(dl + c2) = l;
        }
    }
}

// ---------------------------------------------------------------------------------------------- tile + top-k
__device__ __forceinline__ void knn_copy(const unsigned char* __restrict__ gsrc, unsigned char* lds_dst, int wave, int lane)
{
    for (int pc = wave; pc < KNN_SIDE / 1024; pc += 4)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + (size_t)pc * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(lds_dst + pc * 1024), 16, 0, 0);
}

// Round 4: the AREG path (D <= 384) keeps KNN_NST database chunks in LDS and has three copies in flight.  With two stages the copy of
// chunk g + 1 was issued when chunk g's MFMAs started (0.64 us of work) and awaited right after them: every chunk waited ~0.9 us for
// its copy (MFMA-only run 22 ms against 9.2 ms of matrix-core time; copies-only 11 ms).  LDS-DMA from inline asm, as in the fused forward:
// the compiler neither counts these copies nor drains them in front of a barrier; the waits are explicit and counted (9 pieces per wave
// and chunk, completed in order).  And the sixth query chunk moved from registers to a resident LDS image: with all 192 fragment registers
// the kernel sat at 255 VGPRs + 106 spilled to AGPRs and the compiler fetched the lo halves of the B fragments one by one through ONE
// register quad, an LDS round trip in front of 4 of every 12 MFMAs (ISA: ds_read_b128 v[48:51] ; s_waitcnt lgkmcnt(0) ; v_mfma ...).
constexpr int KNN_NST = 3;                          // + one resident A chunk: 4 x 36 KB of LDS
constexpr int KNN_RCH = 5;                          // query chunks held in registers (160 VGPRs); the sixth lives in LDS
__device__ __forceinline__ void knn_dma_piece(const unsigned char* gsrc_lane, unsigned lds_addr)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc_lane), "s"(lds_addr) : "memory");
}
static_assert(KNN_SIDE % 4096 == 0, "whole 1 KB pieces per wave");

// wave-row layout: this wave's 32 query rows x 128 columns: acc[ni] = 32x32 block of columns 32 ni .. 32 ni + 31
__device__ __forceinline__ void knn_mma_chunk(const half_t* __restrict__ As, const half_t* __restrict__ Bs, f32x16 (&acc)[4],
                                              int lane, int wave)
{
    constexpr int LO = TP * LDH;
    const int r = lane & 31, half = lane >> 5;
    const half_t* ap = As + (32 * wave + r) * LDH + 8 * half;
    const half_t* bp = Bs + r * LDH + 8 * half;
#pragma unroll
    for (int kk = 0; kk < KC; kk += 16) {
        const f16x8 ah = *reinterpret_cast<const f16x8*>(ap + kk), al = *reinterpret_cast<const f16x8*>(ap + LO + kk);
        f16x8 bh[4], bl[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            bh[ni] = *reinterpret_cast<const f16x8*>(bp + ni * 32 * LDH + kk);
            bl[ni] = *reinterpret_cast<const f16x8*>(bp + ni * 32 * LDH + LO + kk);
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[ni], acc[ni], 0, 0, 0);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[ni], acc[ni], 0, 0, 0);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[ni], acc[ni], 0, 0, 0);
    }
}

// A operand from registers (the query block never changes: re-streaming it with every database tile doubled the
// L2 traffic, and the kernel is bound by that traffic, not by the MFMAs)
__device__ __forceinline__ void knn_mma_chunk_areg(const f16x8 (&ah)[KC / 16], const f16x8 (&al)[KC / 16],
                                                   const half_t* __restrict__ Bs, f32x16 (&acc)[4], int lane)
{
    constexpr int LO = TP * LDH;
    const int r = lane & 31, half = lane >> 5;
    const half_t* bp = Bs + r * LDH + 8 * half;
#pragma unroll
    for (int ks = 0; ks < KC / 16; ++ks) {
        const int kk = 16 * ks;
        f16x8 bh[4], bl[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            bh[ni] = *reinterpret_cast<const f16x8*>(bp + ni * 32 * LDH + kk);
            bl[ni] = *reinterpret_cast<const f16x8*>(bp + ni * 32 * LDH + LO + kk);
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], bh[ni], acc[ni], 0, 0, 0);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bl[ni], acc[ni], 0, 0, 0);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bh[ni], acc[ni], 0, 0, 0);
    }
}

__device__ __forceinline__ float readlane_f(float v, int l)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

constexpr int KNN_AREG_CHUNKS = 6;          // D <= 384: the query block's MFMA fragments live in 160 VGPRs + one resident LDS chunk

// grid = (query blocks, NS); block = 256.  AREG: LDS = 2 stages x B chunk, A in registers (D <= 384);
// otherwise 2 stages x (A chunk + B chunk).
template <bool AREG>
__global__ void __launch_bounds__(NTHREADS) knn_tile_kernel(const KnnParams prm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NCH = prm.NCH, k = prm.k;
    const unsigned char* img = static_cast<const unsigned char*>(prm.img);
    const int slot = lane & 31;
    const bool upper = lane >= 32;
    // Persistent workgroup: an even share of the (query block, database tile) units, walked query block by query
    // block.  With one workgroup per (query block, slice) the grid was 3.05 rounds of the chip for N = 100 k and every
    // slice re-warmed its lists; here every CU gets the same number of tiles and a query block is cut at most where
    // two workgroups meet (its partial lists go to consecutive slots and are merged afterwards).
    const long long u_begin = (long long)blockIdx.x * prm.units_per_wg;
    const long long u_end = min(prm.units_total, u_begin + prm.units_per_wg);
    for (long long u = u_begin; u < u_end;) {
    const int qb = (int)(u / prm.nblk);                                // query block relative to q_begin
    const int tile0 = (int)(u - (long long)qb * prm.nblk);
    const int tile1 = (int)min((long long)prm.nblk, tile0 + (u_end - u));
    u += tile1 - tile0;
    const int seg = (int)(((long long)qb * prm.nblk + tile0) / prm.units_per_wg - ((long long)qb * prm.nblk) / prm.units_per_wg);
    const int qblk = (int)(prm.q_begin / TP) + qb;                     // q_begin is a multiple of 128 (host-checked)
    const unsigned char* Aimg = img + (size_t)qblk * NCH * KNN_SIDE;
    __syncthreads();                                                   // the previous segment is done with the stage buffers

    // running top-k: register index rr <-> query rows  32 wave + (rr&3) + 8 (rr>>2) + 4 (lane>>5)
    float lv[16];
    int li[16];
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) { lv[rr] = -INFINITY; li[rr] = -1; }

    const int nstage = (tile1 - tile0) * NCH;
    constexpr int STAGE = AREG ? KNN_SIDE : 2 * KNN_SIDE;
    const unsigned smem_addr = (unsigned)(uintptr_t)((__attribute__((address_space(3))) const unsigned char*)smem);
    auto issue = [&](int g) {
        const int t = tile0 + g / NCH, c = g - (g / NCH) * NCH;
        if constexpr (AREG) {
            const unsigned char* src = img + ((size_t)t * NCH + c) * KNN_SIDE + lane * 16;
            const unsigned dst = smem_addr + (g % KNN_NST) * STAGE;
#pragma unroll
            for (int i = 0; i < KNN_SIDE / 4096; ++i) knn_dma_piece(src + (wave + 4 * i) * 1024, dst + (wave + 4 * i) * 1024);
        } else {
            unsigned char* dst = smem + (g & 1) * STAGE;
            knn_copy(Aimg + (size_t)c * KNN_SIDE, dst, wave, lane);
            knn_copy(img + ((size_t)t * NCH + c) * KNN_SIDE, dst + KNN_SIDE, wave, lane);
        }
    };
    f16x8 Ah[AREG ? KNN_RCH : 1][KC / 16], Al[AREG ? KNN_RCH : 1][KC / 16];
    if constexpr (AREG) {
        const int r = lane & 31, half = lane >> 5;
#pragma unroll
        for (int c = 0; c < KNN_RCH; ++c) {
            const half_t* ap = reinterpret_cast<const half_t*>(Aimg + (size_t)min(c, NCH - 1) * KNN_SIDE) + (32 * wave + r) * LDH + 8 * half;
#pragma unroll
            for (int ks = 0; ks < KC / 16; ++ks) {
                Ah[c][ks] = *reinterpret_cast<const f16x8*>(ap + 16 * ks);
                Al[c][ks] = *reinterpret_cast<const f16x8*>(ap + TP * LDH + 16 * ks);
            }
        }
    }
    if constexpr (AREG) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the query fragments: the counted waits below start from zero
        if (NCH > KNN_RCH) {                                           // the resident query chunk (completed in front of stage 0)
            const unsigned char* src = Aimg + (size_t)KNN_RCH * KNN_SIDE + lane * 16;
            const unsigned dst = smem_addr + KNN_NST * STAGE;
#pragma unroll
            for (int i = 0; i < KNN_SIDE / 4096; ++i) knn_dma_piece(src + (wave + 4 * i) * 1024, dst + (wave + 4 * i) * 1024);
        }
        for (int s0 = 0; s0 < KNN_NST - 1 && s0 < nstage; ++s0) issue(s0);
    } else if (nstage > 0) issue(0);
    f32x16 acc[4];
    int g = 0;
    for (int t = tile0; t < tile1; ++t) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ni][e] = 0.f;
        if constexpr (AREG) {
#pragma unroll
            for (int c = 0; c < KNN_AREG_CHUNKS; ++c) {
                if (c < NCH) {
                    // my pieces of stage g have landed (the copies of g + 1 and g + 2 may still fly), then everybody's; the slot of stage
                    // g - 1 - which everyone has finished reading - takes the copy of stage g + 3
                    const int ahead = nstage - 1 - g;
                    if (ahead >= 1) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    if (g + KNN_NST - 1 < nstage) issue(g + KNN_NST - 1);
                    const half_t* Bst = reinterpret_cast<const half_t*>(smem + (g % KNN_NST) * STAGE);
                    if (!(prm.debug & 2)) {
                        // (c is a constant after unrolling; the index is clamped for the branch that is never taken)
                        if (c < KNN_RCH) knn_mma_chunk_areg(Ah[c < KNN_RCH ? c : 0], Al[c < KNN_RCH ? c : 0], Bst, acc, lane);
                        else knn_mma_chunk(reinterpret_cast<const half_t*>(smem + KNN_NST * STAGE), Bst, acc, lane, wave);
                    }
                    ++g;
                }
            }
        } else {
            for (int c = 0; c < NCH; ++c, ++g) {
                sync_after_lds_dma();                     // stage g landed; stage g-1 is free
                if (g + 1 < nstage) issue(g + 1);
                const unsigned char* st = smem + (g & 1) * STAGE;
                knn_mma_chunk(reinterpret_cast<const half_t*>(st), reinterpret_cast<const half_t*>(st + KNN_SIDE), acc, lane, wave);
            }
        }
        // ---- selection.  Columns past N (zero rows of the last block) must never be chosen.
        if (prm.debug & 1) { lv[0] += acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3]; continue; }
        const long long col0 = (long long)t * TP;
        bool cv[4];                                    // column of this lane in block ni exists (only the last tile has holes)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) cv[ni] = col0 + 32 * ni + slot < prm.N;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            // everything wave-uniform goes through SGPRs (v_readlane / s_ff1 / s_bcnt1): the first version used
            // ds_bpermute shuffles here and the selection cost more than the MFMAs (32 of 55 ms at N = 100 k)
            const float thr = upper ? readlane_f(lv[rr], 32 + k - 1) : readlane_f(lv[rr], k - 1);      // current k-th best of my row
            // one candidate mask per column block (round 4: the insertion used to be a non-inlined call that walked the four blocks with
            // fresh ballots and threshold reads per candidate: ~1600 cycles per entry with one wave per SIMD, 12 of the 36 ms at N = 100 k)
            unsigned long long cm[4];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) cm[ni] = __ballot(cv[ni] && acc[ni][rr] > thr);
            if ((cm[0] | cm[1] | cm[2] | cm[3]) == 0ull || (prm.debug & 8)) continue;       // the common case (debug 8: never insert)
            if ((prm.debug & 4) && lane == 0) atomicAdd(prm.counters, 1ull);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                unsigned long long m = cm[ni];
                const float a = acc[ni][rr];
                while (m != 0ull) {                                          // (uniform) one candidate per half-wave and turn
                    const unsigned mlo = (unsigned)m, mhi = (unsigned)(m >> 32);
                    const int c0 = mlo ? __builtin_ctz(mlo) : 0, c1 = mhi ? __builtin_ctz(mhi) : 0;
                    const float x0 = readlane_f(a, c0), x1 = readlane_f(a, 32 + c1);
                    const bool have = upper ? mhi != 0u : mlo != 0u;
                    const float x = upper ? x1 : x0;
                    const int xi = (int)col0 + 32 * ni + (upper ? c1 : c0);
                    // position = number of entries >= x in my half; beyond k - 1 the candidate has been overtaken meanwhile
                    const unsigned long long ge = __ballot(lv[rr] >= x);
                    const int pos = upper ? __builtin_popcount((unsigned)(ge >> 32)) : __builtin_popcount((unsigned)ge);
                    const float upv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, lv[rr]), 0x138, 0xf, 0xf, false));
                    const int upi = __builtin_amdgcn_update_dpp(0, li[rr], 0x138, 0xf, 0xf, false);      // wave_shr:1
                    if (have && pos < k) {
                        if (slot == pos) { lv[rr] = x; li[rr] = xi; }
                        else if (slot > pos) { lv[rr] = upv; li[rr] = upi; }
                    }
                    m &= ~((mlo ? 1ull << c0 : 0ull) | (mhi ? 1ull << (32 + c1) : 0ull));
                }
            }
        }
    }
    // ---- write the partial lists of this segment (sorted descending)
    const long long qrow_base = (long long)qb * TP + 32 * wave;               // relative to q_begin
    const long long nq_pad = ((prm.q_count + TP - 1) / TP) * TP;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
        const long long qr = qrow_base + (rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5);
        if (slot < k) {
            prm.part_val[((size_t)seg * nq_pad + qr) * k + slot] = lv[rr];
            prm.part_idx[((size_t)seg * nq_pad + qr) * k + slot] = li[rr];
        }
    }
    }       // segments of this workgroup
}

// ---------------------------------------------------------------------------------------------- merge
// one half-wave per query row: lane s < NS walks segment s's sorted list; k rounds of arg-max over the heads.
__global__ void __launch_bounds__(NTHREADS) knn_merge_kernel(const KnnParams prm)
{
    const int tid = threadIdx.x, lane = tid & 63, s = lane & 31;
    const long long q = (long long)blockIdx.x * (NTHREADS / 32) + (tid >> 5);
    if (q >= prm.q_count) return;
    const int k = prm.k;
    const long long nq_pad = (long long)((prm.q_count + TP - 1) / TP) * TP;
    const long long ub = (q / TP) * (long long)prm.nblk;                 // first unit of this row's query block
    const int NS = (int)((ub + prm.nblk - 1) / prm.units_per_wg - ub / prm.units_per_wg) + 1;     // its segments
    int p = 0;                                                   // my list's head
    const float* pv = prm.part_val + ((size_t)min(s, NS - 1) * nq_pad + q) * k;
    const int* pi = prm.part_idx + ((size_t)min(s, NS - 1) * nq_pad + q) * k;
    for (int out = 0; out < k; ++out) {
        float v = (s < NS && p < k) ? pv[p] : -INFINITY;
        int who = s;
        float best = v;
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1) {                      // arg-max (ties: lower segment first)
            const float ov = __shfl_xor(best, m, 32);
            const int ow = __shfl_xor(who, m, 32);
            if (ov > best || (ov == best && ow < who)) { best = ov; who = ow; }
        }
        const int idx = __shfl((s < NS && p < k) ? pi[p] : -1, (lane & 32) + who, 64);
        if (s == 0) {
            prm.out_idx[q * k + out] = (long long)idx;
            if (prm.out_val) prm.out_val[q * k + out] = best;
        }
        if (s == who) ++p;
    }
}

// ---------------------------------------------------------------------------------------------- host side
static int knn_cus() { return device_cu_count(); }

// units per workgroup: an even split over the CUs, but never so fine that a query block has more than 32 segments
static void knn_partition(long long q_count, int nblk, long long* units_total, long long* units_per_wg, int* max_segments)
{
    const long long nqb = (q_count + TP - 1) / TP;
    const long long T = nqb * nblk;
    long long U = (T + knn_cus() - 1) / knn_cus();
    const long long umin = (nblk + 30) / 31;
    if (U < umin) U = umin;
    *units_total = T;
    *units_per_wg = U;
    *max_segments = (int)((nblk + U - 1) / U) + 1;
}

size_t knn_workspace_bytes(long long N, int D, int k, long long q_count)
{
    const long long nblk = (N + TP - 1) / TP;
    const int NCH = (D + KC - 1) / KC;
    const long long nq_pad = ((q_count + TP - 1) / TP) * TP;
    long long T, U;
    int ns;
    knn_partition(q_count, (int)nblk, &T, &U, &ns);
    size_t b = (size_t)nblk * NCH * KNN_SIDE;
    b = (b + 255) & ~(size_t)255;
    b += (size_t)ns * nq_pad * k * 8;
    return b + 512;
}

hipError_t launch_knn(const float* X, long long N, int D, long long ldx, int k, int normalize, long long q_begin,
                      long long q_count, long long* out_idx, float* out_val, void* ws, hipStream_t stream)
{
    KnnParams prm{};
    prm.X = X; prm.N = N; prm.D = D; prm.ldx = ldx; prm.k = k; prm.normalize = normalize;
    prm.q_begin = q_begin; prm.q_count = q_count; prm.out_idx = out_idx; prm.out_val = out_val;
    prm.nblk = (int)((N + TP - 1) / TP);
    prm.NCH = (D + KC - 1) / KC;
    knn_partition(q_count, prm.nblk, &prm.units_total, &prm.units_per_wg, &prm.NS);
    prm.debug = knob(KNOB_DEBUG_KNN);
    const long long nq_pad = ((q_count + TP - 1) / TP) * TP;
    unsigned char* w = static_cast<unsigned char*>(ws);
    prm.img = w;
    size_t off = ((size_t)prm.nblk * prm.NCH * KNN_SIDE + 255) & ~(size_t)255;
    prm.part_val = reinterpret_cast<float*>(w + off);
    prm.part_idx = reinterpret_cast<int*>(w + off + (size_t)prm.NS * nq_pad * k * 4);
    prm.counters = reinterpret_cast<unsigned long long*>(w + off + (size_t)prm.NS * nq_pad * k * 8);      // 256 B of slack

    hipLaunchKernelGGL(knn_prep_kernel, dim3(prm.nblk), dim3(NTHREADS), 0, stream, prm);
    const bool areg = prm.NCH <= KNN_AREG_CHUNKS;
    const int lds = areg ? (KNN_NST + 1) * KNN_SIDE : 4 * KNN_SIDE;
    hipError_t ea = ensure_dynamic_lds(areg ? reinterpret_cast<const void*>(&knn_tile_kernel<true>)
                                            : reinterpret_cast<const void*>(&knn_tile_kernel<false>), lds);
    if (ea != hipSuccess) return ea;
    const dim3 grid((unsigned)((prm.units_total + prm.units_per_wg - 1) / prm.units_per_wg));
    if (areg) hipLaunchKernelGGL(knn_tile_kernel<true>, grid, dim3(NTHREADS), lds, stream, prm);
    else hipLaunchKernelGGL(knn_tile_kernel<false>, grid, dim3(NTHREADS), lds, stream, prm);
    hipLaunchKernelGGL(knn_merge_kernel, dim3((unsigned)((q_count + NTHREADS / 32 - 1) / (NTHREADS / 32))), dim3(NTHREADS), 0,
                       stream, prm);
    return hipGetLastError();
}

}  // namespace stego
