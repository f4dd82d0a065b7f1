// ROUND 6 ATTEMPT, NOT BUILT INTO THE LIBRARY (kept for the record; the product kernel is csrc/dense_stream.hip).
// Parity green (tests/test_dense_corr.py, 15 / 15) behind dense_prep_kernel; measured [32,784,784], C = 384: prep 27.8 + this kernel 99 us
// (84.9 .. 106 over 22 launches; dense_stats + dense_stream: 9.9 + 86.6).  Stamps: the fastest workgroup needs 9.5 us per block of 6 chunks
// (1.6 us per chunk) with NOTHING but copies, fragment reads and MFMAs in the chunk - so dense_stream's 1.7 us per chunk is not its VALU work
// either: what a chunk costs beyond its 1.05 us of MFMAs is the LDS - four waves each read the whole 32 KB chunk as 16-byte fragments
// (128 KB per chunk and compute unit; tools/ubench/mfma_rate.hip: a 1 KB fragment read per MFMA doubles the MFMA's time, i.e. ~64 bytes per
// clock for this access) - and the workgroups spread from 65 to 92 us behind the operand image coming out of the L2s.  The way out straight
// from the accumulators (32-byte runs per row and instruction) costs nothing measurable against the parked 512-byte runs.
// Dense feature correspondence, DMA form (round 6)  out[n,h,w,i,j] = sum_c a[n,c,h,w] * b[n,c,i,j]  (gfx950).
//
//   reference: tensor_correlation() src/modules.py:283-284 = einsum("nchw,ncij->nhwij") on norm()'ed maps (:275-276); the full-resolution
//              callers are plot_dino_correspondence.py:39-58 / plot_pr_curves.py:108-121.  SURVEY.md 8(f) rank 3.
//
// What dense_stream.hip measured: beside MFMAs nothing is free - the conversion of B by the multiplying waves ADDS its VALU time and its
// loads' waits to every chunk (0.45 us on 1.05 us of MFMAs).  So this kernel's chunk loop holds nothing but LDS-DMA copies of B operands
// that are already split fp16 (dense_prep_kernel's image, as the row-block kernel reads it), fragment reads and MFMAs:
//   * decomposition, A prologue and the pinned one-fragment-set pipeline are dense_stream_kernel's (one workgroup of four 512-register waves
//     per (image, 128-pixel block of A), A fragments in 192 registers, wave w owns rows 32 w .. + 31 against the 128 columns of a B block);
//   * B chunks ([hi|lo][128][72] fp16, 36 KB) come through a THREE-stage LDS ring by global_load_lds, two copies in flight, counted waits;
//   * the MFMA operands are SWAPPED - D' = (B fragment) x (A fragment)^T - so that in the result layout a lane owns ONE output row and four
//     consecutive COLUMNS per register quad: the slab leaves straight from the accumulators in 16-byte stores (32 contiguous bytes per row
//     and instruction; the quads of a row complete its 128-byte lines within one block), no LDS park, no transposing pass - the LDS holds
//     the ring and nothing else, and nothing of the way out stands inside a chunk.
#include "corr_common.h"
#include "host_util.h"
#include <type_traits>

namespace stego {

struct DenseDmaParams {
    MapV a;                     // [B,C,H1,W1] channels-last (sc == 1), 16-byte aligned pixels
    const void* imgB;           // [B][nbB][NCH][hi|lo][128][72] fp16 (dense_prep_kernel)
    const float* rsB;           // [B][nbB*128] 1 / staging scale of a B row
    float* out;                 // [B][M][N]
    int B, C, M, N, W1, nbA, nbB, normalize;
    int dbg;
};

constexpr int DD_SIDE = 2 * TP * LDH * 2;           // bytes of one chunk image: hi[128][72] + lo[128][72] fp16
constexpr int DD_NST = 3;
constexpr int DD_LDS = DD_NST * DD_SIDE + TP * 4 + 1024; // + 1 / row scale of the A block, 1 KB for the 1 / staging scales of the B block being multiplied (512 bytes used)
static_assert(DD_SIDE % 4096 == 0, "whole 1 KB pieces per wave");

__device__ __forceinline__ void dd_dma_piece(const unsigned char* gsrc_lane, unsigned lds_addr)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc_lane), "s"(lds_addr) : "memory");
}

// grid = ceil(B / 8) * 8 * nbA, block = 256 (4 waves, 512 registers each); image n on XCD n % 8 (its workgroups stream the same B image)
template <int NCH>
__global__ void __launch_bounds__(NTHREADS) dense_dma_kernel(const DenseDmaParams prm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* ra_s = reinterpret_cast<float*>(smem + DD_NST * DD_SIDE);      // [128] 1 / row scale of the A block
    float* rb_s = ra_s + TP;                                              // [128] 1 / staging scale of the B block being multiplied
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = prm.C;
    const int x = blockIdx.x & 7, sl = blockIdx.x >> 3;
    const int n = (sl / prm.nbA) * 8 + x, mi = sl % prm.nbA;
    if (n >= prm.B) return;
    const int r = lane & 31, half = lane >> 5;
    const int q4 = lane >> 4, s16 = lane & 15;
    unsigned long long ts[8];
    ts[0] = __builtin_amdgcn_s_memrealtime();

    // ---- the A block -> MFMA fragments in registers (lane (r, half) holds channels 16 ks + 8 half .. + 7 of every 64-channel chunk of row
    // 32 w + r).  The block comes in the way the B chunks do - lane (q, s) reads 16 bytes of eight rows per chunk, four whole 256-byte runs
    // per wave instruction, all NCH chunks in flight at once -, gets its row statistics by shuffles inside the 16-lane groups, and passes
    // through the two LDS stages in the split-fp16 chunk layout, from which every lane takes its fragments with conflict-free 16-byte reads.
    // (The row-block kernel had each lane read its own row in 32-byte pieces, twice, behind a branch and a full wait per piece: ~25
    // serialized cold round trips + every line pulled into the L1 four times: 40 us of its 93.)
    f16x8 Ah[NCH][KC / 16], Al[NCH][KC / 16];
    {
        const float* aimg = prm.a.p + (long long)n * prm.a.sn;
        f32x4 ar[NCH][8];
        const float* arow[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int pc = min(mi * TP + 32 * wave + 4 * i + q4, prm.M - 1);
            const int hh = pc / prm.W1, ww = pc - hh * prm.W1;
            arow[i] = aimg + (long long)hh * prm.a.sh + (long long)ww * prm.a.sw;
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int ch = min(64 * c + 4 * s16, C - 4);
#pragma unroll
            for (int i = 0; i < 8; ++i) ar[c][i] = *reinterpret_cast<const f32x4*>(arow[i] + ch);
        }
        const float lastmask = (64 * (NCH - 1) + 4 * s16 >= C) ? 0.f : 1.f;       // channels beyond C (the last chunk only)
        float ainv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float ss = 0.f, mx = 0.f;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const f32x4 v = ar[c][i];
                const float s4 = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                const float m4 = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
                ss += c == NCH - 1 ? s4 * lastmask : s4;
                mx = fmaxf(mx, c == NCH - 1 ? m4 * lastmask : m4);
            }
#pragma unroll
            for (int m = 8; m >= 1; m >>= 1) { ss += __shfl_xor(ss, m, 64); mx = fmaxf(mx, __shfl_xor(mx, m, 64)); }
            float inv = prm.normalize ? 1.f / fmaxf(sqrtf(ss), 1e-10f) : 1.f;        // norm(), modules.py:276
            const float rs = mx * inv > 0.f ? __builtin_ldexpf(1.f, -__builtin_amdgcn_frexp_expf(mx * inv)) : 1.f;
            const int rl = 32 * wave + 4 * i + q4;
            ainv[i] = mi * TP + rl < prm.M ? inv * rs : 0.f;                       // (rows beyond M: the map's last pixel times zero)
            if (s16 == 0) ra_s[rl] = 1.f / rs;
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            unsigned char* stage = smem + (c & 1) * DD_SIDE;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float sc = c == NCH - 1 ? ainv[i] * lastmask : ainv[i];
                unsigned h0, l0, h1, l1;
                split_f16_pair(ar[c][i][0] * sc, ar[c][i][1] * sc, h0, l0);
                split_f16_pair(ar[c][i][2] * sc, ar[c][i][3] * sc, h1, l1);
                half_t* dh = reinterpret_cast<half_t*>(stage) + (32 * wave + 4 * i + q4) * LDH + 4 * s16;
                *reinterpret_cast<u32x2*>(dh) = u32x2{h0, h1};
                *reinterpret_cast<u32x2*>(dh + TP * LDH) = u32x2{l0, l1};
            }
            __syncthreads();                         // the chunk is whole (and everybody has read the stage's previous tenant: see above)
            const half_t* ap = reinterpret_cast<const half_t*>(stage) + (32 * wave + r) * LDH + 8 * half;
#pragma unroll
            for (int ks = 0; ks < KC / 16; ++ks) {
                Ah[c][ks] = *reinterpret_cast<const f16x8*>(ap + 16 * ks);
                Al[c][ks] = *reinterpret_cast<const f16x8*>(ap + TP * LDH + 16 * ks);
            }
        }
    }
    __syncthreads();                                 // the slots are free; ra_s
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the counted waits below start from zero
    ts[1] = __builtin_amdgcn_s_memrealtime();

    // ---- the B blocks of image n: chunk stream g = nj * NCH + c, nine 1 KB pieces per wave and chunk
    const unsigned char* Bimg = static_cast<const unsigned char*>(prm.imgB) + (size_t)n * prm.nbB * NCH * DD_SIDE;
    const unsigned smem_addr = (unsigned)(uintptr_t)((__attribute__((address_space(3))) const unsigned char*)smem);
    const int G = prm.nbB * NCH;
    auto issue = [&](int g) __attribute__((always_inline)) {
        const unsigned char* src = Bimg + (size_t)g * DD_SIDE + lane * 16;
        const unsigned dst = smem_addr + (g % DD_NST) * DD_SIDE;
#pragma unroll
        for (int i = 0; i < DD_SIDE / 4096; ++i) dd_dma_piece(src + (wave + 4 * i) * 1024, dst + (wave + 4 * i) * 1024);
    };
    issue(0);
    if (G > 1) issue(1);
    float* outn = prm.out + (size_t)n * prm.M * prm.N;
    const bool live = mi * TP + 32 * wave < prm.M;                // (a wave whose rows are all beyond M keeps the barriers and the copies)
    const bool row_ok = mi * TP + 32 * wave + r < prm.M;
    const bool v4 = (prm.N & 3) == 0;
    constexpr int LO = TP * LDH;
    // C/D layout of D' = Bfrag x Afrag^T: lane (r, half) holds output row 32 w + r; register e of group ni is column
    // 32 ni + (e & 3) + 8 (e >> 2) + 4 half: four consecutive columns per register quad
    const unsigned orow = 4u * ((unsigned)(mi * TP + 32 * wave + r) * (unsigned)prm.N + 4u * half);     // (an image's output is < 2^32 bytes, host-checked)
    int g = 0;
    const int nfull = prm.N / TP;                    // whole 128-pixel blocks of B
    const bool rows_full = mi * TP + 32 * wave + 32 <= prm.M;
    const bool counted = rows_full && v4;            // my slabs leave in exactly sixteen store instructions

    // the slab of block nj: scaled by 1 / both staging scales, 16-byte stores straight from the accumulators
    auto store_slab = [&](const int nj, const f32x16 (&acc)[4], const int nlive, const bool full) __attribute__((always_inline)) {
        const float sa = ra_s[32 * wave + r];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            if (ni < nlive) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int cl = 32 * ni + 8 * q + 4 * half;                     // my four columns of this quad, inside the block
                    const f32x4 sb = *reinterpret_cast<const f32x4*>(rb_s + cl);
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[ni][4 * q + e] * (sa * sb[e]);
                    float* o = reinterpret_cast<float*>(reinterpret_cast<char*>(outn) + (orow + 4u * (unsigned)(nj * TP + 32 * ni + 8 * q)));
                    if (full) {
                        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(o));
                    } else if (row_ok) {
                        const int col = nj * TP + cl;
                        if (v4) {
                            if (col < prm.N) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(o));
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (col + e < prm.N) __builtin_nontemporal_store(v[e], o + e);
                        }
                    }
                }
            }
        }
    };

    // One chunk of a WHOLE block: wait for my pieces of chunk g (counted: what I issued after them may still fly - the next chunk's nine
    // pieces and, in the first two chunks behind a slab, its sixteen stores), barrier, send chunk g + 2, multiply.
    auto fast_chunk = [&](auto firstc, const int nj, const int c, f32x16 (&acc)[4]) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(firstc)::value;          // the workgroup's first block: no stores in flight
        // (vector-memory operations of a wave complete in the order issued - the compiler's own wait insertion relies on it on gfx9.  Issued
        // behind chunk g's pieces: chunk g + 1's nine; in the first two chunks behind a slab its sixteen stores - when this wave stored all
        // of them unguarded: `counted` -; in a block's second chunk the 1 KB copy of the block's column scales)
        if (g + 1 >= G) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (c == 0) { if (!FIRST && counted && !(prm.dbg & 8)) asm volatile("s_waitcnt vmcnt(25)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); }
        else if (c == 1) { if (!FIRST && counted && !(prm.dbg & 8)) asm volatile("s_waitcnt vmcnt(26)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                // chunk g is whole, chunk g - 1 has been read by everybody
        asm volatile("" ::: "memory");
        // every wave copies the block's 128 column scales (1 KB from their address on: the words behind them are the next block's or padding) for
        // ITSELF - its own wait is all that stands between the copy and its reads at the block's end
        if (c == 0) dd_dma_piece(reinterpret_cast<const unsigned char*>(prm.rsB + ((size_t)n * prm.nbB + nj) * TP) + lane * 16, smem_addr + DD_NST * DD_SIDE + TP * 4);
        if (g + 2 < G) issue(g + 2);
        const half_t* bp = reinterpret_cast<const half_t*>(smem + (g % DD_NST) * DD_SIDE) + r * LDH + 8 * half;
        f16x8 bh[4], bl[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bh[ni] = *reinterpret_cast<const f16x8*>(bp + ni * 32 * LDH);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bl[ni] = *reinterpret_cast<const f16x8*>(bp + ni * 32 * LDH + LO);
#pragma unroll
        for (int ks = 0; ks < KC / 16; ++ks) {
            __builtin_amdgcn_sched_barrier(0);
            if (c == 0 && ks == 0) {
                const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[ni], Ah[c][ks], z, 0, 0, 0);
            } else {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[ni], Ah[c][ks], acc[ni], 0, 0, 0);
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[ni], Al[c][ks], acc[ni], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 1 < KC / 16) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) bh[ni] = *reinterpret_cast<const f16x8*>(bp + ni * 32 * LDH + 16 * (ks + 1));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[ni], Ah[c][ks], acc[ni], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 1 < KC / 16) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) bl[ni] = *reinterpret_cast<const f16x8*>(bp + ni * 32 * LDH + LO + 16 * (ks + 1));
            }
        }
        ++g;
    };
    // One chunk of the map's last, partial block: the nlive column groups that exist, one after the other (accumulators of its own)
    auto tail_chunk = [&](const int nj, const int c, f32x16 (&acc)[4], const int nlive) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (g + 1 < G) issue(g + 1);                 // (one copy in flight here: the waits above are full ones)
        if (c == 0 && tid < TP) rb_s[tid] = prm.rsB[((size_t)n * prm.nbB + nj) * TP + tid];
        if (c == 0) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[ni][e] = 0.f;
        }
        const half_t* bp = reinterpret_cast<const half_t*>(smem + (g % DD_NST) * DD_SIDE) + r * LDH + 8 * half;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            if (ni < nlive) {
#pragma unroll
                for (int ks = 0; ks < KC / 16; ++ks) {
                    const f16x8 bh = *reinterpret_cast<const f16x8*>(bp + ni * 32 * LDH + 16 * ks);
                    const f16x8 bl = *reinterpret_cast<const f16x8*>(bp + ni * 32 * LDH + LO + 16 * ks);
                    acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, Ah[c][ks], acc[ni], 0, 0, 0);
                    acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, Al[c][ks], acc[ni], 0, 0, 0);
                    acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, Ah[c][ks], acc[ni], 0, 0, 0);
                }
            }
        }
        ++g;
    };

    {
        f32x16 acc[4];
        auto full_block = [&](auto firstc, const int nj) __attribute__((always_inline)) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) fast_chunk(firstc, nj, c, acc);
            // the column scales have landed (they are older than the copy of the last chunk's top - if one was sent)
            if (g + 1 < G) asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (live && !(prm.dbg & 8)) {
                store_slab(nj, acc, 4, counted);
                if (!counted) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (a guarded slab: an unknown number of stores - the counts start over)
            }
        };
        if (nfull > 0) full_block(std::true_type{}, 0);
        ts[2] = __builtin_amdgcn_s_memrealtime();
        for (int nj = 1; nj < nfull; ++nj) full_block(std::false_type{}, nj);
    }
    ts[3] = ts[2];
    ts[4] = __builtin_amdgcn_s_memrealtime();
    if (nfull < prm.nbB) {                           // the map's last pixels
        f32x16 acc[4];
        const int nlive = live ? (prm.N - nfull * TP + 31) >> 5 : 0;
        // (the whole blocks' pipeline had chunk g + 1 in flight already: the tail's first wait drains it, then one copy at a time)
#pragma unroll
        for (int c = 0; c < NCH; ++c) tail_chunk(nfull, c, acc, nlive);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();                             // rb_s of the tail block (written at its first chunk) is whole for everybody
        if (nlive > 0) store_slab(nfull, acc, nlive, false);
    }
    ts[5] = __builtin_amdgcn_s_memrealtime();
    if (prm.dbg & 16) {                              // (tools: this workgroup's stamps over the first floats of its first output row)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ts[6] = __builtin_amdgcn_s_memrealtime();
        __syncthreads();
        if (tid == 0) {
            unsigned* o = reinterpret_cast<unsigned*>(outn + (size_t)(mi * TP) * prm.N);
            o[0] = (unsigned)ts[0];
#pragma unroll
            for (int k = 1; k < 7; ++k) o[k] = (unsigned)(ts[k] - ts[0]);
        }
    }
}

// the caller (launch_dense_corr) has checked: channels-last A map, 16-byte aligned pixels, C % 8 == 0, 64 < C <= 384; imgB / rsB are what
// dense_prep_kernel wrote for the B blocks
hipError_t launch_dense_dma(const MapV& a, const void* imgB, const float* rsB, int B, int C, int H1, int W1, int N, int normalize, float* out,
                            hipStream_t stream)
{
    DenseDmaParams prm{};
    prm.a = a; prm.imgB = imgB; prm.rsB = rsB; prm.out = out;
    prm.B = B; prm.C = C; prm.M = H1 * W1; prm.N = N; prm.W1 = W1;
    prm.nbA = (prm.M + TP - 1) / TP; prm.nbB = (N + TP - 1) / TP;
    prm.normalize = normalize;
    prm.dbg = (knob(KNOB_DEBUG) >> 22) & 31;
    const int NCH = (C + KC - 1) / KC;
    const dim3 grid((unsigned)(((B + 7) / 8) * 8 * prm.nbA));
#define STEGO_DD(N_)                                                                                                    \
    case N_: {                                                                                                          \
        hipError_t e_ = ensure_dynamic_lds(reinterpret_cast<const void*>(&dense_dma_kernel<N_>), DD_LDS);               \
        if (e_ != hipSuccess) return e_;                                                                                \
        hipLaunchKernelGGL(dense_dma_kernel<N_>, grid, dim3(NTHREADS), DD_LDS, stream, prm);                            \
        break;                                                                                                          \
    }
    switch (NCH) {
        STEGO_DD(2) STEGO_DD(3) STEGO_DD(4) STEGO_DD(5) STEGO_DD(6)
        default: return hipErrorInvalidValue;
    }
#undef STEGO_DD
    return hipGetLastError();
}

}  // namespace stego
