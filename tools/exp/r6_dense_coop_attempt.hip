// ROUND 6 ATTEMPT, NOT BUILT INTO THE LIBRARY (kept for the record; the product kernel is csrc/dense_stream.hip).
// Parity green (tests/test_dense_corr.py 23 / 23 incl. the odd shapes, with this kernel routed in).  Measured [32,784,784], C = 384: ONE launch,
// 95.3 us by rocprofv3 (97.9 by events) against dense_stats + dense_stream 9.9 + 88.2 (100.4 by events): the statistics launch is gone, the
// chunk stream holds no conversion any more - and a block of six chunks still takes 10.2 us (stamps: 23.8 -> 75.3 us for five blocks), exactly
// as in dense_stream_kernel.  The seven-fold conversion was NOT what a chunk costs beyond its MFMAs.  (A detail for whoever picks this up:
// the block's four column scales loaded by sc1 buffer loads inside a chunk and kept in registers until the park arrived wrong in every
// block but a workgroup's first - the same loads at the park, or staged through LDS as here, are right; not understood.)
// Dense feature correspondence, cooperative form (round 6)  out[n,h,w,i,j] = sum_c a[n,c,h,w] * b[n,c,i,j]  (gfx950).
//
//   reference: tensor_correlation() src/modules.py:283-284 = einsum("nchw,ncij->nhwij") on norm()'ed maps (:275-276); the full-resolution
//              callers are plot_dino_correspondence.py:39-58 / plot_pr_curves.py:108-121.  SURVEY.md 8(f) rank 3.
//
// dense_stream.hip converts B (fp32 -> split fp16) in every workgroup that multiplies it - seven times per image at 28 x 28 maps - and
// tools/ubench/mfma_beside.hip says what that costs: a SIMD has room for ~4 plain (2 packed) VALU instructions per MFMA, every one beyond
// takes matrix-core time, from whichever wave it comes.  Here the workgroups of an image share the work: ONE launch, the decomposition and
// the chunk pipeline of dense_coop_kernel, and in front of it
//   * workgroup (n, mi) converts the B blocks mi, mi + nbA, ... of its image ONCE - whole rows in registers, exact row statistics, the
//     split-fp16 chunk images staged in LDS and written out as contiguous 36 KB runs with write-through (sc1) stores, then a flag per
//     block (a 64-bit token of this launch: the workspace needs no clearing) -, reads its own A block into fragment registers, and
//     waits for the flags of its image's blocks (bounded: a block whose owner does not show up is converted here - identical bytes);
//   * the chunk stream copies the prepared images: nine 16-byte loads (sc1) and nine ds_write_b128 per lane and chunk, a piece reloaded
//     with the chunk after next as soon as it is in LDS - no VALU work in the loop but the addresses.
// No statistics launch, no operand prep launch.  The workgroups of an image must be able to run at the same time for the hand-off to cost
// nothing (they sit on one XCD, next to each other in dispatch order); when they cannot, the bounded wait turns into duplicate work.
#include "corr_common.h"
#include "host_util.h"
#include <type_traits>
#include <atomic>

namespace stego {

struct DenseCoopParams {
    MapV a, b;                  // [B,C,H1,W1], [B,C,H2,W2], channels-last (sc == 1), 16-byte aligned pixels
    float* out;                 // [B][M][N]
    unsigned char* imgB;        // [B][nbB][NCH][hi|lo][128][72] fp16: the split operands of B, written by the workgroups of the launch
    float* rsB;                 // [B][nbB*128]  1 / staging scale of a B row
    unsigned long long* flags;  // [B][nbB]  == token: the block's image and scales are in memory
    unsigned long long token;
    int timeout_ticks;          // of the 100 MHz clock
    int B, C, M, N, W1, W2, nbA, nbB, normalize;
    int dbg;                    // (tools: bit 16 - every workgroup leaves its phase stamps over the first floats of its first output row)
};

#ifndef DC2_ABL
#define DC2_ABL 0                                    // (tools/ubench/dense_stream_bench.hip: compile-time timing ablations of the whole-block chunk)
#endif
constexpr int DC2_STAGE = 2 * TP * LDH * 2;          // bytes of one LDS stage: hi[128][72] + lo[128][72] fp16
constexpr int DC2_PKS = 132;                         // floats per parked row (528 B: conflict-free 16-byte reads along a row)
constexpr int DC2_PARK = 32 * DC2_PKS * 4;            // one wave's parked 32 x 128 slab
constexpr int DC2_LDS = 2 * DC2_STAGE + 4 * DC2_PARK + 2 * TP * 4 + 256;    // + 1 / row scale of the A block, 1 / staging scales of the B block at hand, the missing-block votes

// grid = ceil(B / 8) * 8 * nbA, block = 256 (4 waves, 512 registers each); image n on XCD n % 8 (its workgroups stream the same B map)
template <int NCH>
__global__ void __launch_bounds__(NTHREADS) dense_coop_kernel(const DenseCoopParams prm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* ra_s = reinterpret_cast<float*>(smem + 2 * DC2_STAGE + 4 * DC2_PARK);        // [128] 1 / row scale of the A block
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = prm.C;
    const int x = blockIdx.x & 7, sl = blockIdx.x >> 3;
    const int n = (sl / prm.nbA) * 8 + x, mi = sl % prm.nbA;
    if (n >= prm.B) return;
    const int r = lane & 31, half = lane >> 5;
    unsigned long long ts[8];
    ts[0] = __builtin_amdgcn_s_memrealtime();

    const int q4 = lane >> 4, s16 = lane & 15;
    const int G = prm.nbB * NCH;
    const float lastmask = (64 * (NCH - 1) + 4 * s16 >= C) ? 0.f : 1.f;           // channels beyond C (the last chunk only)

    // ---- a 128-pixel block of a map, whole rows in registers: lane (q, s) reads 16 bytes of eight rows per chunk (four whole 256-byte runs
    // per wave instruction), every chunk in flight at once; pixels beyond the map read its last pixel (and are multiplied by zero later)
    auto load_block = [&](f32x4 (&R)[NCH][8], const MapV& m, int W, int P, int blk) __attribute__((always_inline)) {
        const float* img = m.p + (long long)n * m.sn;
        const float* row[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int pc = min(blk * TP + 32 * wave + 4 * i + q4, P - 1);
            const int hh = pc / W, ww = pc - hh * W;
            row[i] = img + (long long)hh * m.sh + (long long)ww * m.sw;
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int ch = min(64 * c + 4 * s16, C - 4);
#pragma unroll
            for (int i = 0; i < 8; ++i) R[c][i] = *reinterpret_cast<const f32x4*>(row[i] + ch);
        }
    };
    // its rows' statistics by shuffles inside the 16-lane groups: inv[i] = what row 32 w + 4 i + q is multiplied by in front of the split
    // (1 / ||row|| x a power of two that puts its largest magnitude into [0.5, 1); 0 beyond the map), 1 / that power to scale_out (s == 0)
    auto block_stats = [&](const f32x4 (&R)[NCH][8], int P, int blk, float (&inv)[8], float (&rscale)[8]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float ss = 0.f, mx = 0.f;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const f32x4 v = R[c][i];
                const float s4 = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                const float m4 = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
                ss += c == NCH - 1 ? s4 * lastmask : s4;
                mx = fmaxf(mx, c == NCH - 1 ? m4 * lastmask : m4);
            }
#pragma unroll
            for (int d = 8; d >= 1; d >>= 1) { ss += __shfl_xor(ss, d, 64); mx = fmaxf(mx, __shfl_xor(mx, d, 64)); }
            float nv = prm.normalize ? 1.f / fmaxf(sqrtf(ss), 1e-10f) : 1.f;         // norm(), modules.py:276
            const float rs = mx * nv > 0.f ? __builtin_ldexpf(1.f, -__builtin_amdgcn_frexp_expf(mx * nv)) : 1.f;
            inv[i] = blk * TP + 32 * wave + 4 * i + q4 < P ? nv * rs : 0.f;
            rscale[i] = 1.f / rs;
        }
    };
    // chunk c of the block, split fp16, into an LDS stage ([hi|lo][128][72])
    auto block_chunk_to_stage = [&](const f32x4 (&R)[NCH][8], const float (&inv)[8], int c, unsigned char* stage) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float sc = c == NCH - 1 ? inv[i] * lastmask : inv[i];
            unsigned h0, l0, h1, l1;
            split_f16_pair(R[c][i][0] * sc, R[c][i][1] * sc, h0, l0);
            split_f16_pair(R[c][i][2] * sc, R[c][i][3] * sc, h1, l1);
            half_t* dh = reinterpret_cast<half_t*>(stage) + (32 * wave + 4 * i + q4) * LDH + 4 * s16;
            *reinterpret_cast<u32x2*>(dh) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(dh + TP * LDH) = u32x2{l0, l1};
        }
    };

    // ---- the operand images of my image's B blocks: buffer descriptors over the image's part of the workspace (offsets: 32 bits)
    const size_t img_bytes = (size_t)prm.nbB * NCH * DC2_STAGE;
    const __amdgpu_buffer_rsrc_t img_rsrc = __builtin_amdgcn_make_buffer_rsrc(prm.imgB + (size_t)n * img_bytes, 0, (unsigned)img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_rsrc = __builtin_amdgcn_make_buffer_rsrc(prm.rsB + (size_t)n * prm.nbB * TP, 0, (unsigned)(prm.nbB * TP * 4), 0x00020000);
    unsigned long long* myflags = prm.flags + (size_t)n * prm.nbB;
    typedef unsigned int du32x4 __attribute__((ext_vector_type(4)));
    // block j of B: converted, written through, flagged.  (Everybody of the workgroup takes part; R holds the block's rows.)
    auto produce_block = [&](const f32x4 (&R)[NCH][8], int j) __attribute__((always_inline)) {
        float inv[8], rscale[8];
        block_stats(R, prm.N, j, inv, rscale);
        if (s16 == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, rscale[i]), rs_rsrc, 4u * (unsigned)(j * TP + 32 * wave + 4 * i + q4), 0, 16);
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            unsigned char* stage = smem + (c & 1) * DC2_STAGE;
            block_chunk_to_stage(R, inv, c, stage);
            __syncthreads();                         // the chunk is whole (and everybody has copied out the stage's previous tenant)
            const unsigned base = (unsigned)((j * NCH + c) * DC2_STAGE);
#pragma unroll
            for (int i = 0; i < DC2_STAGE / 4096; ++i) {
                const unsigned o = (unsigned)((wave + 4 * i) * 1024 + lane * 16);
                __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const du32x4*>(stage + o), img_rsrc, base + o, 0, 16);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                           // my write-through stores have landed
        __syncthreads();                                                            // everybody's
        if (tid == 0) __hip_atomic_store(myflags + j, prm.token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

    float* outn = prm.out + (size_t)n * prm.M * prm.N;
    float* park = reinterpret_cast<float*>(smem + 2 * DC2_STAGE) + wave * (32 * DC2_PKS);
    const bool live = mi * TP + 32 * wave < prm.M;                // (a wave whose rows are all beyond M converts and keeps the barriers)
    const bool rows_full = mi * TP + 32 * wave + 32 <= prm.M;
    const bool v4 = (prm.N & 3) == 0;
    constexpr int LO = TP * LDH;
    const float* prd = park + half * DC2_PKS + 4 * r;              // my 16 bytes of row pair k: prd + 2 k DC2_PKS
    const unsigned ordo = 4u * ((unsigned)(mi * TP + 32 * wave + half) * (unsigned)prm.N + 4u * r);      // byte offset of my piece of row pair 0 (an image's output is < 2^32 bytes, host-checked)
    const unsigned rowp = 8u * (unsigned)prm.N;                  // bytes between row pairs

    // row pairs [k0, k1) of the sixteen of the parked slab of block njp: row 2 k + half, columns 4 r .. 4 r + 3 (one 512-byte run per half-wave)
    auto store_rows = [&](int njp, int k0, int k1, bool full) __attribute__((always_inline)) {
        if (full) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (k < k0 || k >= k1) continue;
                const f32x4 v = *reinterpret_cast<const f32x4*>(prd + 2 * k * DC2_PKS);
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(reinterpret_cast<char*>(outn) + (ordo + k * rowp + 4u * (unsigned)(njp * TP))));
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (k < k0 || k >= k1) continue;
                const int row = mi * TP + 32 * wave + 2 * k + half, col = njp * TP + 4 * r;
                const f32x4 v = *reinterpret_cast<const f32x4*>(prd + 2 * k * DC2_PKS);
                if (row < prm.M) {
                    float* o = reinterpret_cast<float*>(reinterpret_cast<char*>(outn) + (ordo + k * rowp + 4u * (unsigned)(njp * TP)));
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
    };

    // ---- in front of the stream: my B block(s) out, my A block in
    f16x8 Ah[NCH][KC / 16], Al[NCH][KC / 16];
    {
        f32x4 rb0[NCH][8], ra0[NCH][8];
        const bool own = mi < prm.nbB;
        if (own) load_block(rb0, prm.b, prm.W2, prm.N, mi);
        load_block(ra0, prm.a, prm.W1, prm.M, mi);                                   // (both maps' cold reads in flight together)
        if (own) produce_block(rb0, mi);
        for (int j = mi + prm.nbA; j < prm.nbB; j += prm.nbA) {                      // (B larger than A: more blocks than workgroups)
            load_block(rb0, prm.b, prm.W2, prm.N, j);
            produce_block(rb0, j);
        }
        // the A block -> MFMA fragments in registers (lane (r, half) holds channels 16 ks + 8 half .. + 7 of every chunk of row 32 w + r),
        // through the two LDS stages in the chunk layout: conflict-free 16-byte fragment reads
        float ainv[8], ars[8];
        block_stats(ra0, prm.M, mi, ainv, ars);
        if (s16 == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) ra_s[32 * wave + 4 * i + q4] = ars[i];
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            unsigned char* stage = smem + (c & 1) * DC2_STAGE;
            block_chunk_to_stage(ra0, ainv, c, stage);
            __syncthreads();
            const half_t* ap = reinterpret_cast<const half_t*>(stage) + (32 * wave + r) * LDH + 8 * half;
#pragma unroll
            for (int ks = 0; ks < KC / 16; ++ks) {
                Ah[c][ks] = *reinterpret_cast<const f16x8*>(ap + 16 * ks);
                Al[c][ks] = *reinterpret_cast<const f16x8*>(ap + TP * LDH + 16 * ks);
            }
        }
        __syncthreads();                             // the stages are free; ra_s
        ts[1] = __builtin_amdgcn_s_memrealtime();
        // ---- my image's blocks are out?  One thread per block polls its flag (bounded); a block whose owner did not show up is made here
        int* miss = reinterpret_cast<int*>(ra_s + 2 * TP);                           // [<= 64]
        if (tid < 64) {
            int m_ = 0;
            if (tid < prm.nbB) {
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                for (;;) {
                    if (__hip_atomic_load(myflags + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == prm.token) break;
                    if ((long long)(__builtin_amdgcn_s_memrealtime() - t0) > prm.timeout_ticks) { m_ = 1; break; }
                    __builtin_amdgcn_s_sleep(4);
                }
            }
            miss[tid] = m_;
        }
        __syncthreads();
        for (int j = 0; j < prm.nbB; ++j) {
            if (miss[j]) {                           // (workgroup-uniform; never in normal operation)
                load_block(rb0, prm.b, prm.W2, prm.N, j);
                produce_block(rb0, j);
            }
        }
    }
    // ---- the chunk stream g = nj * NCH + c of prepared images: nine 1 KB pieces per wave and chunk, lane l holds 16 bytes of each
    du32x4 pc[DC2_STAGE / 4096];
    auto load_piece = [&](int g2, int i) __attribute__((always_inline)) {
        pc[i] = __builtin_amdgcn_raw_buffer_load_b128(img_rsrc, (unsigned)((wave + 4 * i) * 1024 + lane * 16), (unsigned)(min(g2, G - 1) * DC2_STAGE), 16);
    };
    auto put_piece = [&](unsigned char* stage, int i) __attribute__((always_inline)) {
        *reinterpret_cast<du32x4*>(stage + (wave + 4 * i) * 1024 + lane * 16) = pc[i];
    };
#pragma unroll
    for (int i = 0; i < DC2_STAGE / 4096; ++i) load_piece(0, i);
#pragma unroll
    for (int i = 0; i < DC2_STAGE / 4096; ++i) { put_piece(smem, i); load_piece(1, i); }
    __syncthreads();

    int pending = -1;                                // block whose slab is parked and not yet stored
    bool pending_full = false;
    float rbv[4];                                    // 1 / staging scale of my four columns of the block being multiplied
    int g = 0;
    // One chunk of a WHOLE block (all four 32-column groups exist: every block in front of the map's last, partial one).  The body is one
    // basic block - no run-time condition inside - so that the scheduler can put the conversion's VALU work, the LDS traffic and the
    // previous slab's stores between the MFMAs (a wave issues in order: what stands behind the last MFMA of a k-step waits for all of
    // them): chunk g + 2 is ALWAYS loaded and chunk g + 1 ALWAYS converted (behind the stream's end: clamped addresses, a stage nobody
    // reads), whether the previous slab is stored (every block but the first) and whether my 32 rows all exist are compile-time flags.
    auto fast_chunk = [&](auto firstc, auto fullc, const int nj, const int c, f32x16 (&acc)[4]) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(firstc)::value, ROWS_FULL = decltype(fullc)::value;
        unsigned char* Sg = smem + (g & 1) * DC2_STAGE;
        unsigned char* Sn = smem + ((g + 1) & 1) * DC2_STAGE;
        if (!(FIRST && c == 0)) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();            // stage g & 1 is complete, the other one has been read
            asm volatile("" ::: "memory");
        }
        // the block's column scales (1 / staging scale of its B rows): loaded at the top of its last chunk, in LDS at that chunk's end
        float rbl = 0.f;
        if (c == NCH - 1 && tid < TP) rbl = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_rsrc, 4u * (unsigned)(nj * TP + tid), 0, 16));
        const half_t* bp = reinterpret_cast<const half_t*>(Sg) + r * LDH + 8 * half;
        // ONE set of fragments (32 registers), reloaded as it dies: per k-step the eight MFMAs that need the hi fragments of B run first, the
        // next k-step's hi fragments are read into their registers under the four MFMAs that need the lo fragments, and those are re-read
        // right behind them.  sched_barrier pins that order (left alone, the scheduler - short of registers - sinks every read to just in
        // front of its MFMA: an LDS round trip exposed per MFMA, 1.9 us per chunk; two full sets in flight made it spill inside the loop).
        f16x8 bh[4], bl[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bh[ni] = *reinterpret_cast<const f16x8*>(bp + ni * 32 * LDH);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bl[ni] = *reinterpret_cast<const f16x8*>(bp + ni * 32 * LDH + LO);
#pragma unroll
        for (int ks = 0; ks < KC / 16; ++ks) {
            __builtin_amdgcn_sched_barrier(0);
            if (!(DC2_ABL & 4)) {
                if (c == 0 && ks == 0) {
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[c][ks], bh[ni], f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                } else {
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[c][ks], bh[ni], acc[ni], 0, 0, 0);
                }
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[c][ks], bh[ni], acc[ni], 0, 0, 0);
            }
            put_piece(Sn, 2 * ks); load_piece(g + 2, 2 * ks);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 1 < KC / 16 && !(DC2_ABL & 1)) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) bh[ni] = *reinterpret_cast<const f16x8*>(bp + ni * 32 * LDH + 16 * (ks + 1));
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!(DC2_ABL & 4)) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[c][ks], bl[ni], acc[ni], 0, 0, 0);
            }
            put_piece(Sn, 2 * ks + 1); load_piece(g + 2, 2 * ks + 1);
            if (ks == KC / 16 - 1) { put_piece(Sn, 8); load_piece(g + 2, 8); }
            if (c == 0 && !FIRST && !(DC2_ABL & 8)) {
                if constexpr (ROWS_FULL) store_rows(nj - 1, 4 * ks, 4 * ks + 4, true);
                else if (pending >= 0) store_rows(nj - 1, 4 * ks, 4 * ks + 4, false);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 1 < KC / 16 && !(DC2_ABL & 1)) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) bl[ni] = *reinterpret_cast<const f16x8*>(bp + ni * 32 * LDH + LO + 16 * (ks + 1));
            }
        }
        if (c == NCH - 1 && tid < TP) (ra_s + TP)[tid] = rbl;
        if (c == 0) pending = -1;
        ++g;
    };
    // One chunk of the map's last, partial block: the nlive groups that exist, one after the other (accumulators of its own: sharing them
    // with the whole blocks' form made the compiler copy all 64 registers between the two allocations in every chunk).
    auto tail_chunk = [&](const int nj, const int c, f32x16 (&acc)[4], const int nlive) __attribute__((always_inline)) {
        unsigned char* Sg = smem + (g & 1) * DC2_STAGE;
        unsigned char* Sn = smem + ((g + 1) & 1) * DC2_STAGE;
        if (g > 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        // (only the rows of the block that exist are converted: the waves behind them have nothing to do but the barriers)
        const bool conv = g + 1 < G;
        if (c == 0) {
            if (tid < TP) (ra_s + TP)[tid] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_rsrc, 4u * (unsigned)(nj * TP + tid), 0, 16));       // (read behind the later chunks' barriers)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[ni][e] = 0.f;
        }
        const half_t* bp = reinterpret_cast<const half_t*>(Sg) + r * LDH + 8 * half;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            if (ni < nlive) {
#pragma unroll
                for (int ks = 0; ks < KC / 16; ++ks) {
                    const f16x8 bh = *reinterpret_cast<const f16x8*>(bp + ni * 32 * LDH + 16 * ks);
                    const f16x8 bl = *reinterpret_cast<const f16x8*>(bp + ni * 32 * LDH + LO + 16 * ks);
                    acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[c][ks], bh, acc[ni], 0, 0, 0);
                    acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[c][ks], bl, acc[ni], 0, 0, 0);
                    acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[c][ks], bh, acc[ni], 0, 0, 0);
                }
            }
        }
        if (conv) {
#pragma unroll
            for (int i = 0; i < DC2_STAGE / 4096; ++i) { put_piece(Sn, i); load_piece(g + 2, i); }
        }
        if (c == 0 && pending >= 0) store_rows(pending, 0, 16, pending_full);
        if (c == 0) pending = -1;
        ++g;
    };
    // park my 32 x 128 slab.  C/D layout: col = lane & 31 (+ 32 ni), row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
    auto park_slab = [&](const int nj, const f32x16 (&acc)[4], const int nlive) __attribute__((always_inline)) {
        (void)rbv;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            if (ni < nlive) {
                const float sb = (ra_s + TP)[32 * ni + r];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int rl = (e & 3) + 8 * (e >> 2) + 4 * half;
                    park[rl * DC2_PKS + 32 * ni + r] = acc[ni][e] * (ra_s[32 * wave + rl] * sb);
                }
            }
        }
        pending = nj;
        pending_full = rows_full && v4 && nj * TP + TP <= prm.N;
    };

    const int nfull = prm.N / TP;                    // whole 128-pixel blocks of B
    // the whole blocks, for a wave whose 32 rows all exist (unguarded 16-byte stores; needs N % 4 == 0) or not
    auto whole_blocks = [&](auto fullc) __attribute__((always_inline)) {
        constexpr bool ROWS_FULL = decltype(fullc)::value;
        f32x16 acc[4];
        auto full_block = [&](auto firstc, const int nj) __attribute__((always_inline)) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) fast_chunk(firstc, fullc, nj, c, acc);
            __syncthreads();                         // the block's column scales are in LDS for everybody
            if (ROWS_FULL || live) park_slab(nj, acc, 4);         // (a wave whose rows are all beyond M multiplied zeros)
        };
        if (nfull > 0) full_block(std::true_type{}, 0);
        ts[2] = __builtin_amdgcn_s_memrealtime();
        for (int nj = 1; nj < nfull; ++nj) full_block(std::false_type{}, nj);
    };
    if (rows_full && v4) whole_blocks(std::true_type{}); else whole_blocks(std::false_type{});
    ts[3] = ts[2];
    ts[4] = __builtin_amdgcn_s_memrealtime();
    if (nfull < prm.nbB) {                           // the map's last pixels
        f32x16 acc[4];
        const int nlive = live ? (prm.N - nfull * TP + 31) >> 5 : 0;
#pragma unroll
        for (int c = 0; c < NCH; ++c) tail_chunk(nfull, c, acc, nlive);
        __syncthreads();
        if (nlive > 0) park_slab(nfull, acc, nlive);
    }
    ts[5] = __builtin_amdgcn_s_memrealtime();
    if (pending >= 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        store_rows(pending, 0, 16, pending_full);
    }
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

size_t dense_coop_workspace_bytes(int B, int C, int N)
{
    const size_t nbB = (N + TP - 1) / TP, NCH = (C + KC - 1) / KC;
    return (size_t)B * nbB * NCH * DC2_STAGE + (size_t)B * nbB * TP * sizeof(float) + (size_t)B * nbB * sizeof(unsigned long long) + 1024;
}

// the caller (launch_dense_corr) has checked: channels-last maps, 16-byte aligned pixels, C % 8 == 0, 64 < C <= 384, a B image of operands
// below 2^31 bytes
hipError_t launch_dense_coop(const MapV& a, const MapV& b, int B, int C, int H1, int W1, int H2, int W2, int normalize, float* out, void* ws,
                             hipStream_t stream)
{
    static std::atomic<unsigned long long> launches{0x243F6A8885A308D3ull};          // (a token per launch: stale flags of an earlier one never match)
    DenseCoopParams prm{};
    prm.a = a; prm.b = b; prm.out = out;
    prm.B = B; prm.C = C; prm.M = H1 * W1; prm.N = H2 * W2; prm.W1 = W1; prm.W2 = W2;
    prm.nbA = (prm.M + TP - 1) / TP; prm.nbB = (prm.N + TP - 1) / TP;
    prm.normalize = normalize;
    prm.dbg = (knob(KNOB_DEBUG) >> 22) & 31;
    const int NCH = (C + KC - 1) / KC;
    unsigned char* w = static_cast<unsigned char*>(ws);
    w += (256 - (reinterpret_cast<uintptr_t>(w) & 255)) & 255;
    prm.imgB = w;
    prm.rsB = reinterpret_cast<float*>(w + (size_t)B * prm.nbB * NCH * DC2_STAGE);
    prm.flags = reinterpret_cast<unsigned long long*>(prm.rsB + (size_t)B * prm.nbB * TP);
    prm.token = launches.fetch_add(0x9E3779B97F4A7C15ull) | 1ull;
    prm.timeout_ticks = 20000;                       // 200 us
    const dim3 grid((unsigned)(((B + 7) / 8) * 8 * prm.nbA));
#define STEGO_DC(N_)                                                                                                    \
    case N_: {                                                                                                          \
        hipError_t e_ = ensure_dynamic_lds(reinterpret_cast<const void*>(&dense_coop_kernel<N_>), DC2_LDS);             \
        if (e_ != hipSuccess) return e_;                                                                                \
        hipLaunchKernelGGL(dense_coop_kernel<N_>, grid, dim3(NTHREADS), DC2_LDS, stream, prm);                          \
        break;                                                                                                          \
    }
    switch (NCH) {
#ifndef DC2_ONLY6
        STEGO_DC(2) STEGO_DC(3) STEGO_DC(4) STEGO_DC(5)
#endif
        STEGO_DC(6)
        default: return hipErrorInvalidValue;
    }
#undef STEGO_DC
    return hipGetLastError();
}

}  // namespace stego
