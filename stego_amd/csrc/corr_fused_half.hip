// Column-half launch of the fused forward for SMALL batches (round 6): 16 B <= compute units, i.e. B <= 16 on an MI355X - the
// reference's shipped batch size (src/configs/train_config.yml:11) and BASELINE config 4 at B = 16.
//
// Reference path: src/modules.py:349-398 (forward), :325-347 (helper), :275-295 - the same arithmetic as corr_fused.hip.
//
// Why.  With one workgroup per 128 x 128 tile a batch of 16 puts 112 tiles on 256 compute units, and the launch takes as long as at
// B = 32 (40.9 vs 50 us): its length is one compute unit pulling the 880 KB of a whole B side through its own memory pipeline -
// ~55 GB/s from a warm L2, ~37 GB/s cold, whatever the rest of the chip does (tools/ubench/fused_skeleton.hip, profiles/r06_skeleton_*:
// the traffic-only skeleton of the full-tile layout needs 30.8 us at B = 16, the column-half layout 26.6).  Here a work ITEM is a column
// half of a tile - all 128 anchor rows x 64 B-side points - so a B side is gathered by TWO compute units, 32 channels x 64 points per
// stage each, and every unit of the chip has an item (B = 16: 224 items + 32 workgroups that only sample anchors).  At B = 32 the same
// split would put two items on every unit - the skeleton measures it slower than whole tiles (53.8 vs 45.4 us: the anchor operand is
// streamed twice, the per-unit bytes go up) - so the launcher takes this kernel only while every item gets a unit of its own.
//
// What differs from corr_fused_kernel (everything else - placement by the XCD of the source image, phase 1, the ring, the code chunks,
// the split-fp16 products, the old_mean rendezvous, the last workgroup's tail - is the same code, included from corr_fused.hip):
//   * the grid: n_anchor_wg workgroups in FRONT (dispatched first) that only sample anchors, then 2 x tiles items; the samplers of
//     XCD x are its anchor workgroups and the workgroups that hold a self-correlation half (no gather stream; the placement puts them
//     on the first slots of the XCD), all twelve waves, as many passes as the share needs;
//   * a workgroup multiplies 128 x 64: wave (wr, wc) of the MFMA team owns rows 64 wr .. + 63 x columns 32 wc .. + 31; the gather
//     team's eight waves hold ONE point per lane group (64 points), two register sets as before;
//   * fd.mean([3, 4]) (modules.py:332) needs whole rows: each half publishes its 121 partial row sums as {tag, value} granules, reads
//     its partner's (which it zeroes again: one reader each) and both add them in the order (half 0) + (half 1) - same bits on both
//     sides.  cd leaves before that wait, w / the sums after it, the negative loss after the pair-set's old_mean as before;
//   * outputs are row segments of 64 / 57 floats: the sweep walks the 16-byte groups of the flat [P][P] layout that touch its columns,
//     whole groups as vectors, the two straddling groups of a row by element.
// Bounded waits: the anchor wait falls back on sampling the anchor itself and the old_mean rendezvous on the last workgroup's repair,
// as in the full-tile kernel; a PARTNER that does not show up for 20 ms (the device is not ours: the launcher never takes this kernel
// with STEGO_FLAG_SHARED_DEVICE) poisons the row means - NaN outputs and scalars, counted in the event words - instead of hanging.
#define STEGO_FUSED_PART 3
#include "corr_fused.hip"

namespace stego {

constexpr int HB = 64;                           // B-side points of a work item

// ------------------------------------------------------------------------------------------ MFMA stages, 128 x 64
// wave (wr, wc): A rows 64 wr + {0, 32} + r, B rows brow + r (brow = 32 wc in the gathered B side, q0 + 32 wc in the A side of a
// self-correlation item)
template <int MB>
__device__ __forceinline__ void mma_half_h(const unsigned char* __restrict__ As, const unsigned char* __restrict__ Bs, int brow,
                                           f32x16 (&acc)[MB], int lane, int wr)
{
    const int r = lane & 31, half = lane >> 5;
    const int ra0 = 32 * MB * wr + r, rb = brow + r;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int u = 2 * ks + half;
        f16x8 ah[MB], al[MB];
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            ah[i] = *reinterpret_cast<const f16x8*>(As + swz_h(ra0 + 32 * i, u));
            al[i] = *reinterpret_cast<const f16x8*>(As + 8192 + swz_h(ra0 + 32 * i, u));
        }
        const f16x8 bh = *reinterpret_cast<const f16x8*>(Bs + swz_h(rb, u)), bl = *reinterpret_cast<const f16x8*>(Bs + 8192 + swz_h(rb, u));
#pragma unroll
        for (int i = 0; i < MB; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MB; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MB; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh, acc[i], 0, 0, 0);
    }
}

// One k-step (16 channels) of a feature stage in format H as two halves - the six fragment reads, the six MFMAs - so that the
// kernel can put OTHER work between them: the LDS pipe needs as long for a stage's twelve 1 KB fragment reads as the matrix core for its
// twelve MFMAs (tools/ubench/mfma_rate.hip: 13.4 ns per MFMA from registers, 27-29 ns with a read per MFMA, one wave per SIMD), and a wave
// that reads, waits and multiplies in turn pays both.
template <int MB> struct HFrag { f16x8 ah[MB], al[MB], bh, bl; };
template <int MB>
__device__ __forceinline__ void half_frag_read(HFrag<MB>& f, const unsigned char* __restrict__ As, const unsigned char* __restrict__ Bs, int brow,
                                               int blo, int ks, int lane, int wr)
{
    const int r = lane & 31, half = lane >> 5;
    const int ra0 = 32 * MB * wr + r, rb = brow + r;
    const int u = 2 * ks + half;
#pragma unroll
    for (int i = 0; i < MB; ++i) f.al[i] = *reinterpret_cast<const f16x8*>(As + 8192 + swz_h(ra0 + 32 * i, u));
    f.bh = *reinterpret_cast<const f16x8*>(Bs + swz_h(rb, u));
#pragma unroll
    for (int i = 0; i < MB; ++i) f.ah[i] = *reinterpret_cast<const f16x8*>(As + swz_h(ra0 + 32 * i, u));
    f.bl = *reinterpret_cast<const f16x8*>(Bs + blo + swz_h(rb, u));
}
template <int MB>
__device__ __forceinline__ void half_frag_mma(const HFrag<MB>& f, f32x16 (&acc)[MB])
{
#pragma unroll
    for (int i = 0; i < MB; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[i], f.bh, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < MB; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bl, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < MB; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bh, acc[i], 0, 0, 0);
}

template <int MB>
__device__ __forceinline__ void mma_half_f(const unsigned char* __restrict__ As, const unsigned char* __restrict__ Bs, int brow, int kper,
                                           f32x16 (&acc)[MB], int lane, int wr)
{
    const int r = lane & 31, half = lane >> 5;
    const int ra0 = 32 * MB * wr + r, rb = brow + r;
    for (int kk = 0; kk < kper; kk += 8) {
        const int u = (kk >> 2) + half;
        f32x4 a[MB];
#pragma unroll
        for (int i = 0; i < MB; ++i) a[i] = *reinterpret_cast<const f32x4*>(As + swz_f(ra0 + 32 * i, u));
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(Bs + swz_f(rb, u));
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < MB; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][j], b0[j], acc[i], 0, 0, 0);
    }
}

// The result tiles are parked in the FLAT layout of the outputs, T[a + row * P + col] with my columns only (the partner's stay whatever
// the ring held: the sweep masks them), so that the sweep reads 16-byte LDS vectors that are the 16-byte global vectors (`a`: see
// park_flat in corr_tile.h).  C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
// Branch-free: padding rows / columns go to a per-lane dummy word behind the tile.
template <int MB>
__device__ __forceinline__ void park_half(const f32x16 (&acc)[MB], float* __restrict__ T, int P, int q0, const float* colscale, int lane, int wr,
                                          int wc)
{
    const int dummy = TP * LDT - 72 + lane;          // (T may be shifted by up to 3 floats)
    const int cl = 32 * wc + (lane & 31), col = q0 + cl;
    const float sc = colscale[cl];
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * MB * wr + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            T[(row < P && col < P) ? row * P + col : dummy] = acc[mi][r] * sc;
        }
}

// ------------------------------------------------------------------------------------------ the last workgroup of the launch
// last_workgroup_tail for 2 B items per pair-set (item = 2 * tile + half): same sums in the same tree order, the repair of an item
// touches its own columns only.
template <int NW>
__device__ __forceinline__ void last_workgroup_tail_half(const FusedParams& prm, float* Tfd, float* timed_out, int tid, int n_items,
                                                         unsigned long long* ts = nullptr)
{
    if (ts && tid == 0) ts[9] = __builtin_amdgcn_s_memrealtime();
    const int B = prm.B, PB = 2 * B, P = prm.P, P2 = P * P;
    const float cmin = prm.cmin, cmax = prm.cmax;
    unsigned long long* gst = prm.gran + n_items;                  // [n_items][3]: sum lp, sum clamp, old_mean applied
    float* sst = Tfd;                            // [n_items][4]
    float* som = sst + n_items * 4;              // [n_sets] old_mean per pair-set, [n_sets] sum of its loss
    float* sums3 = som + 2 * prm.n_sets;         // [n_sets][3]
    bool mine = false;
    {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (int i = tid; i < n_items * 4; i += 64 * NW) {
            const int t = i >> 2, k = i & 3;
            const unsigned long long* src = k == 0 ? prm.gran + t : gst + (size_t)t * 3 + (k - 1);
            unsigned long long x;
            for (;;) {
                x = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((x >> 32) == 1ull) break;
                if ((long long)(__builtin_amdgcn_s_memrealtime() - t0) > prm.timeout_ticks) { timed_out[0] = 1.f; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            const float v = __builtin_bit_cast(float, (unsigned)x);
            sst[i] = v;
            mine |= k == 3 && t >= 2 * PB && v == 0.f;
        }
    }
    const bool repair = __syncthreads_or(mine) && prm.pointwise;
    for (int idx = tid >> 6; idx < 3 * prm.n_sets; idx += NW) {
        const int ps = idx / 3, k = idx - 3 * ps, lane = tid & 63;
        const float* st = sst + (size_t)ps * PB * 4 + k;
        float acc = 0.f;
        for (int i0 = 0; i0 < PB; i0 += 64) acc += wave_tree_sum(i0 + lane < PB ? st[(i0 + lane) * 4] : 0.f);
        if (lane == 0) sums3[idx] = acc;
    }
    if (tid >= 64) {
        constexpr int NZ = 64 * NW - 64;
        for (int i = tid - 64; i < B; i += NZ)
            __hip_atomic_store(prm.anchor_cnt + (size_t)i * ANCHOR_CNT_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int i = tid - 64; i < n_items * 4; i += NZ)
            __hip_atomic_store(prm.gran + i, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (tid < 64) {
        const float inv_cnt = 1.f / ((float)B * (float)P2);
        for (int ps = tid; ps < prm.n_sets; ps += 64) {
            const float fsum = sums3[3 * ps], lsum = sums3[3 * ps + 1], csum = sums3[3 * ps + 2];
            const float omp = prm.pointwise ? fsum * inv_cnt : 0.f;
            som[ps] = fsum * inv_cnt;
            som[prm.n_sets + ps] = lsum - omp * csum;
            const float poison = timed_out[0] != 0.f ? __builtin_nanf("") : 0.f;
            if (prm.saved_mean) prm.saved_mean[ps] = omp + poison;
            if (ps < 2) prm.loss_means[ps] = (lsum - omp * csum) * inv_cnt + poison;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (tid == 0) {
            float nsum = 0.f;
            for (int pp = 2; pp < prm.n_sets; ++pp) nsum += som[prm.n_sets + pp];
            prm.loss_means[2] = (prm.n_neg > 0 ? nsum * inv_cnt / (float)prm.n_neg : 0.f) + (timed_out[0] != 0.f ? __builtin_nanf("") : 0.f);
        }
    }
    if (repair) {
        __syncthreads();
        for (int t = 2 * PB; t < n_items; ++t) {
            if (sst[t * 4 + 3] != 0.f) continue;
            const int tile = t >> 1, c0 = (t & 1) * HB, c1 = min(c0 + HB, P);
            const float omp = som[tile / B];
            float* lossr = prm.neg_loss + (size_t)(tile - 2 * B) * P2;
            const float* cdr = prm.neg_cd + (size_t)(tile - 2 * B) * P2;
            for (int e = tid; e < P2; e += 64 * NW) {
                const int col = e % P;
                if (col < c0 || col >= c1) continue;
                const float cdv = __hip_atomic_load(cdr + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float lpv = __hip_atomic_load(lossr + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float cl = fminf(fmaxf(cdv, cmin), cmax);
                lossr[e] = __builtin_fmaf(-omp, cl, lpv);
            }
        }
    }
    if (tid == 64) __hip_atomic_store(prm.done_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ts && tid == 0) ts[10] = __builtin_amdgcn_s_memrealtime();
}

// ------------------------------------------------------------------------------------------ the kernel
// NW waves per workgroup: the last eight gather, the first MW = NW - 8 multiply.  NW = 12 is corr_fused_kernel's split (four MFMA waves of
// 64 x 32).  NW = 16 (C = 384): EIGHT MFMA waves of 32 x 32, two per SIMD - one wave's fragment reads run under the other's MFMAs (the
// LDS pipe needs as long for a stage's fragments as the matrix core for its MFMAs, tools/ubench/mfma_rate.hip), each issues two of a
// stage's sixteen anchor copies instead of four - and phase 1 puts ONE row pair on each of sixteen waves instead of two pairs on eight
// (its arithmetic, ~600 instructions per row pair, is what the anchors wait for).  128 registers per lane then: fine at C = 384, not for
// the 96 tap registers of a C = 768 row (the launcher keeps NW = 12 there).
template <int PREC, int NJ, int NKCT, int NW>
__global__ void __launch_bounds__(64 * NW) corr_fused_half_kernel(const FusedParams prm)
{
    constexpr int MW = NW - 8;                   // MFMA waves
    constexpr int MB = 8 / MW;                   // 32-row blocks per MFMA wave: wave (wr, wc) owns rows 32 MB wr .. x columns 32 wc ..
    constexpr int PPW = 16 / MW;                 // pieces of a 16 KB anchor stage per MFMA wave
    constexpr int NTHR = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* rowmean = reinterpret_cast<float*>(smem + RD_ROWMEAN);
    float* red = reinterpret_cast<float*>(smem + RD_RED);
    float* csc = reinterpret_cast<float*>(smem + RD_CSC);
    float* cscc = reinterpret_cast<float*>(smem + RD_CSCC);
    int4* tapof = reinterpret_cast<int4*>(smem + RD_TAPOF);
    int4* tapoc = reinterpret_cast<int4*>(smem + RD_TAPOC);
    float4* tapw = reinterpret_cast<float4*>(smem + RD_TAPW);
    unsigned char* ring = smem + RD_RING;
    float* Tfd = reinterpret_cast<float*>(smem + RD_RING);   // epilogue aliases of the ring
    float* Tcd = Tfd + TP * LDT;
    float* rsum = reinterpret_cast<float*>(smem + RD_TAPOF);      // [128] my partial row sums of fd (the tap tables are dead by then)
    typedef P1Layout<NJ, PREC> LY;
    typedef P1Layout<NJ, PREC, true> LYL;
    constexpr int NCH2 = LY::NCH2;
    constexpr int NKC = NKCT;
    constexpr int NT = NKC + NCH2;
    constexpr bool CH = PREC == PREC_F16X3;      // code operands in format H (see p1_sample_rows): the code stages are feature-like stages

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool mfma_team = wave8 < MW;
    const int wave = mfma_team ? wave8 : wave8 - MW;      // index inside the team
    const int gt = mfma_team ? tid : tid - 64 * MW;
    const int wr = wave >> 1, wc = wave & 1;
    const int B = prm.B, P = prm.P;
    const int kper = prm.kper;
    const int me = blockIdx.x;
    const int NA = prm.n_anchor_wg;
    const int n_tiles = prm.n_sets * B, n_items = 2 * n_tiles;
    const bool anchor_wg = me < NA;

    unsigned long long* ts = reinterpret_cast<unsigned long long*>(prm.stats + (size_t)n_tiles * 4 + 256) + (size_t)me * 16;
    const bool stamp_on = (prm.debug & 256) && tid == 0;
    if (stamp_on) ts[0] = __builtin_amdgcn_s_memrealtime();

    const __amdgpu_buffer_rsrc_t fs_rsrc = __builtin_amdgcn_make_buffer_rsrc(prm.fs, 0, prm.fs_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t csf_rsrc = __builtin_amdgcn_make_buffer_rsrc(prm.csf, 0, prm.csf_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t cs_rsrc = __builtin_amdgcn_make_buffer_rsrc(prm.cs, 0, prm.cs_bytes, 0x00020000);

    int* tile_slot = reinterpret_cast<int*>(red + 56);
    unsigned* team_cnt = reinterpret_cast<unsigned*>(red + 57);
    float* fin = red + 48;                       // [0] 1 = I am the last workgroup, [1] a hand-off word timed out in the tail
    if (tid == 0) { tile_slot[0] = -1; team_cnt[0] = 0u; fin[1] = 0.f; }
    __syncthreads();

    if (!anchor_wg && wave8 > MW) {              // coords2 into this CU's L1 while the item is worked out (as corr_fused_kernel)
        const int idx = 32 * ((wave8 - MW - 1) * 64 + lane);
        if (idx < B * P * 2) { const float x = prm.coords2[idx]; asm volatile("" :: "v"(x)); }
    }

    // ---- phase 1: the samplers of XCD x = its anchor workgroups + the workgroups on the first 2 nb tile slots (they hold the
    // self-correlation halves of the XCD's nb anchors: first in item order, preferring the XCD of their own image), all twelve waves,
    // the light layout of corr_fused_kernel (staged linearly in the whole ring), LYL::ROWS rows per pass
    {
        const int x = me & 7;
        const int nb = x < B ? (B - x + 7) >> 3 : 0;
        const int na = NA >> 3;
        const int r = anchor_wg ? me >> 3 : na + ((me - NA) >> 3);
        const int ns = na + 2 * nb;
        if (r < ns && nb > 0) {
            const int R = nb * TP;
            const int beg = (int)((long long)R * r / ns), end = (int)((long long)R * (r + 1) / ns);
            // rows of a pass: NW = 12 as the light workgroups of corr_fused_kernel (2 G rows per wave + the gather waves' second chunk: 64 rows
            // at C = 384, 24 at 768); NW = 16 one row pair per wave (32 rows)
            constexpr int PASS_ROWS = NW == 16 ? 2 * NW : LYL::ROWS;
            const int lr0 = NW == 16 ? 2 * wave8 : (mfma_team ? 2 * LYL::G * wave : LYL::MROWS + 2 * LYL::G * wave);
            unsigned epoch = 0;
            __builtin_amdgcn_s_setprio(3);
            for (int pb = beg; pb < end; pb += PASS_ROWS) {
                const int pe = min(end, pb + PASS_ROWS);
                if constexpr (NW == 16) {
                    if (pb + lr0 < pe) p1_sample_rows<NJ, PREC, NKCT, 1, true, false, CH>(prm, x, pb, pe, lr0, lane, ring, nullptr);
                } else {
                    if (pb + lr0 < pe) p1_sample_rows<NJ, PREC, NKCT, LYL::G, true, false, CH>(prm, x, pb, pe, lr0, lane, ring, nullptr);
                    if constexpr (LYL::XB > 0) {
                        const int lr1 = LYL::MROWS + 8 * 2 * LYL::G + 2 * wave;
                        if (!mfma_team && pb + lr1 < pe) p1_sample_rows<NJ, PREC, NKCT, 1, true, false, CH>(prm, x, pb, pe, lr1, lane, ring, nullptr);
                    }
                }
                epoch += NW;
                team_barrier(team_cnt, epoch, lane);
                const int nrows = pe - pb;
                const int to_edge = (((pb >> 7) + 1) << 7) - pb;
                const int n0 = min(nrows, to_edge);
                p1_copy_out<NJ, PREC, true, CH>(prm, x, pb, 0, n0, wave8, NW, lane, ring, fs_rsrc, csf_rsrc, cs_rsrc);
                if (n0 < nrows) p1_copy_out<NJ, PREC, true, CH>(prm, x, pb, n0, nrows - n0, wave8, NW, lane, ring, fs_rsrc, csf_rsrc, cs_rsrc);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this wave's write-through stores have landed
                epoch += NW;
                team_barrier(team_cnt, epoch, lane);
                if (tid == 0) p1_publish(prm, x, pb, nrows);
            }
            __builtin_amdgcn_s_setprio(0);
        }
    }
    if (stamp_on) ts[1] = __builtin_amdgcn_s_memrealtime();
    if (anchor_wg) {
        if (tid == 0) {
            const unsigned t = __hip_atomic_fetch_add(prm.done_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            fin[0] = t == gridDim.x - 1 ? 1.f : 0.f;
        }
        __syncthreads();
        if (fin[0] != 0.f) last_workgroup_tail_half<NW>(prm, Tfd, fin + 1, tid, n_items, (prm.debug & 256) ? ts : nullptr);
        return;
    }

    // ---- which item am I.  A sampler on a tile slot holds a self-correlation half by construction (slot k of XCD x: the k-th item in
    // item order that prefers x = half k & 1 of anchor x + 8 (k >> 1)): no second look at perms behind its phase 1 (2.3 us, stamps)
    int item;
    const int my_slot = (me - NA) >> 3;
    const int my_nb = (me & 7) < B ? (B - (me & 7) + 7) >> 3 : 0;
    if (my_slot < 2 * my_nb) {
        item = 2 * ((me & 7) + 8 * (my_slot >> 1)) + (my_slot & 1);
    } else if (my_slot < 4 * my_nb) {            // the next 2 nb slots: the halves of the XCD's inter tiles (pure cold gathers: they start first)
        item = 2 * (B + (me & 7) + 8 * ((my_slot - 2 * my_nb) >> 1)) + (my_slot & 1);
    } else if (wave8 == MW) {
        int pref0[ASSIGN_NB];
        assign_prefetch<false, 1>(prm, lane, 0, n_items, pref0);
        item = assign_tile<false, 1>(prm, me - NA, lane, 0, n_items, pref0);
        if (lane == 0) __hip_atomic_store(tile_slot, item, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((prm.debug & 256) && lane == 0) ts[6] = __builtin_amdgcn_s_memrealtime();
    } else {
        for (;;) {
            item = __hip_atomic_load(tile_slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (item >= 0) break;
            __builtin_amdgcn_s_sleep(2);
        }
        item = __builtin_amdgcn_readfirstlane(item);
    }
    const int tile = item >> 1, hf = item & 1;
    const int b = tile % B, p = tile / B;
    const int q0 = hf * HB;                                   // my first B-side point = my first output column
    const int ncols = max(0, min(HB, P - q0));
    const bool sameAB = p == 0;
    const int sA = b;
    const int sB = p == 0 ? b : p * B + b;
    const bool usePos = p == 1;
    int src = b;
    if (p >= 2) src = (int)prm.perms[(size_t)(p - 2) * B + b];
    src = __builtin_amdgcn_readfirstlane(src);
    const unsigned char* fsA = prm.fs + (size_t)sA * NCH2 * RS_SIDE;
    const unsigned char* csfA = prm.csf + (size_t)sA * NKC * RS_SIDE;
    const MapV mfB = usePos ? prm.feats_pos : prm.feats;
    const MapV mcB = usePos ? prm.code_pos : prm.code;
    const float* imgB = mfB.p + (long long)src * mfB.sn;
    const float* cimgB = mcB.p + (long long)src * mcB.sn;
    const float* coordsB = prm.coords2 + (size_t)b * P * 2;

    const int P2 = P * P;
    float* cd_out;
    float* loss_out = nullptr;
    float shift;
    if (p == 0) { cd_out = prm.intra_cd + (size_t)b * P2; shift = prm.shift[0]; }
    else if (p == 1) { cd_out = prm.inter_cd + (size_t)b * P2; shift = prm.shift[1]; }
    else {
        cd_out = prm.neg_cd + ((size_t)(p - 2) * B + b) * P2;
        loss_out = prm.neg_loss + ((size_t)(p - 2) * B + b) * P2;
        shift = prm.shift[2];
    }
    float* w_out = prm.saved_w ? prm.saved_w + ((size_t)p * B + b) * P2 : nullptr;
    const int a = (int)((reinterpret_cast<uintptr_t>(cd_out) >> 2) & 3);
    const bool vec_ok = (!loss_out || (int)((reinterpret_cast<uintptr_t>(loss_out) >> 2) & 3) == a) &&
                        (!w_out || (int)((reinterpret_cast<uintptr_t>(w_out) >> 2) & 3) == a);
    const bool rendezvous = loss_out != nullptr && prm.pointwise;
    f32x16 accf[MB], accc[MB];
    if (mfma_team) {
        // ================================================================= MFMA team
        auto stage_src = [&](int n) { return n < NKC ? csfA + (size_t)n * RS_SIDE : fsA + (size_t)(n - NKC) * RS_SIDE; };
        const unsigned ring_addr = __builtin_amdgcn_readfirstlane(lds_address(ring));
        if (wave == 0) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            bool ready = false;
            for (;;) {
                const unsigned c = __hip_atomic_load(prm.anchor_cnt + (size_t)sA * ANCHOR_CNT_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__builtin_amdgcn_readfirstlane(c) >= (unsigned)TP) { ready = true; break; }
                if ((long long)(__builtin_amdgcn_s_memrealtime() - t0) > prm.timeout_ticks) break;
                __builtin_amdgcn_s_sleep(10);
            }
            if (!ready) {
                // the samplers did not show up in time: sample the anchor here (identical bytes; see corr_fused_kernel)
                if (lane == 0) __hip_atomic_fetch_add(prm.done_cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int qq = 0; qq < TP; qq += 2 * LY::G) {
                    p1_sample_rows<NJ, PREC, NKCT, LY::G, false, false, CH>(prm, sA, qq, TP, 0, lane, ring, nullptr);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    p1_copy_out<NJ, PREC, false, CH>(prm, sA, qq, 0, 2 * LY::G, 0, 1, lane, ring, fs_rsrc, csf_rsrc, cs_rsrc);
                }
            }
            if (stamp_on) ts[2] = __builtin_amdgcn_s_memrealtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int n = 0; n < 3; ++n)
                for (int pc = 0; pc < 16; ++pc)
                    dma_piece_sc1(stage_src(n) + pc * 1024 + lane * 16, ring_addr + n * RS_STAGE + pc * 1024);
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        auto stage_head = [&](int n) {
            // my pieces of stage n have landed: only what I issued after them may still fly (wave 0 issued the first three stages alone)
            if (wave == 0 && n == 0) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            else if (wave == 0 && n == 1) { if constexpr (PPW == 4) asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); }
            else if (n + 2 < NT) { if constexpr (PPW == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
            else if (n + 1 < NT) { if constexpr (PPW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ring_barrier();                          // B(n)
        };
        auto stage_copy = [&](int n) {               // the A side of stage n + 3 into the slot stage n - 1 just left
            if (n + 3 < NT) {
                const unsigned char* s3 = stage_src(n + 3);
                const unsigned dst = ring_addr + ((n + 3) & (RS_NS - 1)) * RS_STAGE;
#pragma unroll
                for (int i = 0; i < PPW; ++i) {
                    const int pc = wave + MW * i;
                    dma_piece_sc1(s3 + pc * 1024 + lane * 16, dst + pc * 1024);
                }
            }
        };
        const int wr = wave >> 1, wc = wave & 1;
        const int brow = sameAB ? q0 + 32 * wc : 32 * wc;
        const bool abl_mfma = prm.debug & 1;         // (timing ablation: the stream without the multiplies)
        __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) { accc[i][e] = 0.f; accf[i][e] = 0.f; }
#pragma unroll
        for (int n = 0; n < NKC; ++n) {
            stage_head(n);
            stage_copy(n);
            const unsigned char* As = ring + (n & (RS_NS - 1)) * RS_STAGE;
            if (abl_mfma) continue;
            if constexpr (PREC == PREC_F32) mma_half_f(As, sameAB ? As : As + RS_SIDE, brow, kper, accc, lane, wr);
            else mma_half_h(As, sameAB ? As : As + RS_SIDE, brow, accc, lane, wr);       // (format H: channels kper .. 31 are zeros on both sides)
        }
        if constexpr (PREC == PREC_F32) {
#pragma unroll 1
            for (int n = NKC; n < NT; ++n) {
                stage_head(n);
                stage_copy(n);
                const unsigned char* As = ring + (n & (RS_NS - 1)) * RS_STAGE;
                const unsigned char* Bs = sameAB ? As : As + RS_SIDE;
                if (abl_mfma) continue;
                mma_half_f(As, Bs, brow, KC2, accf, lane, wr);
            }
        } else {
            // Software pipeline over the k-steps, ACROSS the stage barrier: the fragments of k-step 1 are read while k-step 0 multiplies,
            // then the wave passes B(n + 1) - every LDS read of stage n has landed by then (ring_barrier waits for lgkmcnt(0)), so its slot
            // may be overwritten - issues the A copies and reads k-step 0 of stage n + 1 while k-step 1 of stage n multiplies.  The same
            // 48 fragment registers, the same NT barriers; only the prologue's reads are exposed.
            auto frag = [&](HFrag<MB>& f, int n, int ks) {
                const unsigned char* As = ring + (n & (RS_NS - 1)) * RS_STAGE;
                half_frag_read(f, As, sameAB ? As : As + RS_SIDE, brow, 8192, ks, lane, wr);
            };
            HFrag<MB> x0, x1;
            stage_head(NKC);
            stage_copy(NKC);
            frag(x0, NKC, 0);
#pragma unroll 1
            for (int n = NKC; n < NT; ++n) {
                frag(x1, n, 1);
                __builtin_amdgcn_sched_barrier(0);
                if (!abl_mfma) half_frag_mma(x0, accf);
                __builtin_amdgcn_sched_barrier(0);
                if (n + 1 < NT) {
                    stage_head(n + 1);
                    stage_copy(n + 1);
                    frag(x0, n + 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!abl_mfma) half_frag_mma(x1, accf);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
    } else if (sameAB) {
        // ================================================================= gather team of a self-correlation item (B = A)
        for (int n = 0; n < NT; ++n) ring_barrier();
        if (gt < HB) { csc[gt] = 1.f; cscc[gt] = 1.f; }
    } else {
        // ================================================================= gather team: one point per lane group (64 points)
        const int g8 = gt & 7, prow = gt >> 3;       // prow = my point inside the half, q0 + prow inside the set
        const int gwave = wave;
        const int q = q0 + prow;
        float ss = 0.f, bsc = 0.f, ssc = 0.f, bscc = 0.f;
        if (lane < 8) {
            const int ql = 8 * gwave + lane, qq = q0 + ql;
            const f32x2 cxy = *reinterpret_cast<const f32x2*>(coordsB + coord_index(prm, qq));
            int4 yx;
            float4 w;
            point_taps(prm, cxy, qq, yx, w);
            tapof[ql] = taps_to_offsets(yx, mfB.sh, mfB.sw);
            tapoc[ql] = yx;
            tapw[ql] = w;
            prm.tapyx[(size_t)sB * TP + qq] = yx;                  // saved context of the backward
            prm.tapw[(size_t)sB * TP + qq] = w;
        }
        unsigned fo[4], pk;
        float4 tw;
        {
            const int4 o = tapof[prow];
            const int4 c = tapoc[prow];
            tw = tapw[prow];
            fo[0] = (unsigned)(o.x + 4 * g8) * 4u; fo[1] = (unsigned)(o.y + 4 * g8) * 4u;
            fo[2] = (unsigned)(o.z + 4 * g8) * 4u; fo[3] = (unsigned)(o.w + 4 * g8) * 4u;
            pk = (unsigned)(c.x >> 16) | ((unsigned)(c.x & 0xff) << 8) | ((unsigned)(c.w >> 16) << 16) | ((unsigned)(c.w & 0xff) << 24);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        struct HSet { f32x4 tv[4]; };
        auto issue_code = [&](HSet& g, int m) {
            const int k = m * kper + 4 * g8;
            const bool in = 4 * g8 < kper;
            const char* cb = reinterpret_cast<const char*>(cimgB);
            const bool v0 = in && k + 1 < prm.K, v1 = in && k + 3 < prm.K;
            const unsigned k0 = v0 ? 4u * k : 0u, k1 = v1 ? 4u * (k + 2) : 0u;
            const unsigned y0 = (pk & 0xff) * (unsigned)mcB.sh, x0 = ((pk >> 8) & 0xff) * (unsigned)mcB.sw;
            const unsigned y1 = ((pk >> 16) & 0xff) * (unsigned)mcB.sh, x1 = (pk >> 24) * (unsigned)mcB.sw;
            const unsigned co[4] = {(y0 + x0) * 4u, (y0 + x1) * 4u, (y1 + x0) * 4u, (y1 + x1) * 4u};
#pragma unroll
            for (int tq = 0; tq < 4; ++tq) {
                const f32x2 lo = *reinterpret_cast<const f32x2*>(cb + (co[tq] + k0));
                const f32x2 hi = *reinterpret_cast<const f32x2*>(cb + (co[tq] + k1));
                g.tv[tq] = f32x4{v0 ? lo[0] : 0.f, v0 ? lo[1] : 0.f, v1 ? hi[0] : 0.f, v1 ? hi[1] : 0.f};
            }
        };
        const __amdgpu_buffer_rsrc_t imgB_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(imgB), 0, 0x7fffffff, 0x00020000);
        auto issue_feat = [&](HSet& g, int f) {
            const int so = __builtin_amdgcn_readfirstlane(f * KC2 * 4);
#pragma unroll
            for (int tq = 0; tq < 4; ++tq)
                g.tv[tq] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(imgB_rsrc, (int)fo[tq], so, 0));
        };
        auto blend = [&](const HSet& g, float (&v)[4]) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = tw.x * g.tv[0][e] + tw.y * g.tv[1][e] + tw.z * g.tv[2][e] + tw.w * g.tv[3][e];
        };
        auto commit_feat = [&](const HSet& g, int m, bool decide) {
            unsigned char* dst = ring + (m & (RS_NS - 1)) * RS_STAGE + RS_SIDE;
            float v[4];
            blend(g, v);
            ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
            if constexpr (PREC == PREC_F32) {
                *reinterpret_cast<f32x4*>(dst + swz_f(prow, g8)) = f32x4{v[0], v[1], v[2], v[3]};
            } else {
                if (decide && bsc == 0.f) {
                    float mx = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
#pragma unroll
                    for (int mm = 4; mm >= 1; mm >>= 1) mx = fmaxf(mx, __shfl_xor(mx, mm, 64));
                    if (mx > 0.f) bsc = __builtin_ldexpf(1.f, -__builtin_amdgcn_frexp_expf(mx));
                }
                const float sc = bsc == 0.f ? 1.f : bsc;
                unsigned h0, l0, h1, l1;
                split_f16_pair(v[0] * sc, v[1] * sc, h0, l0);
                split_f16_pair(v[2] * sc, v[3] * sc, h1, l1);
                const bool odd = g8 & 1;
                const unsigned s0 = odd ? h0 : l0, s1 = odd ? h1 : l1;
                const unsigned r0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s0, 0xB1, 0xF, 0xF, true);
                const unsigned r1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s1, 0xB1, 0xF, 0xF, true);
                const u32x4 d = odd ? u32x4{r0, r1, l0, l1} : u32x4{h0, h1, r0, r1};
                *reinterpret_cast<u32x4*>(dst + (odd ? 8192 : 0) + swz_h(prow, g8 >> 1)) = d;
            }
        };
        f32x4 cv[NKC];                               // my point's raw code samples (4 channels per chunk): the context rows need them once the norm is known
        auto commit_code = [&](const HSet& g, int m) {
            unsigned char* dst = ring + (m & (RS_NS - 1)) * RS_STAGE + RS_SIDE;
            float v[4];
            blend(g, v);
            ssc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
            cv[m] = f32x4{v[0], v[1], v[2], v[3]};
            if constexpr (!CH) {
                *reinterpret_cast<f32x4*>(dst + swz_f(prow, g8)) = cv[m];
            } else {
                // format H, as a feature stage: a power-of-two prescale per point from the first chunk in which it is non-zero among the
                // first two, fp16 hi / lo split, one 16-byte store per lane (lanes beyond kper / 4 hold zeros: the stage's padding)
                if (m < 2 && bscc == 0.f) {
                    float mx = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
#pragma unroll
                    for (int mm = 4; mm >= 1; mm >>= 1) mx = fmaxf(mx, __shfl_xor(mx, mm, 64));
                    if (mx > 0.f) bscc = __builtin_ldexpf(1.f, -__builtin_amdgcn_frexp_expf(mx));
                }
                const float sc = bscc == 0.f ? 1.f : bscc;
                unsigned h0, l0, h1, l1;
                split_f16_pair(v[0] * sc, v[1] * sc, h0, l0);
                split_f16_pair(v[2] * sc, v[3] * sc, h1, l1);
                const bool odd = g8 & 1;
                const unsigned s0 = odd ? h0 : l0, s1 = odd ? h1 : l1;
                const unsigned r0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s0, 0xB1, 0xF, 0xF, true);
                const unsigned r1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s1, 0xB1, 0xF, 0xF, true);
                const u32x4 d = odd ? u32x4{r0, r1, l0, l1} : u32x4{h0, h1, r0, r1};
                *reinterpret_cast<u32x4*>(dst + (odd ? 8192 : 0) + swz_h(prow, g8 >> 1)) = d;
            }
        };
        auto finish_codes = [&]() {
            float sq = ssc;
#pragma unroll
            for (int mm = 4; mm >= 1; mm >>= 1) sq += __shfl_xor(sq, mm, 64);
            const bool valid = q < P;
            const float nr = valid ? sqrtf(sq) : 0.f;
            const float inv = valid ? __builtin_amdgcn_rcpf(fmaxf(nr, 1e-10f)) : 0.f;
            const float unsc = (CH && bscc != 0.f) ? 1.f / bscc : 1.f;
            if (g8 == 0) { cscc[prow] = inv * unsc; prm.nrm[(size_t)sB * TP + q] = nr; }
            float* crow = prm.cs + ((size_t)sB * TP + q) * prm.LDK;
#pragma unroll
            for (int m = 0; m < NKC; ++m) {
                const int k = m * kper + 4 * g8;
                if (4 * g8 < kper && k < prm.KQ) *reinterpret_cast<f32x4*>(crow + k) = cv[m] * inv;
            }
        };
        HSet ga, gb;
        auto issue = [&](HSet& g, int m) {
            if (m < NKC) issue_code(g, m);
            else issue_feat(g, m - NKC);
        };
        auto commit = [&](HSet& g, int m) {
            if (m < NKC) { commit_code(g, m); if (m == NKC - 1) finish_codes(); }
            else commit_feat(g, m, m - NKC < 2);
        };
        static_assert(NT >= 8, "the static head covers the code chunks and the first feature stages");
        {
            HSet gc;
            issue(ga, 0);
            issue(gb, 1);
            issue(gc, 2);
            commit(ga, 0);
            issue(ga, 3);
            commit(gb, 1);
            commit(gc, 2);
            commit(ga, 3);
        }
        if ((prm.debug & 256) && tid == 64 * MW) ts[11] = __builtin_amdgcn_s_memrealtime();
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            ring_barrier();                          // B(n)
            if (n & 1) { if (n >= 2) commit(gb, n + 2); issue(gb, n + 4); }
            else { if (n >= 2) commit(ga, n + 2); issue(ga, n + 4); }
        }
        constexpr int IT = (NT - 8) / 2;
        constexpr int TAIL0 = 4 + 2 * IT;
#pragma unroll 1
        for (int n = 4; n < TAIL0; n += 2) {
            ring_barrier();                          // B(n)
            commit_feat(ga, n + 2, false);
            __builtin_amdgcn_sched_barrier(0);
            issue_feat(ga, n + 4 - NKC);
            ring_barrier();                          // B(n + 1)
            commit_feat(gb, n + 3, false);
            __builtin_amdgcn_sched_barrier(0);
            issue_feat(gb, n + 5 - NKC);
        }
#pragma unroll
        for (int n = TAIL0; n < NT; ++n) {
            ring_barrier();                          // B(n)
            if (n + 2 < NT) { if (n & 1) commit_feat(gb, n + 2, false); else commit_feat(ga, n + 2, false); }
            if (n + 4 < NT) { if (n & 1) issue_feat(gb, n + 4 - NKC); else issue_feat(ga, n + 4 - NKC); }
        }
        {
            float sq = ss;
#pragma unroll
            for (int mm = 4; mm >= 1; mm >>= 1) sq += __shfl_xor(sq, mm, 64);
            if (g8 == 0) csc[prow] = ((PREC == PREC_F16X3 && bsc != 0.f) ? 1.f / bsc : 1.f) / fmaxf(sqrtf(sq), 1e-10f);
        }
    }
    if (stamp_on) ts[7] = __builtin_amdgcn_s_memrealtime();

    __syncthreads();                             // E0: the ring is dead, csc / cscc complete
    if (stamp_on) ts[3] = __builtin_amdgcn_s_memrealtime();
    if (mfma_team) park_half<MB>(accf, Tfd + a, P, q0, csc, lane, wave >> 1, wave & 1);
    __syncthreads();                             // E1: Tfd complete
    if (mfma_team) {
        park_half<MB>(accc, Tcd + a, P, q0, cscc, lane, wave >> 1, wave & 1);
    } else {
        // my partial row sums of fd: four lanes per row over the 8 gather waves, a fixed trip count of independent predicated loads
        const int row = gt >> 2, t = gt & 3;
        const float* srcr = Tfd + a + (row < P ? row : 0) * P + q0;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int i = 0; i < HB / 4; i += 2) {
            const int c0 = 4 * i + t, c1 = c0 + 4;
            const float v0 = srcr[c0 < ncols ? c0 : 0], v1 = srcr[c1 < ncols ? c1 : 0];
            s0 += c0 < ncols ? v0 : 0.f;
            s1 += c1 < ncols ? v1 : 0.f;
        }
        float sfull = s0 + s1;
        sfull += __shfl_xor(sfull, 1, 64);
        sfull += __shfl_xor(sfull, 2, 64);
        const float mine = row < P ? sfull : 0.f;
        if (t == 0) rsum[row] = mine;
        float ws = t == 0 ? mine : 0.f;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) ws += __shfl_xor(ws, m, 64);
        if (lane == 0) red[40 + wave] = ws;
    }
    __syncthreads();                             // E2: Tcd, rsum, the eight partial sums of fd
    unsigned tick = 0u;                          // (the closing ticket, taken here and not at the end: see corr_fused_kernel)
    const bool early_ticket = !(prm.debug & 4);
    if (tid == 0) {
        const float sfd = ((red[40] + red[41]) + (red[42] + red[43])) + ((red[44] + red[45]) + (red[46] + red[47]));
        __hip_atomic_store(prm.gran + item, (1ull << 32) | __builtin_bit_cast(unsigned, sfd), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (early_ticket) tick = __hip_atomic_fetch_add(prm.done_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // my row sums for my partner (one 8-byte write-through granule per row)
    if (prm.pointwise && !mfma_team && wave < 2 && gt < P)
        __hip_atomic_store(prm.rowg + (size_t)item * TP + gt, (1ull << 32) | __builtin_bit_cast(unsigned, rsum[gt]), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    if (stamp_on) ts[4] = __builtin_amdgcn_s_memrealtime();

    // ---- the sweeps: the 16-byte groups of the flat [P][P] outputs that touch my columns [q0, q0 + ncols) - whole groups as vectors,
    // the straddling ones by element.  E = a + row * P + col is the element's index relative to the 16-byte aligned word before the
    // output tile (the three outputs share `a` when vec_ok).
    const float cmin = prm.cmin, cmax = prm.cmax;
    constexpr int GPR = HB / 4 + 1;              // groups a row's run of <= 64 elements can touch
    float loss_part = 0.f, clamp_part = 0.f;
    float om = 0.f;
    // mode 0: cd; 1: w + the two sums; 2: the negative loss
    auto sweep = [&](int mode) {
        for (int it = tid; it < P * GPR; it += NTHR) {
            const int r = it / GPR, gi = it - r * GPR;
            const int E0 = a + r * P + q0, E1 = E0 + ncols;
            const int g = (E0 >> 2) + gi;
            if (4 * g >= E1) continue;
            const float rm = rowmean[r] + shift;
            const f32x4 cd4 = *reinterpret_cast<const f32x4*>(Tcd + 4 * g);
            f32x4 fd4 = f32x4{0.f, 0.f, 0.f, 0.f};
            if (mode != 0) fd4 = *reinterpret_cast<const f32x4*>(Tfd + 4 * g);
            f32x4 o4;
            bool in[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int E = 4 * g + k;
                in[k] = E >= E0 && E < E1;
                const float cd = cd4[k];
                if (mode == 0) { o4[k] = cd; continue; }
                const float wv = fd4[k] - rm;
                const float cl = fminf(fmaxf(cd, cmin), cmax);
                const float lp = -cl * wv;
                if (mode == 1) {
                    const unsigned pass = (cd >= cmin && cd <= cmax) ? 1u : 0u;
                    o4[k] = __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, wv) & ~1u) | pass);
                    if (in[k]) { loss_part += lp; clamp_part += cl; }
                } else {
                    o4[k] = __builtin_fmaf(-om, cl, lp);               // loss = -clamp(cd) * (fd_centred + old_mean - shift)
                }
            }
            float* out = mode == 0 ? cd_out : (mode == 1 ? w_out : loss_out);
            if (!out) continue;
            const int e0 = 4 * g - a;
            if (vec_ok && in[0] && in[3]) {
                __builtin_nontemporal_store(o4, reinterpret_cast<f32x4*>(out + e0));
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (in[k]) out[e0 + k] = o4[k];
            }
        }
    };
    sweep(0);                                    // cd needs nobody
    if (stamp_on) ts[12] = __builtin_amdgcn_s_memrealtime();
    // ---- whole-row means: my partner's partial row sums (bounded wait; it zeroes mine after reading them, I zero its)
    if (!mfma_team && wave < 2) {
        float other = 0.f;
        bool bad = false;
        if (prm.pointwise && gt < P) {
            unsigned long long* src = prm.rowg + (size_t)(item ^ 1) * TP + gt;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            unsigned long long x;
            for (;;) {
                x = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((x >> 32) == 1ull) break;
                if ((long long)(__builtin_amdgcn_s_memrealtime() - t0) > 100ll * prm.timeout_ticks) { bad = true; break; }     // 20 ms: not our device
                __builtin_amdgcn_s_sleep(8);
            }
            other = __builtin_bit_cast(float, (unsigned)x);
            __hip_atomic_store(src, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (__any(bad) && lane == 0) __hip_atomic_fetch_add(prm.done_cnt + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gt < TP) {
            const float mine = rsum[gt];
            const float h0 = hf == 0 ? mine : other, h1 = hf == 0 ? other : mine;       // the same order on both sides: the same bits
            rowmean[gt] = (prm.pointwise && gt < P) ? (bad ? __builtin_nanf("") : (h0 + h1) / (float)P) : 0.f;
        }
    }
    __syncthreads();                             // E3: rowmean
    if (stamp_on) ts[8] = __builtin_amdgcn_s_memrealtime();
    sweep(1);
    if (stamp_on) ts[13] = __builtin_amdgcn_s_memrealtime();
    bool gave_up = false;
    float* omv = red + 8;                        // [0] old_mean, [1] applied
    if (loss_out) {                              // (workgroup-uniform)
        if (wave8 == 0) {
            float omx = 0.f, applied = 1.f;
            if (rendezvous) {
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                const int PB = 2 * B;
                float acc = 0.f;
                bool ok_all = true;
                for (int i0 = 0; i0 < PB && ok_all; i0 += 64) {
                    const int i = i0 + lane;
                    unsigned long long x = 0;
                    for (;;) {
                        bool ok = true;
                        if (i < PB) {
                            x = __hip_atomic_load(prm.gran + (size_t)p * PB + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            ok = (x >> 32) == 1ull;
                        }
                        if (__all(ok)) break;
                        if ((long long)(__builtin_amdgcn_s_memrealtime() - t0) > prm.timeout_ticks) { ok_all = false; break; }
                        __builtin_amdgcn_s_sleep(16);
                    }
                    const float v = i < PB ? __builtin_bit_cast(float, (unsigned)x) : 0.f;
                    acc += wave_tree_sum(v);
                }
                if (ok_all) omx = acc * (1.f / ((float)B * (float)P2));     // same expression as the tail
                else applied = 0.f;
            }
            if (lane == 0) { omv[0] = omx; omv[1] = applied; }
        }
        __syncthreads();                         // E4: old_mean
        if (stamp_on) ts[14] = __builtin_amdgcn_s_memrealtime();
        om = omv[0];
        gave_up = omv[1] == 0.f;
        sweep(2);
        if (stamp_on) ts[15] = __builtin_amdgcn_s_memrealtime();
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        loss_part += __shfl_xor(loss_part, m, 64);
        clamp_part += __shfl_xor(clamp_part, m, 64);
    }
    if (lane == 0) { csc[wave8 * 2] = loss_part; csc[wave8 * 2 + 1] = clamp_part; }      // (csc is dead since the parks; red[16 ..] would reach red[40 ..] with sixteen waves)
    if (gave_up) __threadfence();                // (rare) whoever repairs this item must see its cd / loss
    __syncthreads();
    unsigned long long* gst = prm.gran + n_items;                  // [n_items][3]
    if (tid == 0) {
        float s1 = 0.f, s2 = 0.f;
        for (int w = 0; w < NW; ++w) { s1 += csc[w * 2]; s2 += csc[w * 2 + 1]; }
        if (gave_up) __hip_atomic_fetch_add(prm.done_cnt + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long* g3 = gst + (size_t)item * 3;
        __hip_atomic_store(g3 + 0, (1ull << 32) | __builtin_bit_cast(unsigned, s1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(g3 + 1, (1ull << 32) | __builtin_bit_cast(unsigned, s2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(g3 + 2, (1ull << 32) | __builtin_bit_cast(unsigned, gave_up ? 0.f : 1.f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!early_ticket) tick = __hip_atomic_fetch_add(prm.done_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fin[0] = tick == gridDim.x - 1 ? 1.f : 0.f;
    }
    if (stamp_on) ts[5] = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
    if (fin[0] == 0.f) return;
    last_workgroup_tail_half<NW>(prm, Tfd, fin + 1, tid, n_items, (prm.debug & 256) ? ts : nullptr);
}

// ------------------------------------------------------------------------------------------------------ launch
#define STEGO_HALF_ONE(PR, N, NK, NWV)                                                                 \
    do {                                                                                               \
        hipError_t e_ = ensure_dynamic_lds(reinterpret_cast<const void*>(&corr_fused_half_kernel<PR, N, NK, NWV>), lds); \
        if (e_ != hipSuccess) return e_;                                                               \
        hipLaunchKernelGGL((corr_fused_half_kernel<PR, N, NK, NWV>), grid, dim3(64 * NWV), lds, stream, prm); \
        return hipSuccess;                                                                             \
    } while (0)
#define STEGO_HALF_NK(PR, N, NWV)                                                                      \
    do {                                                                                               \
        if (prm.NKC == 1) STEGO_HALF_ONE(PR, N, 1, NWV);                                               \
        else if (prm.NKC == 2) STEGO_HALF_ONE(PR, N, 2, NWV);                                          \
        else if (prm.NKC == 3) STEGO_HALF_ONE(PR, N, 3, NWV);                                          \
        else STEGO_HALF_ONE(PR, N, 4, NWV);                                                            \
    } while (0)
// (the caller - launch_corr_fused - has checked half_launch_covers)
hipError_t launch_fused_half(const FusedParams& prm, int precision, hipStream_t stream)
{
    const int n_items = 2 * prm.n_sets * prm.B;
    const dim3 grid(prm.n_anchor_wg + n_items);
    const int lds = RING_LDS_BYTES;
    // C = 384: sixteen waves (eight MFMA waves, one row pair per wave in phase 1); C = 768: twelve (its phase 1 holds 96 tap registers);
    // STEGO_DEBUG bit 2: twelve everywhere (same-process A/B)
    if (prm.C == 384 && !(prm.debug & 2)) {
        if (precision == PREC_F32) STEGO_HALF_NK(PREC_F32, 3, 16); else STEGO_HALF_NK(PREC_F16X3, 3, 16);
    }
    if (precision == PREC_F32) { if (prm.C == 384) STEGO_HALF_NK(PREC_F32, 3, 12); else STEGO_HALF_NK(PREC_F32, 6, 12); }
    else { if (prm.C == 384) STEGO_HALF_NK(PREC_F16X3, 3, 12); else STEGO_HALF_NK(PREC_F16X3, 6, 12); }
}
#undef STEGO_HALF_NK
#undef STEGO_HALF_ONE

}  // namespace stego
