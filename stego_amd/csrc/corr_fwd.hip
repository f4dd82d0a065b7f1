// Forward of STEGO's ContrastiveCorrelationLoss for gfx950 (MI355X): the correlation / loss tile kernel.
//
// One workgroup (4 waves, one per SIMD) = one (pair-set p, image b) tile:  A side = anchor set b (image b
// sampled at coords1[b]),  B side = {same | feats_pos/code_pos[b] @ coords2[b] | feats/code[perm[b]] @ coords2[b]}.
//   fd = A_f . B_f^T (contraction C, no_grad side)      cd = A_c . B_c^T (contraction K)
// followed by the reference's epilogue: row-centring of fd (fd -= fd.mean([3,4]), modules.py:332),
// -clamp(cd) * (fd - shift), flat 16-byte stores, per-tile partial sums.  The batch-global old_mean (:331)
// needs every tile of a pair-set, so corr_finalize_kernel applies it from the per-tile sums (fixed order).
//
// Operands (hybrid staging, round 1 third design):
//   * A features: the anchor set is used by all 2+n_neg tiles of its image, so sample_norm_kernel
//     (corr_sample.hip) samples + normalises it ONCE into ready-made LDS images; the MFMA waves copy chunk t+1
//     with global_load_lds (no VGPRs, no VALU) while they multiply chunk t;
//   * B features: every B set is used by exactly one tile, so materialising it (52 MB written + 52 MB read
//     per step in the second design, which made the sampler HBM-bound at 43 us) is pure overhead: the tile
//     gathers its B operand straight from the source image.  The 4 bilinear taps of chunk t+1 are loaded into
//     REGISTERS (32 x 16 B per lane) before the MFMAs of chunk t are issued and blended into LDS after them,
//     so the gather latency hides in the MFMA shadow of the SAME wave (an f32 MFMA stream starves any OTHER
//     wave on its SIMD, so loader waves do not work).  B is staged RAW; its L2 norm is accumulated during the
//     gather and applied as a column scale of fd in the epilogue;
//   * codes (K <= 72, both sides): normalised by the sampler (the backward needs them anyway), one
//     global_load_lds stage at the end.
// Contraction arithmetic: PREC_F32 = v_mfma_f32_32x32x2_f32 (exact fp32) for both correlations;
// PREC_F16X3 = the feature correlation on split-fp16 (hi*hi + hi*lo + lo*hi, v_mfma_f32_32x32x16_f16, fp32
// accumulate, ~1e-6 abs error on a cosine); the code correlation stays exact f32 because its sign decides the
// clamp mask of the backward.
//
// Reference path: src/modules.py:275-398.
#include "corr_tile.h"
#include "host_util.h"

namespace stego {

// Epilogue of the dense kernel (4 waves): row means by two lanes per row, then a flat sweep over the tile
// (the first version looped rows with 4-byte stores: 186 store instructions per wave, ~24 us of the kernel).
__device__ __forceinline__ void tile_epilogue_flat(const CorrParams& prm, const float* __restrict__ Tfd,
                                                   const float* __restrict__ Tcd, float* __restrict__ rowmean,
                                                   float* __restrict__ red, int p, int b, bool direct, int a,
                                                   float* cd_out, float* loss_out, float* w_out, float shift, bool vec_ok)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = prm.B, P = prm.P;
    float fd_part = 0.f;
    {
        const int row = tid >> 1, half = tid & 1;          // threads 0..255; any further waves idle here
        const int hl = (P + 1) >> 1;
        float s0 = 0.f, s1 = 0.f;
        if (row < P) {
            const float* src = Tfd + a + row * P;
            const int c1 = half ? P : hl;
            int c = half ? hl : 0;
            for (; c + 2 <= c1; c += 2) { s0 += src[c]; s1 += src[c + 1]; }
            if (c < c1) s0 += src[c];
        }
        float sfull = s0 + s1;
        sfull += __shfl_xor(sfull, 1, 64);
        if (half == 0) {
            fd_part = sfull;
            if (row < TP) rowmean[row] = prm.pointwise ? sfull / (float)P : 0.f;    // fd.mean([3,4]) (modules.py:332)
            else fd_part = 0.f;
        }
    }
    __syncthreads();

    const int P2 = P * P;
    const float cmin = prm.cmin, cmax = prm.cmax;
    const float invP = 1.f / (float)P;
    float loss_part = 0.f, clamp_part = 0.f;
    if (!(prm.debug & 8)) {
        const int nvec = (a + P2 + 3) >> 2;
        for (int v = tid; v < nvec; v += (int)blockDim.x) {
            const int f0 = 4 * v;
            const f32x4 fd4 = *reinterpret_cast<const f32x4*>(Tfd + f0);
            const f32x4 cd4 = *reinterpret_cast<const f32x4*>(Tcd + f0);
            f32x4 w4, lp4;
            bool ok[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = f0 + k - a;
                ok[k] = e >= 0 && e < P2;
                const int r = min(max((int)(((float)e + 0.5f) * invP), 0), P - 1);
                const float w = fd4[k] - (rowmean[r] + shift);                 // fd_centred - shift
                const float cl = fminf(fmaxf(cd4[k], cmin), cmax);
                const float lp = -cl * w;                                      // loss without the old_mean term
                // saved for the backward: w with the clamp pass-mask in its mantissa LSB (1 ulp), so that the
                // backward never has to read cd again (13 MB less HBM traffic in its bandwidth-bound phase)
                const unsigned pass = (cd4[k] >= cmin && cd4[k] <= cmax) ? 1u : 0u;
                w4[k] = __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, w) & ~1u) | pass);
                lp4[k] = lp;
                if (ok[k]) { loss_part += lp; clamp_part += cl; }
            }
            if (prm.debug & 4) continue;
            const int e0 = f0 - a;
            if (vec_ok && ok[0] && ok[3]) {
                *reinterpret_cast<f32x4*>(cd_out + e0) = cd4;
                if (loss_out) *reinterpret_cast<f32x4*>(loss_out + e0) = lp4;
                if (w_out) *reinterpret_cast<f32x4*>(w_out + e0) = w4;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (ok[k]) {
                        cd_out[e0 + k] = cd4[k];
                        if (loss_out) loss_out[e0 + k] = lp4[k];
                        if (w_out) w_out[e0 + k] = w4[k];
                    }
            }
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        fd_part += __shfl_xor(fd_part, m, 64);
        loss_part += __shfl_xor(loss_part, m, 64);
        clamp_part += __shfl_xor(clamp_part, m, 64);
    }
    if (lane == 0) { red[wave * 3 + 0] = fd_part; red[wave * 3 + 1] = loss_part; red[wave * 3 + 2] = clamp_part; }
    __syncthreads();
    if (tid == 0) {
        float* st = prm.stats + ((size_t)p * B + b) * 4;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { s0 += red[w * 3]; s1 += red[w * 3 + 1]; s2 += red[w * 3 + 2]; }
        st[0] = s0; st[1] = s1; st[2] = s2; st[3] = 0.f;
    }
}


template <int PREC, int V>
__global__ void __launch_bounds__(TILE_THREADS) corr_tile_kernel(const CorrParams prm, const int stage_bytes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* rowmean = reinterpret_cast<float*>(smem + SD_ROWMEAN);
    float* red = reinterpret_cast<float*>(smem + SD_RED);
    float* csc = reinterpret_cast<float*>(smem + SD_CSC);
    int4* tapo = reinterpret_cast<int4*>(smem + SD_TAPO);
    float4* tapw = reinterpret_cast<float4*>(smem + SD_TAPW);
    unsigned char* stage = smem + SD_BIG;
    float* Tfd = reinterpret_cast<float*>(smem + SD_BIG);   // epilogue alias of the stage buffers
    float* Tcd = Tfd + TP * LDT;
    constexpr int FSIDE = PREC == PREC_F32 ? FEAT_SIDE_F32 : FEAT_SIDE_F16;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool mfma_team = wave8 < 4;            // wave-uniform role
    const int wave = wave8 & 3;
    const int gt = tid & (NTHREADS - 1);         // index inside the 256-thread team
    const int wr = wave >> 1, wc = wave & 1;
    const int B = prm.B;
    const int tile = blockIdx.x;
    const int b = tile % B, p = tile / B;       // the 2+n_neg tiles of image b share blockIdx%8: its anchor set is L2-local
    const bool direct = prm.mode == 1;
    const bool sameAB = !direct && p == 0;
    const int sA = b;
    const int sB = direct ? B + b : (p == 0 ? b : p * B + b);
    const int NCH = prm.NCH;
    const int cside = TP * prm.LDK * 4;          // bytes of one code operand (a multiple of 1 KiB)
    const unsigned char* fsA = static_cast<const unsigned char*>(prm.fs) + (size_t)sA * NCH * FSIDE;
    const unsigned char* csA = reinterpret_cast<const unsigned char*>(prm.cs) + (size_t)sA * cside;
    const unsigned char* csB = reinterpret_cast<const unsigned char*>(prm.cs) + (size_t)sB * cside;

    // ---- B side source image (modules.py:369-386) and its tap table (written by the sampler)
    const bool usePos = direct || p == 1;
    const float* imgB;
    int scB;
    {
        int src = b;
        if (!direct && p >= 2) src = (int)prm.perms[(size_t)(p - 2) * B + b];
        const float* bp = usePos ? prm.feats_pos.p : prm.feats.p;
        const long long sn = usePos ? prm.feats_pos.sn : prm.feats.sn;
        const int sh = usePos ? prm.feats_pos.sh : prm.feats.sh, sw = usePos ? prm.feats_pos.sw : prm.feats.sw;
        scB = usePos ? prm.feats_pos.sc : prm.feats.sc;
        imgB = bp + (long long)src * sn;
        if (tid < TP) {
            tapo[tid] = taps_to_offsets(prm.tapyx[(size_t)sB * TP + tid], sh, sw);
            tapw[tid] = prm.tapw[(size_t)sB * TP + tid];      // padding points: weights 0 -> zero rows
        }
    }
    // debug bit 256: phase stamps of every workgroup on the 100 MHz global clock (tools/stamps.py)
    unsigned long long* ts = reinterpret_cast<unsigned long long*>(prm.stats + (size_t)prm.n_sets * B * 4 + 256) + (size_t)blockIdx.x * 8;
    const bool stamp_on = (prm.debug & 256) && tid == 0;
    if (stamp_on) ts[0] = __builtin_amdgcn_s_memrealtime();
    __syncthreads();                              // tap table visible

    const bool gatherB = !sameAB;
    // debug bit 16 (experiment): every image walks the channel chunks from its own start, so that the tiles in flight
    // do not all hit the same 256-byte residue of every pixel at the same time (L2 / fabric channel spreading)
    const int rot = (prm.debug & 16) ? b % NCH : 0;
    auto chunk_of = [&](int t) { int tt = t + rot; return tt >= NCH ? tt - NCH : tt; };
    auto copies = [&](int t) {                    // async LDS copies of stage t: A features / both code operands
        unsigned char* dst = stage + (t & 1) * stage_bytes;
        if (t < NCH) {
            issue_copy(fsA + (size_t)chunk_of(t) * FSIDE, dst, FSIDE / 1024, wave, lane);
        } else {
            issue_copy(csA, dst, cside / 1024, wave, lane);
            if (!sameAB) issue_copy(csB, dst + cside, cside / 1024, wave, lane);
        }
    };

    f32x16 accf[2][2], accc[2][2];
    if (mfma_team) {
        // ================================================================= MFMA team
        zero_acc(accf);
        copies(0);
        for (int t = 0; t < NCH; ++t) {
            sync_after_lds_dma();                // B1(t): stage t complete (copies landed; gathered B: written before)
            copies(t + 1);                       // stage NCH = the code operands
            const unsigned char* Ab = stage + (t & 1) * stage_bytes;
            const unsigned char* Bb = sameAB ? Ab : Ab + FSIDE;
            if constexpr (PREC == PREC_F32)
                mma_chunk_f32(reinterpret_cast<const float*>(Ab), reinterpret_cast<const float*>(Bb), accf, lane, wr, wc);
            else
                mma_chunk_f16x3(reinterpret_cast<const half_t*>(Ab), reinterpret_cast<const half_t*>(Bb), accf, lane, wr, wc);
        }
        zero_acc(accc);
        sync_after_lds_dma();                    // B1(NCH): code stage landed
        const unsigned char* Ab = stage + (NCH & 1) * stage_bytes;
        const unsigned char* Bb = sameAB ? Ab : Ab + cside;
        mma_code_f32(reinterpret_cast<const float*>(Ab), reinterpret_cast<const float*>(Bb), prm.KQ, prm.LDK, accc, lane, wr, wc);
    } else {
        // ================================================================= gather team
        // Rolling pipeline over the items (point group x channel slot) of the B operand: while the MFMA team
        // multiplies chunk t, item j of chunk t+1 (loaded one chunk time ago) is blended into the free stage buffer
        // and its registers immediately take the taps of item j of chunk t+2.  In split-fp16 mode the MFMAs run on
        // the matrix cores, so this VALU / memory work genuinely overlaps them (in f32 mode the fp32 MFMA owns the
        // VALU and the two teams time-slice: no worse than doing it in one wave).
        GatherRegs<V> g;
        constexpr int ITEMS = GatherRegs<V>::ITEMS;
        float ss[ITEMS], bsc[ITEMS];
        const int gslot = gt % GatherRegs<V>::SLOTS, gprow = gt / GatherRegs<V>::SLOTS;
        const int lane_off = gslot * V * scB;                         // this lane's channel slot inside a chunk
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            ss[j] = 0.f;
            bsc[j] = 0.f;
        }
        // Channels beyond C exist only in the generic path (V == 1, the host takes V == 4 only when C % 64 == 0):
        // such a lane re-reads the last channel and its values are zeroed at commit.  Prefetches past the last
        // chunk re-read it and are never committed.
        auto chunk_ptr = [&](int t) { return V == 4 ? imgB + (long long)min(chunk_of(min(t, NCH - 1)) * KC, prm.C - KC) * scB : imgB; };
        auto lane_ofs = [&](int t) { return V == 4 ? lane_off : min(chunk_of(min(t, NCH - 1)) * KC + gslot, prm.C - 1) * scB; };
        auto chunk_ok = [&](int t) { return V == 4 || chunk_of(min(t, NCH - 1)) * KC + gslot < prm.C; };
        if (gatherB) {
            gather_issue<V>(g, chunk_ptr(0), tapo, lane_ofs(0), gprow, 0, ITEMS);
            gather_commit<V, PREC>(g, tapw, chunk_ok(0), stage + FSIDE, ss, bsc, gslot, gprow, 0, ITEMS);
            gather_issue<V>(g, chunk_ptr(1), tapo, lane_ofs(1), gprow, 0, ITEMS);
        }
        for (int t = 0; t < NCH; ++t) {
            __syncthreads();                     // B1(t): the MFMA team is done with stage t-1 = the buffer written next
            if (gatherB && t + 1 < NCH) {
                const float* nxt = chunk_ptr(t + 2);
                const int nxt_off = lane_ofs(t + 2);
                const bool ok = chunk_ok(t + 1);
                void* dstB = stage + ((t + 1) & 1) * stage_bytes + FSIDE;
#pragma unroll
                for (int j = 0; j < ITEMS; ++j) {
                    gather_commit<V, PREC>(g, tapw, ok, dstB, ss, bsc, gslot, gprow, j, 1);
                    gather_issue<V>(g, nxt, tapo, nxt_off, gprow, j, 1);
                }
            }
        }
        __syncthreads();                         // B1(NCH)
        // ---- 1 / ||b_j|| of the gathered side (F.normalize eps, modules.py:276); the anchor side is pre-normalised
        constexpr int SLOTS = GatherRegs<V>::SLOTS, PPI = GatherRegs<V>::PPI;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            float sq = ss[j];
#pragma unroll
            for (int m = SLOTS / 2; m >= 1; m >>= 1) sq += __shfl_xor(sq, m, 64);
            if (gslot == 0)
                csc[j * PPI + gprow] = sameAB ? 1.f : ((PREC == PREC_F16X3 && bsc[j] != 0.f) ? 1.f / bsc[j] : 1.f) / fmaxf(sqrtf(sq), 1e-10f);
        }
    }

    // ---- where this tile's outputs go
    const int P = prm.P, P2 = P * P;
    float* cd_out;
    float* loss_out = nullptr;
    float shift;
    if (direct) { cd_out = prm.neg_cd + (size_t)b * P2; loss_out = prm.neg_loss + (size_t)b * P2; shift = prm.shift[0]; }
    else if (p == 0) { cd_out = prm.intra_cd + (size_t)b * P2; shift = prm.shift[0]; }
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
    __syncthreads();                             // stage buffers are dead, csc complete: park the result tiles
    if (stamp_on) ts[1] = __builtin_amdgcn_s_memrealtime();
    if (mfma_team) {
        park_flat(accf, Tfd + a, P, csc, lane, wr, wc);
        park_flat(accc, Tcd + a, P, nullptr, lane, wr, wc);
    }
    __syncthreads();
    if (stamp_on) ts[2] = __builtin_amdgcn_s_memrealtime();
    tile_epilogue_flat(prm, Tfd, Tcd, rowmean, red, p, b, direct, a, cd_out, loss_out, w_out, shift, vec_ok);
    if (stamp_on) ts[4] = __builtin_amdgcn_s_memrealtime();
}

// Applies the batch-global mean of each pair-set:  old_mean_p = mean_{b,hw,ij} fd  (modules.py:331),
// fd_final = fd_centred + old_mean (:333, the middle fd.mean() is identically 0), hence
//   loss = lp - old_mean * clamp(cd);   mean(loss) = (sum lp - old_mean * sum clamp) / (B*P^2).
// grid.x = blocks over the loss tensor elements of the sets that output one (4096 per block, inside
// one set); block 0 also writes the scalar means and saved_mean.
__global__ void __launch_bounds__(NTHREADS) corr_finalize_kernel(const CorrParams prm)
{
    const int B = prm.B, P2 = prm.P * prm.P;
    const float inv_cnt = 1.f / ((float)B * (float)P2);
    const int first_loss_set = prm.mode == 1 ? 0 : 2;
    if (blockIdx.x == 0) {
        __shared__ float set_loss[NTHREADS];
        if (threadIdx.x < prm.n_sets) {
            const int p = threadIdx.x;
            float fs = 0.f, ls = 0.f, cs = 0.f;
            for (int b = 0; b < B; ++b) {
                const float* st = prm.stats + ((size_t)p * B + b) * 4;
                fs += st[0]; ls += st[1]; cs += st[2];
            }
            const float om = prm.pointwise ? fs * inv_cnt : 0.f;
            set_loss[p] = ls - om * cs;
            if (prm.saved_mean) prm.saved_mean[p] = om;
            if (prm.mode == 0 && p < 2) prm.loss_means[p] = (ls - om * cs) * inv_cnt;
        }
        __syncthreads();
        if (prm.mode == 0 && threadIdx.x == 0) {     // torch.cat(negative losses).mean(), pair-set order
            float nsum = 0.f;
            for (int pp = 2; pp < prm.n_sets; ++pp) nsum += set_loss[pp];
            prm.loss_means[2] = prm.n_sets > 2 ? nsum * inv_cnt / (float)(prm.n_sets - 2) : 0.f;
        }
    }
    if (!prm.pointwise) return;
    const int n_loss_sets = prm.n_sets - first_loss_set;
    const size_t per_set = (size_t)B * P2;
    const size_t span = (size_t)NTHREADS * 16;
    const size_t blocks_per_set = (per_set + span - 1) / span;
    const int ps = (int)(blockIdx.x / blocks_per_set);
    if (ps >= n_loss_sets) return;
    // Every lane sums the pair-set's B tile sums itself (same addresses across the wave: one request per load, same
    // order as block 0 above), so these loads fly together with the data loads below: one round trip, no barrier.
    const float cmin = prm.cmin, cmax = prm.cmax;
    const size_t beg = (size_t)(blockIdx.x % blocks_per_set) * span;
    float* loss = prm.neg_loss + (size_t)ps * per_set;
    const float* cd = prm.neg_cd + (size_t)ps * per_set;
    const float* stp = prm.stats + (size_t)(first_loss_set + ps) * B * 4;
    const bool vec_ok = (per_set % 4 == 0) && ((reinterpret_cast<uintptr_t>(loss) & 15) == 0) &&
                        ((reinterpret_cast<uintptr_t>(cd) & 15) == 0);
    if (vec_ok) {
        f32x4 l[4], c[4];
        size_t e[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            e[i] = beg + ((size_t)i * NTHREADS + threadIdx.x) * 4;
            const size_t ec = e[i] < per_set ? e[i] : 0;
            l[i] = *reinterpret_cast<const f32x4*>(loss + ec);
            c[i] = *reinterpret_cast<const f32x4*>(cd + ec);
        }
        float fs = 0.f;
        for (int b = 0; b < B; ++b) fs += stp[b * 4];
        const float om = fs * inv_cnt;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (e[i] < per_set) {
#pragma unroll
                for (int k = 0; k < 4; ++k) l[i][k] -= om * fminf(fmaxf(c[i][k], cmin), cmax);
                *reinterpret_cast<f32x4*>(loss + e[i]) = l[i];
            }
        }
    } else {
        float fs = 0.f;
        for (int b = 0; b < B; ++b) fs += stp[b * 4];
        const float om = fs * inv_cnt;
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const size_t e = beg + (size_t)i * NTHREADS + threadIdx.x;
            if (e < per_set) loss[e] -= om * fminf(fmaxf(cd[e], cmin), cmax);
        }
    }
}

// ------------------------------------------------------------------------------ launch
hipError_t launch_corr_tile(const CorrParams& prm, int precision, hipStream_t stream)
{
    const int stage = dense_stage_bytes(precision, prm.LDK);
    const int lds = dense_lds_bytes(precision, prm.LDK);
    // 16-byte gathers need channels-last maps with 16-byte aligned pixels; anything else takes the scalar path
    auto ok4 = [&](const MapV& m) {
        return m.sc == 1 && prm.C % KC == 0 && (m.sn % 4) == 0 && (m.sh % 4) == 0 && (m.sw % 4) == 0 &&
               (reinterpret_cast<uintptr_t>(m.p) % 16) == 0 &&
               ((long long)(prm.H - 1) * m.sh + (long long)(prm.W - 1) * m.sw + prm.C) * 4 < (1ll << 32);
    };
    const bool v4 = ok4(prm.feats) && ok4(prm.feats_pos);
    const int which = (precision == PREC_F32 ? 0 : 2) + (v4 ? 0 : 1);
    const void* fns[4] = {reinterpret_cast<const void*>(&corr_tile_kernel<PREC_F32, 4>),
                          reinterpret_cast<const void*>(&corr_tile_kernel<PREC_F32, 1>),
                          reinterpret_cast<const void*>(&corr_tile_kernel<PREC_F16X3, 4>),
                          reinterpret_cast<const void*>(&corr_tile_kernel<PREC_F16X3, 1>)};
    hipError_t ea = ensure_dynamic_lds(fns[which], lds);
    if (ea != hipSuccess) return ea;
    const dim3 grid(prm.n_sets * prm.B), block(TILE_THREADS);
    switch (which) {
        case 0: hipLaunchKernelGGL((corr_tile_kernel<PREC_F32, 4>), grid, block, lds, stream, prm, stage); break;
        case 1: hipLaunchKernelGGL((corr_tile_kernel<PREC_F32, 1>), grid, block, lds, stream, prm, stage); break;
        case 2: hipLaunchKernelGGL((corr_tile_kernel<PREC_F16X3, 4>), grid, block, lds, stream, prm, stage); break;
        default: hipLaunchKernelGGL((corr_tile_kernel<PREC_F16X3, 1>), grid, block, lds, stream, prm, stage); break;
    }
    return hipGetLastError();
}

// finalize: block 0 writes scalars; the rest fix the loss tensors of the sets that output one
hipError_t launch_corr_finalize(const CorrParams& prm, hipStream_t stream)
{
    const int first_loss_set = prm.mode == 1 ? 0 : 2;
    const int n_loss_sets = prm.n_sets - first_loss_set;
    const size_t per_set = (size_t)prm.B * prm.P * prm.P;
    const size_t span = (size_t)NTHREADS * 16;
    const size_t blocks_per_set = (per_set + span - 1) / span;
    size_t nblk = prm.pointwise ? blocks_per_set * (size_t)n_loss_sets : 1;
    if (nblk < 1) nblk = 1;
    hipLaunchKernelGGL(corr_finalize_kernel, dim3((unsigned)nblk), dim3(NTHREADS), 0, stream, prm);
    return hipGetLastError();
}

}  // namespace stego
