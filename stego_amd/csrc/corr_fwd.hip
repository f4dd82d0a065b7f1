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
#include "corr_common.h"

namespace stego {

// ------------------------------------------------------------------ smem carve (dynamic LDS)
constexpr int SD_ROWMEAN = 0;                     // float rowmean[128]
constexpr int SD_RED = SD_ROWMEAN + TP * 4;       // float red[64]
constexpr int SD_CSC = SD_RED + 64 * 4;           // float csc[128]: 1 / ||b_j|| of the gathered B points
constexpr int SD_TAPO = SD_CSC + TP * 4;          // int4 tapo[128]: element offsets of the 4 taps in the B image
constexpr int SD_TAPW = SD_TAPO + TP * 16;        // float4 tapw[128]
constexpr int SD_BIG = SD_TAPW + TP * 16;         // 5376: two stage buffers, aliased by the result tiles
constexpr int FEAT_SIDE_F32 = TP * LDA * 4;       // 34816 = 34 x 1 KB : one operand, one 64-channel chunk
constexpr int FEAT_SIDE_F16 = 2 * TP * LDH * 2;  // 36864 = 36 x 1 KB : hi + lo
constexpr int SM_TILES_BYTES = 2 * TP * LDT * 4;  // epilogue: fd + cd tiles

// One staged chunk of the contraction on v_mfma_f32_32x32x2_f32.  Wave (wr,wc) owns the
// 64x64 quadrant; lanes 0-31 take k = kk..kk+3, lanes 32-63 k = kk+4..kk+7 of every 8-wide
// k group via one ds_read_b128 per operand (any k permutation is fine as long as A and B agree).
__device__ __forceinline__ void mma_chunk_f32(const float* __restrict__ As, const float* __restrict__ Bs,
                                              f32x16 (&acc)[2][2], int lane, int wr, int wc)
{
    const int r = lane & 31, half = lane >> 5;
    const float* a0p = As + (64 * wr + r) * LDA + 4 * half;
    const float* a1p = a0p + 32 * LDA;
    const float* b0p = Bs + (64 * wc + r) * LDA + 4 * half;
    const float* b1p = b0p + 32 * LDA;
#pragma unroll 2
    for (int kk = 0; kk < KC; kk += 8) {       // always the full 64 channels: both operand images are zero-padded
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(a0p + kk);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(a1p + kk);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(b0p + kk);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(b1p + kk);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[1][1], 0, 0, 0);
        }
    }
}

// Code contraction (exact f32) over the whole K in one go; operands [128][ld] floats, ld = KQ+4.
__device__ __forceinline__ void mma_code_f32(const float* __restrict__ As, const float* __restrict__ Bs, int kq, int ld,
                                             f32x16 (&acc)[2][2], int lane, int wr, int wc)
{
    const int r = lane & 31, half = lane >> 5;
    const float* a0p = As + (64 * wr + r) * ld + 4 * half;
    const float* a1p = a0p + 32 * ld;
    const float* b0p = Bs + (64 * wc + r) * ld + 4 * half;
    const float* b1p = b0p + 32 * ld;
    for (int kk = 0; kk < kq; kk += 8) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(a0p + kk);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(a1p + kk);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(b0p + kk);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(b1p + kk);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[1][1], 0, 0, 0);
        }
    }
}

// Split-fp16 contraction of one chunk: a.b ~= ah.bh + ah.bl + al.bh (the al.bl term is < 2^-22).
// Stage layout: hi[128][LDH] then lo[128][LDH] (fp16).  Each lane reads 8 consecutive k
// (lanes 0-31: kk..kk+7, lanes 32-63: kk+8..kk+15) per operand with one ds_read_b128.
__device__ __forceinline__ void mma_chunk_f16x3(const half_t* __restrict__ As, const half_t* __restrict__ Bs,
                                                f32x16 (&acc)[2][2], int lane, int wr, int wc)
{
    constexpr int LO = TP * LDH;
    const int r = lane & 31, half = lane >> 5;
    const half_t* a0p = As + (64 * wr + r) * LDH + 8 * half;
    const half_t* a1p = a0p + 32 * LDH;
    const half_t* b0p = Bs + (64 * wc + r) * LDH + 8 * half;
    const half_t* b1p = b0p + 32 * LDH;
#pragma unroll 2
    for (int kk = 0; kk < KC; kk += 16) {
        const f16x8 ah0 = *reinterpret_cast<const f16x8*>(a0p + kk), al0 = *reinterpret_cast<const f16x8*>(a0p + LO + kk);
        const f16x8 ah1 = *reinterpret_cast<const f16x8*>(a1p + kk), al1 = *reinterpret_cast<const f16x8*>(a1p + LO + kk);
        const f16x8 bh0 = *reinterpret_cast<const f16x8*>(b0p + kk), bl0 = *reinterpret_cast<const f16x8*>(b0p + LO + kk);
        const f16x8 bh1 = *reinterpret_cast<const f16x8*>(b1p + kk), bl1 = *reinterpret_cast<const f16x8*>(b1p + LO + kk);
        // small cross terms first, then the leading term; accumulators interleaved
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, bh0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, bh1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1, bh0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1, bh1, acc[1][1], 0, 0, 0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bl0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bl1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bl0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bl1, acc[1][1], 0, 0, 0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bh0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bh1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bh0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bh1, acc[1][1], 0, 0, 0);
    }
}

// Linear async copy of npieces KiB from global to LDS, split over the 4 waves.  The LDS destination
// of a global_load_lds is wave-uniform base + lane*16, i.e. each piece is one contiguous KiB.
__device__ __forceinline__ void issue_copy(const unsigned char* __restrict__ gsrc, unsigned char* lds_dst, int npieces,
                                           int wave, int lane)
{
    for (int pc = wave; pc < npieces; pc += 4) {
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(gsrc + (size_t)pc * 1024 + lane * 16),
            (__attribute__((address_space(3))) void*)(lds_dst + pc * 1024), 16, 0, 0);
    }
}

// ---- B-operand gather, software-pipelined through the MFMA stream of the previous chunk.
// Thread mapping (256 threads): SLOTS = 64/V lanes cover one point's chunk (one contiguous 256-byte read per
// tap per point when channels are contiguous), PPI points per pass, ITEMS passes ("items").
// Per chunk and lane: ITEMS x 4 tap loads into registers during the first half of the MFMA k groups, blended
// and written to the other LDS stage buffer during the second half (by then they have landed; the blend's VALU
// work rides in the MFMA shadow).
template <int V> struct GatherRegs {
    typedef typename VecT<V>::type vec;
    static constexpr int SLOTS = KC / V;
    static constexpr int ITEMS = TP * SLOTS / NTHREADS;
    static constexpr int PPI = NTHREADS / SLOTS;
    vec tv[ITEMS][4];               // the 4 taps of every item of one chunk (two of these are alive: chunk t+1, t+2)
};
// items [j0, j0 + n): loads.  `chunk` = image + c0 * channel_stride (wave-uniform): SGPR base + 32-bit lane offsets
// (tap offset from the LDS table + this lane's channel slot).
template <int V>
__device__ __forceinline__ void gather_issue(GatherRegs<V>& g, const float* __restrict__ chunk,
                                             const int4* __restrict__ tapo, int lane_off, int prow, int j0, int n)
{
    typedef typename VecT<V>::type vec;
    const char* cb = reinterpret_cast<const char*>(chunk);
#pragma unroll
    for (int i = 0; i < n; ++i) {
        const int j = j0 + i;
        const int4 o = tapo[j * GatherRegs<V>::PPI + prow];
        if constexpr (V == 4) {
            g.tv[j][0] = *reinterpret_cast<const vec*>(cb + (unsigned)(o.x + lane_off) * 4u);
            g.tv[j][1] = *reinterpret_cast<const vec*>(cb + (unsigned)(o.y + lane_off) * 4u);
            g.tv[j][2] = *reinterpret_cast<const vec*>(cb + (unsigned)(o.z + lane_off) * 4u);
            g.tv[j][3] = *reinterpret_cast<const vec*>(cb + (unsigned)(o.w + lane_off) * 4u);
        } else {
            const float* b = chunk + lane_off;
            g.tv[j][0] = b[o.x]; g.tv[j][1] = b[o.y]; g.tv[j][2] = b[o.z]; g.tv[j][3] = b[o.w];
        }
    }
}

// items [j0, j0 + n): blend the 4 taps, accumulate the points' sums of squares, write the LDS operand image
//   PREC_F32   : float [128][LDA]            PREC_F16X3: fp16 hi[128][LDH] then lo[128][LDH]
// Split mode stages the RAW sampled values as fp16 halves, so every point gets its own power-of-two scale bsc[j]
// (chosen from the first chunk in which the point is non-zero: |x| * bsc in [0.5, 1)) - F.normalize is scale invariant
// and so is this path for features of any magnitude; the epilogue's column scale divides it out again.
template <int V, int PREC>
__device__ __forceinline__ void gather_commit(const GatherRegs<V>& g, const float4* __restrict__ tapw, bool chok,
                                              void* __restrict__ dst_, float (&ss)[GatherRegs<V>::ITEMS],
                                              float (&bsc)[GatherRegs<V>::ITEMS], int slot, int prow, int j0, int n)
{
    constexpr int PPI = GatherRegs<V>::PPI;
    const int col = slot * V;
#pragma unroll
    for (int i = 0; i < n; ++i) {
        const int j = j0 + i;
        const int q = j * PPI + prow;
        const float4 w = tapw[q];
        float v[V];
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float t0, t1, t2, t3;
            if constexpr (V == 1) { t0 = g.tv[j][0]; t1 = g.tv[j][1]; t2 = g.tv[j][2]; t3 = g.tv[j][3]; }
            else { t0 = g.tv[j][0][e]; t1 = g.tv[j][1][e]; t2 = g.tv[j][2][e]; t3 = g.tv[j][3][e]; }
            float r = w.x * t0 + w.y * t1 + w.z * t2 + w.w * t3;
            r = chok ? r : 0.f;                       // channels beyond C (generic path): zero padding
            v[e] = r;
            s += r * r;
        }
        ss[j] += s;
        if constexpr (PREC == PREC_F32) {
            float* d = static_cast<float*>(dst_) + q * LDA + col;
            if constexpr (V == 4) *reinterpret_cast<f32x4*>(d) = f32x4{v[0], v[1], v[2], v[3]};
            else d[0] = v[0];
        } else {
            half_t* dh = static_cast<half_t*>(dst_) + q * LDH + col;
            half_t* dl = dh + TP * LDH;
            if (bsc[j] == 0.f) {                      // (uniform over the lanes of a point)
                float mx = 0.f;
#pragma unroll
                for (int e = 0; e < V; ++e) mx = fmaxf(mx, fabsf(v[e]));
#pragma unroll
                for (int m = GatherRegs<V>::SLOTS / 2; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
                if (mx > 0.f) bsc[j] = __builtin_ldexpf(1.f, -__builtin_amdgcn_frexp_expf(mx));
            }
            const float sc = bsc[j] == 0.f ? 1.f : bsc[j];
#pragma unroll
            for (int e = 0; e < V; ++e) v[e] *= sc;
            if constexpr (V == 4) {
                unsigned h0, l0, h1, l1;
                split_f16_pair(v[0], v[1], h0, l0);
                split_f16_pair(v[2], v[3], h1, l1);
                *reinterpret_cast<u32x2*>(dh) = u32x2{h0, h1};
                *reinterpret_cast<u32x2*>(dl) = u32x2{l0, l1};
            } else {
                unsigned h0, l0;
                split_f16_pair(v[0], 0.f, h0, l0);
                *reinterpret_cast<unsigned short*>(dh) = (unsigned short)(h0 & 0xffffu);
                *reinterpret_cast<unsigned short*>(dl) = (unsigned short)(l0 & 0xffffu);
            }
        }
    }
}

// Result tiles are parked in LDS in the FLAT layout of the outputs, T[a + row * P + col], so that the epilogue
// is a linear sweep: 16-byte LDS reads, 16-byte global stores.  `a` = the output tile's start address / 4 mod 4
// (tiles are P*P floats apart and P*P is odd, so they are only 4-byte aligned): with the same shift in LDS
// both sides of the copy are 16-byte aligned at the same time.  colscale (or null) = 1/||b_j|| of a raw B side.
// C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
__device__ __forceinline__ void park_flat(const f32x16 (&acc)[2][2], float* __restrict__ T, int P, const float* colscale,
                                          int lane, int wr, int wc)
{
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int col = 64 * wc + 32 * ni + (lane & 31);
        const float sc = colscale ? colscale[col] : 1.f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 64 * wr + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < P && col < P) T[row * P + col] = acc[mi][ni][r] * sc;
            }
    }
}

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

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][2])
{
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
}

constexpr int TILE_THREADS = 2 * NTHREADS;        // waves 0-3: MFMA team, waves 4-7: gather team

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
    if (blockIdx.x == 0 && threadIdx.x < prm.n_sets) {
        const int p = threadIdx.x;
        float fs = 0.f, ls = 0.f, cs = 0.f;
        for (int b = 0; b < B; ++b) {
            const float* st = prm.stats + ((size_t)p * B + b) * 4;
            fs += st[0]; ls += st[1]; cs += st[2];
        }
        const float om = prm.pointwise ? fs * inv_cnt : 0.f;
        if (prm.saved_mean) prm.saved_mean[p] = om;
        if (prm.mode == 0 && p < 2) prm.loss_means[p] = (ls - om * cs) * inv_cnt;
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
// bytes of one stage buffer: max(feature chunk pair, code operand pair)
int dense_stage_bytes(int precision, int LDK)
{
    const int f = 2 * (precision == PREC_F32 ? FEAT_SIDE_F32 : FEAT_SIDE_F16);
    const int c = 2 * TP * LDK * 4;
    return f > c ? f : c;
}

int dense_lds_bytes(int precision, int LDK)
{
    const int st = 2 * dense_stage_bytes(precision, LDK);
    return SD_BIG + (st > SM_TILES_BYTES ? st : SM_TILES_BYTES);
}

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
    static int have[4] = {0, 0, 0, 0};
    const void* fns[4] = {reinterpret_cast<const void*>(&corr_tile_kernel<PREC_F32, 4>),
                          reinterpret_cast<const void*>(&corr_tile_kernel<PREC_F32, 1>),
                          reinterpret_cast<const void*>(&corr_tile_kernel<PREC_F16X3, 4>),
                          reinterpret_cast<const void*>(&corr_tile_kernel<PREC_F16X3, 1>)};
    if (have[which] < lds) {
        hipError_t e = hipFuncSetAttribute(fns[which], hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        have[which] = lds;
    }
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
