// Fused forward of STEGO's ContrastiveCorrelationLoss for gfx950 (MI355X).
//
// One workgroup = one (pair-set p, image b) tile:  A side = image b sampled at coords1[b],
// B side = {same | feats_pos/code_pos[b] @ coords2[b] | feats/code[perm[b]] @ coords2[b]}.
// Per tile:  bilinear gather (channels-last: one contiguous 256 B read per tap per 64-channel
// chunk) -> LDS -> MFMA for fd = A_f.B_f^T (contraction C) and cd = A_c.B_c^T (contraction K) on
// RAW sampled values; the L2 normalisation is applied in the epilogue as row/col scales
// (fd[i][j] * invn_A[i] * invn_B[j]) - the norms are accumulated during the gather.
// Epilogue: row-centring of fd (the reference's fd -= fd.mean([3,4]), modules.py:332),
// clamp(cd)*(fd-shift), coalesced row stores, per-tile partial sums.  The batch-global
// old_mean (modules.py:331) needs every tile of the pair-set, so it is applied by
// corr_finalize_kernel from the per-tile sums (deterministic order, no atomics).
//
// Production path (two launches + finalize):
//   corr_sample.hip : sample_norm_kernel  - every (role, image) set sampled + L2-normalised ONCE, placed on
//                     the XCD of its source image, written as ready-made LDS images;
//   corr_tile_kernel (here) - one workgroup (4 waves, one per SIMD) per (pair-set, image) tile: both
//                     operands are dense, so the MFMA waves themselves issue async global_load_lds copies
//                     of chunk t+1 (no VGPRs, no VALU) and run the MFMAs of chunk t in their shadow
//                     (an f32 MFMA stream starves any OTHER wave on its SIMD, so loader waves do not work).
// Contraction arithmetic: PREC_F32 = v_mfma_f32_32x32x2_f32 (exact fp32) for both correlations;
// PREC_BF16X3 = the feature correlation fd (no_grad side, 85 % of the flops) on split-bf16
// (hi*hi + hi*lo + lo*hi, v_mfma_f32_32x32x16_bf16, fp32 accumulate, ~1e-6 abs error on a cosine); the code
// correlation cd stays exact f32 because its sign decides the clamp mask of the backward.
//
// corr_fwd_kernel (fused gather, 4 waves, no overlap, f32) is the first version kept as a cross-check
// (STEGO_FWD_VARIANT=0).
//
// Reference path: src/modules.py:275-398.
#include "corr_common.h"

namespace stego {

// ------------------------------------------------------------------ simple kernel smem carve
constexpr int SM_NRM = 0;                         // float nrm[4][128]: Af, Bf, Ac, Bc
constexpr int SM_ROWMEAN = SM_NRM + 4 * TP * 4;   // float rowmean[128]
constexpr int SM_RED = SM_ROWMEAN + TP * 4;       // float red[64]
constexpr int SM_BIG = SM_RED + 64 * 4;           // 2816, 16-byte aligned
constexpr int SM_STAGE_BYTES = 2 * TP * LDA * 4 + 3 * 256 * 16;
constexpr int SM_TILES_BYTES = 2 * TP * LDT * 4;
constexpr int SM_FWD_TOTAL = SM_BIG + (SM_TILES_BYTES > SM_STAGE_BYTES ? SM_TILES_BYTES : SM_STAGE_BYTES);

// ------------------------------------------------------------------ dense tile kernel smem carve
constexpr int SD_ROWMEAN = 0;                     // float rowmean[128]
constexpr int SD_RED = SD_ROWMEAN + TP * 4;       // float red[64]
constexpr int SD_BIG = SD_RED + 64 * 4;           // 768: two stage buffers, aliased by the result tiles
constexpr int FEAT_SIDE_F32 = TP * LDA * 4;       // 34816 = 34 x 1 KB : one operand, one 64-channel chunk
constexpr int FEAT_SIDE_BF16 = 2 * TP * LDH * 2;  // 36864 = 36 x 1 KB : hi + lo

// One staged chunk of the contraction on v_mfma_f32_32x32x2_f32.  Wave (wr,wc) owns the
// 64x64 quadrant; lanes 0-31 take k = kk..kk+3, lanes 32-63 k = kk+4..kk+7 of every 8-wide
// k group via one ds_read_b128 per operand (any k permutation is fine as long as A and B agree).
__device__ __forceinline__ void mma_chunk_f32(const float* __restrict__ As, const float* __restrict__ Bs, int kc8,
                                              f32x16 (&acc)[2][2], int lane, int wr, int wc)
{
    const int r = lane & 31, half = lane >> 5;
    const float* a0p = As + (64 * wr + r) * LDA + 4 * half;
    const float* a1p = a0p + 32 * LDA;
    const float* b0p = Bs + (64 * wc + r) * LDA + 4 * half;
    const float* b1p = b0p + 32 * LDA;
    for (int kk = 0; kk < kc8; kk += 8) {
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

// Split-bf16 contraction of one chunk: a.b ~= ah.bh + ah.bl + al.bh (the al.bl term is < 2^-16).
// Stage layout: hi[128][LDH] then lo[128][LDH] (bf16).  Each lane reads 8 consecutive k
// (lanes 0-31: kk..kk+7, lanes 32-63: kk+8..kk+15) per operand with one ds_read_b128.
__device__ __forceinline__ void mma_chunk_bf16x3(const __bf16* __restrict__ As, const __bf16* __restrict__ Bs, int kc16,
                                                 f32x16 (&acc)[2][2], int lane, int wr, int wc)
{
    constexpr int LO = TP * LDH;
    const int r = lane & 31, half = lane >> 5;
    const __bf16* a0p = As + (64 * wr + r) * LDH + 8 * half;
    const __bf16* a1p = a0p + 32 * LDH;
    const __bf16* b0p = Bs + (64 * wc + r) * LDH + 8 * half;
    const __bf16* b1p = b0p + 32 * LDH;
    for (int kk = 0; kk < kc16; kk += 16) {
        const bf16x8 ah0 = *reinterpret_cast<const bf16x8*>(a0p + kk), al0 = *reinterpret_cast<const bf16x8*>(a0p + LO + kk);
        const bf16x8 ah1 = *reinterpret_cast<const bf16x8*>(a1p + kk), al1 = *reinterpret_cast<const bf16x8*>(a1p + LO + kk);
        const bf16x8 bh0 = *reinterpret_cast<const bf16x8*>(b0p + kk), bl0 = *reinterpret_cast<const bf16x8*>(b0p + LO + kk);
        const bf16x8 bh1 = *reinterpret_cast<const bf16x8*>(b1p + kk), bl1 = *reinterpret_cast<const bf16x8*>(b1p + LO + kk);
        // small cross terms first, then the leading term; accumulators interleaved
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al0, bh0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al0, bh1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al1, bh0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al1, bh1, acc[1][1], 0, 0, 0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bl0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bl1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bl0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bl1, acc[1][1], 0, 0, 0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bh0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bh1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bh0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bh1, acc[1][1], 0, 0, 0);
    }
}

// Scale the raw accumulators by the inverse norms and park the tile in LDS.
// C/D layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
__device__ __forceinline__ void park_tile(const f32x16 (&acc)[2][2], float* __restrict__ T,
                                          const float* __restrict__ nrmA, const float* __restrict__ nrmB,
                                          int lane, int wr, int wc)
{
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int col = 64 * wc + 32 * ni + (lane & 31);
        const float sB = 1.f / fmaxf(nrmB[col], 1e-10f);      // F.normalize eps (modules.py:276)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 64 * wr + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float sA = 1.f / fmaxf(nrmA[row], 1e-10f);
                T[row * LDT + col] = acc[mi][ni][r] * sA * sB;
            }
        }
    }
}

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][2])
{
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

// Epilogue shared by both kernels (NW waves).  Tfd/Tcd hold the normalised correlation tiles.
// A wave owns rows wave, wave+NW, ...; its lanes walk the columns, so LDS reads are conflict-free
// and every global store instruction covers one contiguous run of a row.
template <int NW>
__device__ __forceinline__ void tile_epilogue(const CorrParams& prm, const float* __restrict__ Tfd,
                                              const float* __restrict__ Tcd, float* __restrict__ rowmean,
                                              float* __restrict__ red, int p, int b, bool direct)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = prm.B, P = prm.P;
    // row means of fd over the B-side points (fd.mean([3,4]), modules.py:332)
    float fdsum_part = 0.f;
    for (int r = wave; r < TP; r += NW) {
        float s = 0.f;
        if (r < P)
            for (int c = lane; c < P; c += 64) s += Tfd[r * LDT + c];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
        if (lane == 0) {
            rowmean[r] = prm.pointwise ? s / (float)P : 0.f;
            fdsum_part += s;
        }
    }
    const float fd_sum = block_sum<NW>(fdsum_part, red);   // (barriers inside: rowmean visible after)

    const int P2 = P * P;
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
    const float cmin = prm.cmin, cmax = prm.cmax;
    float loss_part = 0.f, clamp_part = 0.f;
    for (int r = wave; r < P; r += NW) {
        const float rm = rowmean[r] + shift;
        for (int c = lane; c < P; c += 64) {
            const int idx = r * P + c;
            const float w = Tfd[r * LDT + c] - rm;                     // fd_centred - shift
            const float cdv = Tcd[r * LDT + c];
            const float cl = fminf(fmaxf(cdv, cmin), cmax);
            const float lp = -cl * w;                                  // loss without the old_mean term
            if (!(prm.debug & 4)) {
                cd_out[idx] = cdv;
                if (loss_out) loss_out[idx] = lp;
                if (w_out) w_out[idx] = w;
            }
            loss_part += lp;
            clamp_part += cl;
        }
    }
    const float loss_sum = block_sum<NW>(loss_part, red);
    const float clamp_sum = block_sum<NW>(clamp_part, red);
    if (tid == 0) {
        float* st = prm.stats + ((size_t)p * B + b) * 4;
        st[0] = fd_sum; st[1] = loss_sum; st[2] = clamp_sum; st[3] = 0.f;
    }
}

// Which maps feed the B side of tile (p, b) (modules.py:369-386).  Selected by value: a pointer
// into the kernarg struct would force the whole struct into scratch.
struct TileSel {
    MapV mfB, mcB;
    const float* coordsB;
    int imgB;
    bool sameAB, direct;
};

__device__ __forceinline__ TileSel select_tile(const CorrParams& prm, int p, int b)
{
    TileSel s;
    s.direct = prm.mode == 1;
    const bool usePos = s.direct || p == 1;
    s.sameAB = !s.direct && p == 0;
    s.mfB.p = usePos ? prm.feats_pos.p : prm.feats.p;     s.mcB.p = usePos ? prm.code_pos.p : prm.code.p;
    s.mfB.sn = usePos ? prm.feats_pos.sn : prm.feats.sn;  s.mcB.sn = usePos ? prm.code_pos.sn : prm.code.sn;
    s.mfB.sc = usePos ? prm.feats_pos.sc : prm.feats.sc;  s.mcB.sc = usePos ? prm.code_pos.sc : prm.code.sc;
    s.mfB.sh = usePos ? prm.feats_pos.sh : prm.feats.sh;  s.mcB.sh = usePos ? prm.code_pos.sh : prm.code.sh;
    s.mfB.sw = usePos ? prm.feats_pos.sw : prm.feats.sw;  s.mcB.sw = usePos ? prm.code_pos.sw : prm.code.sw;
    s.coordsB = (!s.direct && p >= 1) ? prm.coords2 : prm.coords1;
    s.imgB = b;
    if (!s.direct && p >= 2) s.imgB = (int)prm.perms[(size_t)(p - 2) * prm.B + b];
    return s;
}

// tap tables for the 2 x 128 points of a tile; t in [0,256): t<128 -> A point t, else B point t-128
__device__ __forceinline__ void build_taps(const CorrParams& prm, const TileSel& s, int b, int t, int4* tapf, int4* tapc,
                                           float4* tapw)
{
    const int side = t >> 7, q = t & (TP - 1);
    const float* cimg = s.direct ? nullptr
                                 : (side == 0 ? prm.coords1 + (size_t)b * prm.P * 2 : s.coordsB + (size_t)b * prm.P * 2);
    int4 yx; float4 w;
    tap_for_point(q, prm.P, prm.S, prm.H, prm.W, s.direct, cimg, yx, w);
    tapf[t] = taps_to_offsets(yx, side == 0 ? prm.feats.sh : s.mfB.sh, side == 0 ? prm.feats.sw : s.mfB.sw);
    tapc[t] = taps_to_offsets(yx, side == 0 ? prm.code.sh : s.mcB.sh, side == 0 ? prm.code.sw : s.mcB.sw);
    tapw[t] = w;
}

// =============================================================== simple kernel (4 waves, f32)
template <int VF, int VC>
__global__ void __launch_bounds__(NTHREADS) corr_fwd_kernel(const CorrParams prm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* nrm = reinterpret_cast<float*>(smem + SM_NRM);
    float* rowmean = reinterpret_cast<float*>(smem + SM_ROWMEAN);
    float* red = reinterpret_cast<float*>(smem + SM_RED);
    float* As = reinterpret_cast<float*>(smem + SM_BIG);
    float* Bs = As + TP * LDA;
    int4* tapf = reinterpret_cast<int4*>(Bs + TP * LDA);
    int4* tapc = tapf + 256;
    float4* tapw = reinterpret_cast<float4*>(tapc + 256);
    float* Tfd = reinterpret_cast<float*>(smem + SM_BIG);  // epilogue alias
    float* Tcd = Tfd + TP * LDT;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int B = prm.B;
    const int tile = blockIdx.x;
    const int b = tile % B, p = tile / B;      // all pair-sets of image b share blockIdx%8 (XCD L2) when B%8==0
    const TileSel sel = select_tile(prm, p, b);
    build_taps(prm, sel, b, tid, tapf, tapc, tapw);
    __syncthreads();

    const float* Bsrc = sel.sameAB ? As : Bs;
    constexpr int BF = VF == 4 ? 4 : 8, BC = 4;

    f32x16 accf[2][2];
    zero_acc(accf);
    {
        float ssA[TP * (KC / VF) / NTHREADS], ssB[TP * (KC / VF) / NTHREADS];
#pragma unroll
        for (int i = 0; i < TP * (KC / VF) / NTHREADS; ++i) { ssA[i] = 0.f; ssB[i] = 0.f; }
        const float* imgA = prm.feats.p + (long long)b * prm.feats.sn;
        const float* imgBp = sel.mfB.p + (long long)sel.imgB * sel.mfB.sn;
        for (int c0 = 0; c0 < prm.C; c0 += KC) {
            const int kc = min(KC, prm.C - c0);
            const int kc8 = (kc + 7) & ~7;
            if (!(prm.debug & 2)) {
                gather_chunk<VF, LDA, PREC_F32, BF>(imgA, prm.feats.sc, tapf, tapw, c0, prm.C, kc8, As, ssA, tid);
                if (!sel.sameAB)
                    gather_chunk<VF, LDA, PREC_F32, BF>(imgBp, sel.mfB.sc, tapf + TP, tapw + TP, c0, prm.C, kc8, Bs, ssB, tid);
            }
            __syncthreads();
            if (!(prm.debug & 1)) mma_chunk_f32(As, Bsrc, kc8, accf, lane, wr, wc);
            __syncthreads();
        }
        publish_norms<VF>(ssA, nrm + 0 * TP, tid);
        if (!sel.sameAB) publish_norms<VF>(ssB, nrm + 1 * TP, tid);
    }

    f32x16 accc[2][2];
    zero_acc(accc);
    {
        float ssA[TP * (KC / VC) / NTHREADS], ssB[TP * (KC / VC) / NTHREADS];
#pragma unroll
        for (int i = 0; i < TP * (KC / VC) / NTHREADS; ++i) { ssA[i] = 0.f; ssB[i] = 0.f; }
        const float* imgA = prm.code.p + (long long)b * prm.code.sn;
        const float* imgBp = sel.mcB.p + (long long)sel.imgB * sel.mcB.sn;
        for (int c0 = 0; c0 < prm.K; c0 += KC) {
            const int kc = min(KC, prm.K - c0);
            const int kc8 = (kc + 7) & ~7;
            if (!(prm.debug & 2)) {
                gather_chunk<VC, LDA, PREC_F32, BC>(imgA, prm.code.sc, tapc, tapw, c0, prm.K, kc8, As, ssA, tid);
                if (!sel.sameAB)
                    gather_chunk<VC, LDA, PREC_F32, BC>(imgBp, sel.mcB.sc, tapc + TP, tapw + TP, c0, prm.K, kc8, Bs, ssB, tid);
            }
            __syncthreads();
            if (!(prm.debug & 1)) mma_chunk_f32(As, Bsrc, kc8, accc, lane, wr, wc);
            __syncthreads();
        }
        publish_norms<VC>(ssA, nrm + 2 * TP, tid);
        if (!sel.sameAB) publish_norms<VC>(ssB, nrm + 3 * TP, tid);
    }
    __syncthreads();   // norms visible; staging area is dead from here on

    const float* nAf = nrm, *nBf = sel.sameAB ? nrm : nrm + TP;
    const float* nAc = nrm + 2 * TP, *nBc = sel.sameAB ? nrm + 2 * TP : nrm + 3 * TP;
    park_tile(accf, Tfd, nAf, nBf, lane, wr, wc);
    park_tile(accc, Tcd, nAc, nBc, lane, wr, wc);
    __syncthreads();
    tile_epilogue<4>(prm, Tfd, Tcd, rowmean, red, p, b, sel.direct);
}

// ============================================================ dense tile kernel (production)
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

__device__ __forceinline__ void park_plain(const f32x16 (&acc)[2][2], float* __restrict__ T, int lane, int wr, int wc)
{
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int col = 64 * wc + 32 * ni + (lane & 31);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 64 * wr + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                T[row * LDT + col] = acc[mi][ni][r];
            }
    }
}

// Result tiles are parked in LDS in the FLAT layout of the outputs, T[a + row * P + col], so that the epilogue
// is a linear sweep: 16-byte LDS reads, 16-byte global stores.  `a` = the output tile's start address / 4 mod 4
// (tiles are P*P floats apart and P*P is odd, so they are only 4-byte aligned): with the same shift in LDS
// both sides of the copy are 16-byte aligned at the same time.
__device__ __forceinline__ void park_flat(const f32x16 (&acc)[2][2], float* __restrict__ T, int P, int lane, int wr, int wc)
{
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int col = 64 * wc + 32 * ni + (lane & 31);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 64 * wr + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < P && col < P) T[row * P + col] = acc[mi][ni][r];
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
        const int row = tid >> 1, half = tid & 1;
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
        }
    }
    __syncthreads();

    const int P2 = P * P;
    const float cmin = prm.cmin, cmax = prm.cmax;
    const float invP = 1.f / (float)P;
    float loss_part = 0.f, clamp_part = 0.f;
    if (!(prm.debug & 8)) {
        const int nvec = (a + P2 + 3) >> 2;
        for (int v = tid; v < nvec; v += NTHREADS) {
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
                w4[k] = w; lp4[k] = lp;
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
        st[0] = (red[0] + red[3]) + (red[6] + red[9]);
        st[1] = (red[1] + red[4]) + (red[7] + red[10]);
        st[2] = (red[2] + red[5]) + (red[8] + red[11]);
        st[3] = 0.f;
    }
}

template <int PREC>
__global__ void __launch_bounds__(NTHREADS) corr_tile_kernel(const CorrParams prm, const int stage_bytes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* rowmean = reinterpret_cast<float*>(smem + SD_ROWMEAN);
    float* red = reinterpret_cast<float*>(smem + SD_RED);
    unsigned char* stage = smem + SD_BIG;
    float* Tfd = reinterpret_cast<float*>(smem + SD_BIG);   // epilogue alias of the stage buffers
    float* Tcd = Tfd + TP * LDT;
    constexpr int FSIDE = PREC == PREC_F32 ? FEAT_SIDE_F32 : FEAT_SIDE_BF16;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    const unsigned char* fsB = static_cast<const unsigned char*>(prm.fs) + (size_t)sB * NCH * FSIDE;
    const unsigned char* csA = reinterpret_cast<const unsigned char*>(prm.cs) + (size_t)sA * cside;
    const unsigned char* csB = reinterpret_cast<const unsigned char*>(prm.cs) + (size_t)sB * cside;

    auto issue = [&](int t) {
        unsigned char* dst = stage + (t & 1) * stage_bytes;
        if (t < NCH) {
            issue_copy(fsA + (size_t)t * FSIDE, dst, FSIDE / 1024, wave, lane);
            if (!sameAB) issue_copy(fsB + (size_t)t * FSIDE, dst + FSIDE, FSIDE / 1024, wave, lane);
        } else {
            issue_copy(csA, dst, cside / 1024, wave, lane);
            if (!sameAB) issue_copy(csB, dst + cside, cside / 1024, wave, lane);
        }
    };

    // debug bit 256: phase stamps of every workgroup on the 100 MHz global clock (tools/stamps.py)
    unsigned long long* ts = reinterpret_cast<unsigned long long*>(prm.stats + (size_t)prm.n_sets * B * 4 + 256) + (size_t)blockIdx.x * 8;
    const bool stamp_on = (prm.debug & 256) && tid == 0;
    if (stamp_on) ts[0] = __builtin_amdgcn_s_memrealtime();
    f32x16 accf[2][2], accc[2][2];
    zero_acc(accf);
    zero_acc(accc);
    const int T = NCH + 1;                       // feature chunks, then the whole code operand pair
    if (!(prm.debug & 2)) issue(0);
    for (int t = 0; t < T; ++t) {
        __syncthreads();                         // (waits vmcnt(0)) chunk t has landed; everyone is done with chunk t-1
        if (t + 1 < T && !(prm.debug & 2)) issue(t + 1);
        if (prm.debug & 1) continue;
        const unsigned char* Ab = stage + (t & 1) * stage_bytes;
        if (t < NCH) {
            const unsigned char* Bb = sameAB ? Ab : Ab + FSIDE;
            const int kc = min(KC, prm.C - t * KC);
            if constexpr (PREC == PREC_F32)
                mma_chunk_f32(reinterpret_cast<const float*>(Ab), reinterpret_cast<const float*>(Bb), (kc + 7) & ~7, accf, lane, wr, wc);
            else
                mma_chunk_bf16x3(reinterpret_cast<const __bf16*>(Ab), reinterpret_cast<const __bf16*>(Bb), (kc + 15) & ~15, accf, lane, wr, wc);
        } else {
            const unsigned char* Bb = sameAB ? Ab : Ab + cside;
            mma_code_f32(reinterpret_cast<const float*>(Ab), reinterpret_cast<const float*>(Bb), prm.KQ, prm.LDK, accc, lane, wr, wc);
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
    __syncthreads();                             // stage buffers are dead: park the result tiles over them
    if (stamp_on) ts[1] = __builtin_amdgcn_s_memrealtime();
    park_flat(accf, Tfd + a, P, lane, wr, wc);
    park_flat(accc, Tcd + a, P, lane, wr, wc);
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
    __shared__ float s_mean;
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
    if (threadIdx.x == 0) {
        float fs = 0.f;
        const int p = first_loss_set + ps;
        for (int b = 0; b < B; ++b) fs += prm.stats[((size_t)p * B + b) * 4];
        s_mean = fs * inv_cnt;
    }
    __syncthreads();
    const float om = s_mean;
    const float cmin = prm.cmin, cmax = prm.cmax;
    const size_t beg = (size_t)(blockIdx.x % blocks_per_set) * span;
    float* loss = prm.neg_loss + (size_t)ps * per_set;
    const float* cd = prm.neg_cd + (size_t)ps * per_set;
    const bool vec_ok = (per_set % 4 == 0) && ((reinterpret_cast<uintptr_t>(loss) & 15) == 0) &&
                        ((reinterpret_cast<uintptr_t>(cd) & 15) == 0);
    if (vec_ok) {
        f32x4 l[4], c[4];
        size_t e[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            e[i] = beg + ((size_t)i * NTHREADS + threadIdx.x) * 4;
            if (e[i] < per_set) {
                l[i] = *reinterpret_cast<const f32x4*>(loss + e[i]);
                c[i] = *reinterpret_cast<const f32x4*>(cd + e[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (e[i] < per_set) {
#pragma unroll
                for (int k = 0; k < 4; ++k) l[i][k] -= om * fminf(fmaxf(c[i][k], cmin), cmax);
                *reinterpret_cast<f32x4*>(loss + e[i]) = l[i];
            }
        }
    } else {
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const size_t e = beg + (size_t)i * NTHREADS + threadIdx.x;
            if (e < per_set) loss[e] -= om * fminf(fmaxf(cd[e], cmin), cmax);
        }
    }
}

// ------------------------------------------------------------------------------ launch
static int pick_vec(const MapV& a, const MapV& b, int channels, bool allow2)
{
    auto ok = [&](const MapV& m, int v) {
        return m.sc == 1 && channels % v == 0 && (m.sn % v) == 0 && (m.sh % v) == 0 && (m.sw % v) == 0 &&
               (reinterpret_cast<uintptr_t>(m.p) % (4 * v)) == 0;
    };
    if (ok(a, 4) && ok(b, 4)) return 4;
    if (allow2 && ok(a, 2) && ok(b, 2)) return 2;
    return 1;
}

template <typename K>
static hipError_t launch_one(K kernel, int lds, bool& attr_done, const CorrParams& prm, int threads, hipStream_t stream)
{
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(kernel, dim3(prm.n_sets * prm.B), dim3(threads), lds, stream, prm);
    return hipGetLastError();
}

#define STEGO_LAUNCH(KERNEL, LDS, THREADS)                                         \
    do {                                                                           \
        static bool done = false;                                                  \
        return launch_one(KERNEL, LDS, done, prm, THREADS, stream);                \
    } while (0)

// bytes of one stage buffer of the dense kernel: max(feature chunk pair, code operand pair)
int dense_stage_bytes(int precision, int LDK)
{
    const int f = 2 * (precision == PREC_F32 ? FEAT_SIDE_F32 : FEAT_SIDE_BF16);
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
    static int attr_f32 = 0, attr_bf16 = 0;
    int& have = precision == PREC_F32 ? attr_f32 : attr_bf16;
    const void* fn = precision == PREC_F32 ? reinterpret_cast<const void*>(&corr_tile_kernel<PREC_F32>)
                                           : reinterpret_cast<const void*>(&corr_tile_kernel<PREC_BF16X3>);
    if (have < lds) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        have = lds;
    }
    const dim3 grid(prm.n_sets * prm.B), block(NTHREADS);
    if (precision == PREC_F32) hipLaunchKernelGGL((corr_tile_kernel<PREC_F32>), grid, block, lds, stream, prm, stage);
    else hipLaunchKernelGGL((corr_tile_kernel<PREC_BF16X3>), grid, block, lds, stream, prm, stage);
    return hipGetLastError();
}

// the fused-gather cross-check kernel (f32 only)
hipError_t launch_corr_fwd_simple(const CorrParams& prm, hipStream_t stream)
{
    const int vf = pick_vec(prm.feats, prm.feats_pos, prm.C, false);
    const int vc = pick_vec(prm.code, prm.code_pos, prm.K, true);
    if (vf == 4 && vc == 4) STEGO_LAUNCH((corr_fwd_kernel<4, 4>), SM_FWD_TOTAL, NTHREADS);
    if (vf == 4 && vc == 2) STEGO_LAUNCH((corr_fwd_kernel<4, 2>), SM_FWD_TOTAL, NTHREADS);
    STEGO_LAUNCH((corr_fwd_kernel<1, 1>), SM_FWD_TOTAL, NTHREADS);
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
