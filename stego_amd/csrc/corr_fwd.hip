// Fused forward of STEGO's ContrastiveCorrelationLoss for gfx950 (MI355X).
//
// One workgroup = one (pair-set p, image b) tile:  A side = image b sampled at coords1[b],
// B side = {same | feats_pos/code_pos[b] @ coords2[b] | feats/code[perm[b]] @ coords2[b]}.
// Per tile:  bilinear gather (channels-last: one contiguous KC*4 B read per tap) -> LDS ->
// fp32 MFMA 32x32x2 for fd = A_f.B_f^T (contraction C) and cd = A_c.B_c^T (contraction K)
// on RAW sampled values; the L2 normalisation is applied in the epilogue as row/col scales
// (fd[i][j] * invn_A[i] * invn_B[j]) - the norms are accumulated during the gather.
// Epilogue: row-centring of fd (the reference's fd -= fd.mean([3,4]), modules.py:332),
// clamp(cd)*(fd-shift), coalesced stores, per-tile partial sums.  The batch-global
// old_mean (modules.py:331) needs every tile of the pair-set, so it is applied by
// corr_finalize_kernel from the per-tile sums (deterministic order, no atomics).
//
// Reference path: src/modules.py:275-398.
#include "corr_common.h"

namespace stego {

// smem carve (bytes): small persistent arrays first, then one big region that is the staging
// area during the contraction and the two 128x129 result tiles during the epilogue.
constexpr int SM_NRM = 0;                         // float nrm[4][128]: Af, Bf, Ac, Bc
constexpr int SM_ROWMEAN = SM_NRM + 4 * TP * 4;   // float rowmean[128]
constexpr int SM_RED = SM_ROWMEAN + TP * 4;       // float red[64]
constexpr int SM_BIG = SM_RED + 64 * 4;           // 2816, 16-byte aligned
constexpr int SM_STAGE_BYTES = 2 * TP * LDA * 4 + 3 * 256 * 16;
constexpr int SM_TILES_BYTES = 2 * TP * LDT * 4;
constexpr int SM_FWD_TOTAL = SM_BIG + (SM_TILES_BYTES > SM_STAGE_BYTES ? SM_TILES_BYTES : SM_STAGE_BYTES);

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

template <int VF, int VC>
__global__ void __launch_bounds__(NTHREADS) corr_fwd_kernel(const CorrParams prm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* nrm = reinterpret_cast<float*>(smem + SM_NRM);
    float* rowmean = reinterpret_cast<float*>(smem + SM_ROWMEAN);
    float* red = reinterpret_cast<float*>(smem + SM_RED);
    float* As = reinterpret_cast<float*>(smem + SM_BIG);
    float* Bs = As + TP * LDA;
    int4* tapf = reinterpret_cast<int4*>(Bs + TP * LDA);   // feature-map tap offsets [256]
    int4* tapc = tapf + 256;                               // code-map tap offsets    [256]
    float4* tapw = reinterpret_cast<float4*>(tapc + 256);  // tap weights             [256]
    float* Tfd = reinterpret_cast<float*>(smem + SM_BIG);  // epilogue alias
    float* Tcd = Tfd + TP * LDT;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int B = prm.B, P = prm.P;
    const int tile = blockIdx.x;
    const int b = tile % B, p = tile / B;      // all pair-sets of image b share blockIdx%8 (XCD L2) when B%8==0

    // ---- which maps feed the two sides (modules.py:369-386)
    // (selected by value: a pointer into the kernarg struct would force it into scratch)
    const bool direct = prm.mode == 1;
    const bool usePos = direct || p == 1;
    const bool sameAB = !direct && p == 0;
    const MapV mfA = prm.feats;
    const MapV mcA = prm.code;
    MapV mfB, mcB;
    mfB.p = usePos ? prm.feats_pos.p : prm.feats.p;     mcB.p = usePos ? prm.code_pos.p : prm.code.p;
    mfB.sn = usePos ? prm.feats_pos.sn : prm.feats.sn;  mcB.sn = usePos ? prm.code_pos.sn : prm.code.sn;
    mfB.sc = usePos ? prm.feats_pos.sc : prm.feats.sc;  mcB.sc = usePos ? prm.code_pos.sc : prm.code.sc;
    mfB.sh = usePos ? prm.feats_pos.sh : prm.feats.sh;  mcB.sh = usePos ? prm.code_pos.sh : prm.code.sh;
    mfB.sw = usePos ? prm.feats_pos.sw : prm.feats.sw;  mcB.sw = usePos ? prm.code_pos.sw : prm.code.sw;
    const float* coordsB = (!direct && p >= 1) ? prm.coords2 : prm.coords1;
    int imgB = b;
    if (!direct && p >= 2) imgB = (int)prm.perms[(size_t)(p - 2) * B + b];

    // ---- tap tables
    {
        const int side = tid >> 7, q = tid & (TP - 1);
        const float* cimg = direct ? nullptr
                                   : (side == 0 ? prm.coords1 + (size_t)b * P * 2 : coordsB + (size_t)b * P * 2);
        int4 yx; float4 w;
        tap_for_point(q, P, prm.S, prm.H, prm.W, direct, cimg, yx, w);
        tapf[tid] = taps_to_offsets(yx, side == 0 ? mfA.sh : mfB.sh, side == 0 ? mfA.sw : mfB.sw);
        tapc[tid] = taps_to_offsets(yx, side == 0 ? mcA.sh : mcB.sh, side == 0 ? mcA.sw : mcB.sw);
        tapw[tid] = w;
    }
    __syncthreads();

    const float* Bsrc = sameAB ? As : Bs;

    // ---- feature contraction: fd_raw = A_f . B_f^T over C
    f32x16 accf[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) accf[i][j][r] = 0.f;
    {
        float ssA[TP * (KC / VF) / NTHREADS], ssB[TP * (KC / VF) / NTHREADS];
#pragma unroll
        for (int i = 0; i < TP * (KC / VF) / NTHREADS; ++i) { ssA[i] = 0.f; ssB[i] = 0.f; }
        const float* imgA = mfA.p + (long long)b * mfA.sn;
        const float* imgBp = mfB.p + (long long)imgB * mfB.sn;
        for (int c0 = 0; c0 < prm.C; c0 += KC) {
            const int kc = min(KC, prm.C - c0);
            const int kc8 = (kc + 7) & ~7;
            gather_chunk<VF, LDA>(imgA, mfA.sc, tapf, tapw, c0, prm.C, kc8, P, As, ssA);
            if (!sameAB) gather_chunk<VF, LDA>(imgBp, mfB.sc, tapf + TP, tapw + TP, c0, prm.C, kc8, P, Bs, ssB);
            __syncthreads();
            mma_chunk_f32(As, Bsrc, kc8, accf, lane, wr, wc);
            __syncthreads();
        }
        publish_norms<VF>(ssA, nrm + 0 * TP);
        if (!sameAB) publish_norms<VF>(ssB, nrm + 1 * TP);
    }

    // ---- code contraction: cd_raw = A_c . B_c^T over K
    f32x16 accc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) accc[i][j][r] = 0.f;
    {
        float ssA[TP * (KC / VC) / NTHREADS], ssB[TP * (KC / VC) / NTHREADS];
#pragma unroll
        for (int i = 0; i < TP * (KC / VC) / NTHREADS; ++i) { ssA[i] = 0.f; ssB[i] = 0.f; }
        const float* imgA = mcA.p + (long long)b * mcA.sn;
        const float* imgBp = mcB.p + (long long)imgB * mcB.sn;
        for (int c0 = 0; c0 < prm.K; c0 += KC) {
            const int kc = min(KC, prm.K - c0);
            const int kc8 = (kc + 7) & ~7;
            gather_chunk<VC, LDA>(imgA, mcA.sc, tapc, tapw, c0, prm.K, kc8, P, As, ssA);
            if (!sameAB) gather_chunk<VC, LDA>(imgBp, mcB.sc, tapc + TP, tapw + TP, c0, prm.K, kc8, P, Bs, ssB);
            __syncthreads();
            mma_chunk_f32(As, Bsrc, kc8, accc, lane, wr, wc);
            __syncthreads();
        }
        publish_norms<VC>(ssA, nrm + 2 * TP);
        if (!sameAB) publish_norms<VC>(ssB, nrm + 3 * TP);
    }
    __syncthreads();   // norms visible; staging area is dead from here on

    const float* nAf = nrm, *nBf = sameAB ? nrm : nrm + TP;
    const float* nAc = nrm + 2 * TP, *nBc = sameAB ? nrm + 2 * TP : nrm + 3 * TP;
    park_tile(accf, Tfd, nAf, nBf, lane, wr, wc);
    park_tile(accc, Tcd, nAc, nBc, lane, wr, wc);
    __syncthreads();

    // ---- row means of fd over the B-side points (fd.mean([3,4]), modules.py:332); two threads per row
    float fdsum_part = 0.f;
    {
        const int r = tid >> 1, hsel = tid & 1;
        const int cbeg = hsel ? (P >> 1) : 0, cend = hsel ? P : (P >> 1);
        float s = 0.f;
        if (r < P)
            for (int c = cbeg; c < cend; ++c) s += Tfd[r * LDT + c];
        s += __shfl_xor(s, 1, 64);
        if (hsel == 0) {
            rowmean[r] = prm.pointwise ? s / (float)P : 0.f;
            fdsum_part = (r < P) ? s : 0.f;
        }
    }
    const float fd_sum = block_sum(fdsum_part, red);   // (syncs inside: rowmean visible after)

    // ---- elementwise epilogue over the P*P outputs, linear (coalesced) order
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
    for (int idx = tid; idx < P2; idx += NTHREADS) {
        const int r = idx / P, c = idx - r * P;
        const float w = Tfd[r * LDT + c] - rowmean[r] - shift;     // fd_centred - shift
        const float cdv = Tcd[r * LDT + c];
        const float cl = fminf(fmaxf(cdv, cmin), cmax);
        const float lp = -cl * w;                                  // loss without the old_mean term
        cd_out[idx] = cdv;
        if (loss_out) loss_out[idx] = lp;
        if (w_out) w_out[idx] = w;
        loss_part += lp;
        clamp_part += cl;
    }
    const float loss_sum = block_sum(loss_part, red);
    const float clamp_sum = block_sum(clamp_part, red);
    if (tid == 0) {
        float* st = prm.stats + ((size_t)p * B + b) * 4;
        st[0] = fd_sum; st[1] = loss_sum; st[2] = clamp_sum; st[3] = 0.f;
    }
}

// Applies the batch-global mean of each pair-set:  old_mean_p = mean_{b,hw,ij} fd  (modules.py:331),
// fd_final = fd_centred + old_mean (:333, the middle fd.mean() is identically 0), hence
//   loss = lp - old_mean * clamp(cd);   mean(loss) = (sum lp - old_mean * sum clamp) / (B*P^2).
// grid.x = blocks over the loss tensor elements of the sets that output one; block 0 also
// writes the scalar means and saved_mean.
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
    const size_t total = (size_t)n_loss_sets * B * P2;
    const size_t per_set = (size_t)B * P2;
    // each block handles a contiguous span that lies inside one set (host sizes the grid so)
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
    (void)total;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const size_t e = beg + (size_t)i * NTHREADS + threadIdx.x;
        if (e < per_set) loss[e] -= om * fminf(fmaxf(cd[e], cmin), cmax);
    }
}

// ------------------------------------------------------------------------------ launch
static int pick_vec(const MapV& a, const MapV& b, int channels)
{
    auto ok = [&](const MapV& m, int v) {
        return m.sc == 1 && channels % v == 0 && (m.sn % v) == 0 && (m.sh % v) == 0 && (m.sw % v) == 0 &&
               (reinterpret_cast<uintptr_t>(m.p) % (4 * v)) == 0;
    };
    if (ok(a, 4) && ok(b, 4)) return 4;
    if (ok(a, 2) && ok(b, 2)) return 2;
    return 1;
}

template <int VF>
static hipError_t launch_fwd_vc(const CorrParams& prm, int vc, dim3 grid, hipStream_t stream)
{
    switch (vc) {
        case 4: hipLaunchKernelGGL((corr_fwd_kernel<VF, 4>), grid, dim3(NTHREADS), SM_FWD_TOTAL, stream, prm); break;
        case 2: hipLaunchKernelGGL((corr_fwd_kernel<VF, 2>), grid, dim3(NTHREADS), SM_FWD_TOTAL, stream, prm); break;
        default: hipLaunchKernelGGL((corr_fwd_kernel<VF, 1>), grid, dim3(NTHREADS), SM_FWD_TOTAL, stream, prm); break;
    }
    return hipGetLastError();
}

template <int VF, int VC>
static hipError_t set_attr()
{
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&corr_fwd_kernel<VF, VC>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, SM_FWD_TOTAL);
}

hipError_t launch_corr_fwd_main(const CorrParams& prm, hipStream_t stream)
{
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e;
        if ((e = set_attr<4, 4>()) != hipSuccess) return e;
        if ((e = set_attr<4, 2>()) != hipSuccess) return e;
        if ((e = set_attr<4, 1>()) != hipSuccess) return e;
        if ((e = set_attr<2, 4>()) != hipSuccess) return e;
        if ((e = set_attr<2, 2>()) != hipSuccess) return e;
        if ((e = set_attr<2, 1>()) != hipSuccess) return e;
        if ((e = set_attr<1, 4>()) != hipSuccess) return e;
        if ((e = set_attr<1, 2>()) != hipSuccess) return e;
        if ((e = set_attr<1, 1>()) != hipSuccess) return e;
        attr_done = true;
    }
    const int vf = pick_vec(prm.feats, prm.feats_pos, prm.C);
    const int vc = pick_vec(prm.code, prm.code_pos, prm.K);
    const dim3 grid(prm.n_sets * prm.B);
    switch (vf) {
        case 4: return launch_fwd_vc<4>(prm, vc, grid, stream);
        case 2: return launch_fwd_vc<2>(prm, vc, grid, stream);
        default: return launch_fwd_vc<1>(prm, vc, grid, stream);
    }
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

hipError_t launch_corr_fwd(const CorrParams& prm, hipStream_t stream)
{
    hipError_t e = launch_corr_fwd_main(prm, stream);
    if (e != hipSuccess) return e;
    return launch_corr_finalize(prm, stream);
}

}  // namespace stego
