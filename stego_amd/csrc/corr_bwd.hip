// Backward of STEGO's ContrastiveCorrelationLoss w.r.t. orig_code / orig_code_pos (gfx950).
//
// Two launches, no global atomics, no memsets:
//
// 1. corr_bwd_tile_kernel - one workgroup per (pair-set p, image b) tile, same tiling as the forward:
//      G[hw][ij]  = dL/dcd = g_cd + g_loss * (-(fd_final - shift)) * 1[cmin <= cd <= cmax]   (clamp/mul backward;
//                   fd_final - shift = saved w + old_mean, the clamp mask rides in the LSB of the saved w)
//      dAn = G . Bn          dBn = G^T . An          (the two bmm adjoints; An/Bn = normalised sampled codes,
//                                                      re-used from the forward's saved context: no re-gather)
//      dA  = (dAn - An <An,dAn>) / ||a||             (F.normalize backward), likewise dB
//    GEMMs on v_mfma_f32_16x16x4_f32 (exact fp32).  Output: per (tile, side) the gradient w.r.t. the RAW
//    sampled codes, DT[tile][side][128][LDK].
//
// 2. corr_unsample_row_kernel - the adjoint of the bilinear sampling (grid_sampler_2d_backward) and of the
//    orig_code[perm] gather (index_put, modules.py:385) written as a GATHER per destination pixel row, accumulated
//    in registers as a sparse GEMM on the f32 MFMA (details at the kernel).  corr_unsample_kernel (rows in an LDS
//    band, worklists, plain LDS read-modify-write) is the general fallback for maps wider than 64 pixels.
//
// The first version of this backward scattered with 15 M global fp32 atomics (93 of its 136 us).
//
// Reference: autograd through src/modules.py:335-347, 369-391.
#include "corr_common.h"
#include "host_util.h"

namespace stego {

constexpr int LDG = 130;   // G row stride (floats)
constexpr int SMB_NRM = 0;                          // float nrm[2][128]
constexpr int SMB_AN = SMB_NRM + 2 * TP * 4;        // float An[128][LDK], then Bn[128][LDK], then G[128][LDG]

// normalize-backward of one side's gradient held in MFMA 16x16 C/D layout and store to DT:
// d[mt][nt][reg] <-> point 32*wave + 16*mt + 4*(lane>>4) + reg, channel 16*nt + (lane&15).
template <int NT>
__device__ __forceinline__ void normalize_bwd_store(f32x4 (&d)[2][NT], const float* __restrict__ Cn, int ldk,
                                                    const float* __restrict__ nrm, float* __restrict__ dt_out,
                                                    int K, int KQ, int lane, int wave)
{
    const int cl = lane & 15, rg = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int pt = 32 * wave + 16 * mt + 4 * rg + reg;
            float cn[NT];
            float dot = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int ch = 16 * nt + cl;
                cn[nt] = ch < K ? Cn[pt * ldk + ch] : 0.f;       // columns >= K hold padding / neighbours
                dot += cn[nt] * (ch < K ? d[mt][nt][reg] : 0.f);
            }
#pragma unroll
            for (int m = 8; m >= 1; m >>= 1) dot += __shfl_xor(dot, m, 64);
            const float nr = nrm[pt];
            const bool big = nr > 1e-10f;
            const float inv = 1.f / fmaxf(nr, 1e-10f);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int ch = 16 * nt + cl;
                if (ch < KQ) {
                    const float dn = d[mt][nt][reg];
                    float v = big ? (dn - cn[nt] * dot) * inv : dn * inv;
                    if (ch >= K) v = 0.f;
                    dt_out[pt * ldk + ch] = v;
                }
            }
        }
    }
}

template <int NT>
__device__ __forceinline__ void bwd_tile_body(const BwdParams& prm, const int tile, unsigned char* smem)
{
    const int ldk = prm.LDK;
    const int cside = TP * ldk * 4;
    float* nrm = reinterpret_cast<float*>(smem + SMB_NRM);
    unsigned char* An_b = smem + SMB_AN;
    unsigned char* Bn_b = An_b + cside;
    float* An = reinterpret_cast<float*>(An_b);
    float* G = reinterpret_cast<float*>(Bn_b + cside);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int B = prm.B, P = prm.P, K = prm.K;
    const int b = tile % B, p = tile / B;
    const bool direct = prm.mode == 1;
    const bool sameAB = !direct && p == 0;
    const int sA = b;
    const int sB = direct ? B + b : (p == 0 ? b : p * B + b);
    const float* Bn = sameAB ? An : reinterpret_cast<const float*>(Bn_b);

    // debug bit 8: phase stamps (100 MHz global clock), 4 per tile, second half of the workspace tail
    unsigned long long* ts = reinterpret_cast<unsigned long long*>(prm.dt + (size_t)prm.n_sets * B * 2 * TP * ldk) + 4096 + (size_t)tile * 4;
    const bool stamp_on = (prm.debug & 8) && tid == 0 && tile < 1024;
    if (stamp_on) ts[0] = __builtin_amdgcn_s_memrealtime();
    // ---- async copies of the normalised sampled codes (saved by the forward) + their norms
    {
        const unsigned char* srcA = reinterpret_cast<const unsigned char*>(prm.cs) + (size_t)sA * cside;
        const unsigned char* srcB = reinterpret_cast<const unsigned char*>(prm.cs) + (size_t)sB * cside;
        const int npieces = cside / 1024;
        for (int pc = wave; pc < npieces; pc += 4) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA + (size_t)pc * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(An_b + pc * 1024), 16, 0, 0);
            if (!sameAB)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcB + (size_t)pc * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void*)(Bn_b + pc * 1024), 16, 0, 0);
        }
        nrm[tid] = prm.nrm[(size_t)(tid < TP ? sA : sB) * TP + (tid & (TP - 1))];
    }

    // ---- G tile: row-wise, branch-free, many loads in flight per lane.
    // Upstreams are folded into (pointer, index multiplier, scale) triples so that absent / broadcast
    // gradients need no branches: g = -(w + old_mean) * gl * 1[cmin<=cd<=cmax] + gc.
    {
        const int P2 = P * P;
        const size_t t0 = direct ? (size_t)b * P2 : (p < 2 ? (size_t)b * P2 : ((size_t)(p - 2) * B + b) * P2);
        const float* wp = prm.saved_w + ((size_t)p * B + b) * P2;
        const float inv_numel = 1.f / ((float)B * (float)P2);
        const float* glp = wp;      // dummy when there is no upstream (scale 0)
        int gl_mul = 0;
        float gl_scale = 0.f;
        if (direct || p >= 2) {
            if (prm.g_neg_loss) {
                const bool dense = direct || prm.g_neg_loss_stride > 0;
                glp = dense ? prm.g_neg_loss + t0 : prm.g_neg_loss;
                gl_mul = dense ? 1 : 0;
                // stride -1: the upstream of torch.cat(neg losses).mean(): one scalar, spread over n_neg B P^2 elements
                gl_scale = (!direct && prm.g_neg_loss_stride < 0) ? inv_numel / (float)prm.n_neg : 1.f;
            }
        } else {
            const float* gs = p == 0 ? prm.g_intra : prm.g_inter;     // .mean() backward (modules.py:393,395)
            if (gs) { glp = gs; gl_scale = inv_numel; }
        }
        const float* gcd = direct ? prm.g_neg_cd : (p == 0 ? prm.g_intra_cd : (p == 1 ? prm.g_inter_cd : prm.g_neg_cd));
        const float* gcp = gcd ? gcd + t0 : wp;
        const float om = prm.saved_mean[p];
        // absent / broadcast upstreams cost no loads (uniform branches around whole batches of loads)
        const bool has_gl = gl_mul != 0, has_gc = gcd != nullptr;
        const float gl_b = has_gl ? 0.f : glp[0] * gl_scale;
        constexpr int RB = 16;
        for (int i0 = 0; i0 < TP / 4; i0 += RB) {
            float wv[RB][2], glv[RB][2], gcv[RB][2];
#pragma unroll
            for (int j = 0; j < RB; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int r = wave + 4 * (i0 + j), c = lane + 64 * h;
                    wv[j][h] = wp[min(r, P - 1) * P + min(c, P - 1)];
                }
            if (has_gl) {
#pragma unroll
                for (int j = 0; j < RB; ++j)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int r = wave + 4 * (i0 + j), c = lane + 64 * h;
                        glv[j][h] = glp[min(r, P - 1) * P + min(c, P - 1)] * gl_scale;
                    }
            }
            if (has_gc) {
#pragma unroll
                for (int j = 0; j < RB; ++j)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int r = wave + 4 * (i0 + j), c = lane + 64 * h;
                        gcv[j][h] = gcp[min(r, P - 1) * P + min(c, P - 1)];
                    }
            }
#pragma unroll
            for (int j = 0; j < RB; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int r = wave + 4 * (i0 + j), c = lane + 64 * h;
                    // the forward left the clamp pass-mask 1[cmin <= cd <= cmax] in the mantissa LSB of w
                    const unsigned wb = __builtin_bit_cast(unsigned, wv[j][h]);
                    const float w = __builtin_bit_cast(float, wb & ~1u);
                    float g = (wb & 1u) ? -(w + om) * (has_gl ? glv[j][h] : gl_b) : 0.f;
                    if (has_gc) g += gcv[j][h];
                    G[r * LDG + c] = (r < P && c < P) ? g : 0.f;
                }
        }
    }
    sync_after_lds_dma();  // An/Bn landed, G complete
    if (stamp_on) ts[1] = __builtin_amdgcn_s_memrealtime();

    // ---- dAn = G . Bn  and  dBn = G^T . An   on v_mfma_f32_16x16x4_f32
    // operand lane map: A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15]
    f32x4 dA[2][NT], dB[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { dA[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f}; dB[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    // dA first, its normalize-backward + stores next (they drain while the second GEMM runs), then dB
    const int cl = lane & 15, kq = lane >> 4;
    const int row0 = 32 * wave + cl;
    const int kend = (prm.debug & 1) ? 0 : TP;
    float* dtA = prm.dt + ((size_t)tile * 2 + 0) * TP * ldk;
    for (int kk = 0; kk < kend; kk += 4) {
        const int k = kk + kq;
        float ga[2], bn[NT];
        ga[0] = G[row0 * LDG + k];                 // G[i][k]
        ga[1] = G[(row0 + 16) * LDG + k];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bn[nt] = Bn[k * ldk + 16 * nt + cl];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            dA[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[0], bn[nt], dA[0][nt], 0, 0, 0);
            dA[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[1], bn[nt], dA[1][nt], 0, 0, 0);
        }
    }
    if (!sameAB) normalize_bwd_store<NT>(dA, An, ldk, nrm, dtA, K, prm.KQ, lane, wave);
    for (int kk = 0; kk < kend; kk += 4) {
        const int k = kk + kq;
        float gt[2], an[NT];
        gt[0] = G[k * LDG + row0];                 // G^T[i][k] = G[k][i]
        gt[1] = G[k * LDG + row0 + 16];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) an[nt] = An[k * ldk + 16 * nt + cl];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            dB[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(gt[0], an[nt], dB[0][nt], 0, 0, 0);
            dB[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(gt[1], an[nt], dB[1][nt], 0, 0, 0);
        }
    }
    if (stamp_on) ts[2] = __builtin_amdgcn_s_memrealtime();
    // ---- normalize backward -> DT[tile][side]
    if (sameAB) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) dA[mt][nt] += dB[mt][nt];   // c1 is c2: both adjoints hit the same samples
        normalize_bwd_store<NT>(dA, An, ldk, nrm, dtA, K, prm.KQ, lane, wave);
    } else {
        normalize_bwd_store<NT>(dB, Bn, ldk, nrm + TP, dtA + (size_t)TP * ldk, K, prm.KQ, lane, wave);
    }
    if (stamp_on) ts[3] = __builtin_amdgcn_s_memrealtime();
}

template <int NT>
__global__ void __launch_bounds__(NTHREADS) corr_bwd_tile_kernel(const BwdParams prm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bwd_tile_body<NT>(prm, (int)blockIdx.x, smem);
}

// ------------------------------------------------------------------------------- tile kernel, split-fp16 GEMMs
// The same backward with the two code GEMMs on the fp16 matrix cores (forward() semantics: precision F16X3, and any mode once
// K > 80; the fp32 MFMA of the kernel above runs at the VALU rate on gfx950: 2 x 4.3 us per tile).  Every fp32 operand is
// split into fp16 hi + lo and a product is hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x16_f16 with fp32 accumulation, as in
// the forward.  MFMA operands must be k-contiguous per lane:
//   CT [side][hi|lo][channel][point]   the normalised sampled codes of both sides, TRANSPOSED on the way into LDS
//   Gh/Gl [row][col]                   G, row-major, times one power of two per tile
//   dAn^T = Bn^T . G^T    A operand = rows of CT(B), B operand = rows of G (k = column: 8-byte reads)
//   dBn^T = An^T . G      A operand = rows of CT(A) (k = anchor point), B operand = COLUMNS of G, gathered as 4 x 2 bytes
//                         per lane and step (a transposed copy of G does not fit next to CT: 157 KB are in use)
// Scale: G's entries are ~1e-7 (upstream 1 / (B P^2)).  One power of two per tile brings the largest |g| (scalar upstreams: the
// bound 4 |gl|; tensor upstreams, DENSE: the exact maximum, one wave reduction + 8 partials through LDS) into [0.5, 1); fp16
// hi + lo then resolve 2^-25 of that absolutely - fp32 accuracy relative to the largest entries, which is what a 128-term sum needs.
// Channel tiles beyond 5 (K > 80) are processed in two groups that share the G image.
constexpr int HB_LDR = 136;                           // halves per row of an operand image (272 B: conflict-free 8-byte reads)
constexpr int SMH_NRM = 0;                            // float nrm[2][128], red[8]
constexpr int SMH_CT = 1280;

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// normalize-backward of the B side's gradient held TRANSPOSED in MFMA 16x16 C/D layout and store to DT:
// d[mc][np][reg] <-> channel 16 mc + 4 (lane >> 4) + reg, point 16 (NP wave + np) + (lane & 15).
constexpr int HW_WAVES = 8;                           // waves of the split kernel (two per SIMD: the phases are latency chains)
constexpr int HW_THREADS = 64 * HW_WAVES;
constexpr int HW_NP = TP / (16 * HW_WAVES);           // 16-point tiles per wave
template <int NT>
__device__ __forceinline__ void normalize_bwd_store_t(f32x4 (&d)[NT][HW_NP], const float* __restrict__ Cn, int ldk,
                                                      const float* __restrict__ nrm, float* __restrict__ dt_out,
                                                      int K, int KQ, int lane, int wave, bool wt = false)
{
    const int cl = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int np = 0; np < HW_NP; ++np) {
        const int pt = 16 * (HW_NP * wave + np) + cl;
        f32x4 cn[NT];
        float dot = 0.f;
#pragma unroll
        for (int mc = 0; mc < NT; ++mc) {
            const int ch0 = 16 * mc + 4 * kq;
            cn[mc] = *reinterpret_cast<const f32x4*>(Cn + (size_t)pt * ldk + ch0);     // columns >= K hold padding / neighbours
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                if (ch0 + reg >= K) cn[mc][reg] = 0.f;
                dot += cn[mc][reg] * (ch0 + reg < K ? d[mc][np][reg] : 0.f);
            }
        }
        dot += __shfl_xor(dot, 16, 64);
        dot += __shfl_xor(dot, 32, 64);
        const float nr = nrm[pt];
        const bool big = nr > 1e-10f;
        const float inv = 1.f / fmaxf(nr, 1e-10f);
#pragma unroll
        for (int mc = 0; mc < NT; ++mc) {
            const int ch0 = 16 * mc + 4 * kq;
            if (ch0 < KQ) {                           // KQ is a multiple of 8: a 4-channel group is inside or outside
                f32x4 v;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const float dn = d[mc][np][reg];
                    v[reg] = big ? (dn - cn[mc][reg] * dot) * inv : dn * inv;
                    if (ch0 + reg >= K) v[reg] = 0.f;
                }
                float* dst = dt_out + (size_t)pt * ldk + ch0;
                // wt: written through (16-byte sc1 store, the cost of a plain one) instead of staying dirty in the L2 until the launch ends
                if (wt) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst), "v"(v) : "memory");
                else *reinterpret_cast<f32x4*>(dst) = v;
            }
        }
    }
}

// One tile of the split-fp16 backward (the body of corr_bwd_tile_h_kernel and of the tile role of corr_bwd_tile_build_kernel).
template <int NT, bool DENSE>      // DENSE: some upstream gradient is a tensor (not the training case)
__device__ __forceinline__ void bwd_tile_h_body(const BwdParams& prm, const int tile, unsigned char* smem)
{
    // channel tiles are processed in groups whose operand images fit LDS next to G: all of them up to K = 80, two groups beyond
    constexpr int NTG = NT <= 5 ? NT : (NT + 1) / 2;
    constexpr int NG = (NT + NTG - 1) / NTG;
    constexpr int CTR = 16 * NTG;                     // channel rows per operand image
    float* nrm = reinterpret_cast<float*>(smem + SMH_NRM);
    float* red = nrm + 2 * TP;                        // [HW_WAVES] partial maxima of |g|
    half_t* CT = reinterpret_cast<half_t*>(smem + SMH_CT);                       // [side][hi|lo][CTR][HB_LDR]
    half_t* Gh = CT + 4 * CTR * HB_LDR;
    half_t* Gl = Gh + TP * HB_LDR;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int B = prm.B, P = prm.P, K = prm.K, ldk = prm.LDK;
    const int b = tile % B, p = tile / B;
    const bool sameAB = p == 0;
    const int sA = b;
    const int sB = p == 0 ? b : p * B + b;
    const float* csA = prm.cs + (size_t)sA * TP * ldk;
    const float* csB = prm.cs + (size_t)sB * TP * ldk;

    // debug bit 8: phase stamps (100 MHz global clock), 4 per tile, second half of the workspace tail
    unsigned long long* ts = reinterpret_cast<unsigned long long*>(prm.dt + (size_t)prm.n_sets * B * 2 * TP * ldk) + 4096 + (size_t)tile * 4;
    const bool stamp_on = (prm.debug & 8) && tid == 0 && tile < 1024;
    if (stamp_on) ts[0] = __builtin_amdgcn_s_memrealtime();

    // ---- upstreams folded into (pointer, index multiplier, scale) triples: absent / broadcast gradients need no branches
    //      g = -(w + old_mean) * gl * 1[cmin <= cd <= cmax] + gc
    const int P2 = P * P;
    const size_t t0 = p < 2 ? (size_t)b * P2 : ((size_t)(p - 2) * B + b) * P2;
    const float* wp = prm.saved_w + ((size_t)p * B + b) * P2;
    const float* glp = wp;                            // dummy when there is no upstream (scale 0)
    bool has_gl = false;
    float gl_b = 0.f;                                 // broadcast / scalar upstream of the loss
    if (p >= 2) {
        if (prm.g_neg_loss) {
            if (DENSE && prm.g_neg_loss_stride > 0) { glp = prm.g_neg_loss + t0; has_gl = true; }
            else if (prm.g_neg_loss_stride < 0) gl_b = prm.g_neg_loss[0] * (1.f / ((float)B * (float)P2 * (float)prm.n_neg));   // mean upstream
            else gl_b = prm.g_neg_loss[0];
        }
    } else {
        const float* gs = p == 0 ? prm.g_intra : prm.g_inter;     // .mean() backward (modules.py:393,395)
        if (gs) gl_b = gs[0] * (1.f / ((float)B * (float)P2));
    }
    const float* gcd = !DENSE ? nullptr : (p == 0 ? prm.g_intra_cd : (p == 1 ? prm.g_inter_cd : prm.g_neg_cd));
    const bool has_gc = gcd != nullptr;
    const float* gcp = has_gc ? gcd + t0 : wp;
    const float om = prm.saved_mean[p];

    // ---- every global load of the first operands goes out first (one round trip: a wave has 512 registers here): the saved
    // w (and dense upstreams) of my 16 rows of G, and my share of the first group's code rows of both sides
    constexpr int RB = TP / HW_WAVES;                 // rows of G per wave
    constexpr int c4n = CTR / 4;                      // 4-channel groups per row of a group (reads past ldk stay inside the slack)
    constexpr int NIT = ((TP / 2) * c4n + HW_THREADS - 1) / HW_THREADS;
    float wv[RB][2], glv[RB][2], gcv[RB][2];
    f32x4 r0[2][NIT], r1[2][NIT];
    auto load_codes = [&](int grp) {
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const float* src = ((side == 0 || sameAB) ? csA : csB) + grp * CTR;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int i = (prm.debug & 16) ? 0 : min(tid + it * HW_THREADS, (TP / 2) * c4n - 1);    // (ablation 16: one element)
                const int pp = i / c4n, c4 = i - pp * c4n;
                r0[side][it] = *reinterpret_cast<const f32x4*>(src + (size_t)(2 * pp) * ldk + 4 * c4);
                r1[side][it] = *reinterpret_cast<const f32x4*>(src + (size_t)(2 * pp + 1) * ldk + 4 * c4);
            }
        }
    };
    // [point][channel] fp32 rows -> [channel][point] fp16 hi / lo images (two points per store)
    auto convert_codes = [&](int grp) {
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            if (side == 1 && sameAB) break;
            half_t* dh = CT + (size_t)(2 * side) * CTR * HB_LDR;
            half_t* dl = dh + CTR * HB_LDR;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int i = tid + it * HW_THREADS;
                if (i < (TP / 2) * c4n) {
                    const int pp = i / c4n, c4 = i - pp * c4n;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int ch = 4 * c4 + e;
                        const bool in = grp * CTR + ch < K;
                        unsigned h, l;
                        split_f16_pair(in ? r0[side][it][e] : 0.f, in ? r1[side][it][e] : 0.f, h, l);
                        // (4-point groups of a channel row XOR-swizzled by ch >> 4: the 20 channel groups of a wave's store
                        // otherwise fall into 4 banks - rows are 68 dwords apart - a 5-way conflict on each of the 48 stores per lane;
                        // ch >> 4 is the MFMA's channel tile mc, so the fragment reads below stay conflict-free)
                        const int col = (((pp >> 1) ^ (ch >> 4)) << 2) + 2 * (pp & 1);
                        *reinterpret_cast<unsigned*>(dh + ch * HB_LDR + col) = h;
                        *reinterpret_cast<unsigned*>(dl + ch * HB_LDR + col) = l;
                    }
                }
            }
        }
    };
    {
        const int c0 = min(2 * lane, P - 1), c1 = min(2 * lane + 1, P - 1);
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            const int r = (prm.debug & 4) ? 0 : min(wave + HW_WAVES * j, P - 1);       // (ablation 4: one row)
            wv[j][0] = wp[r * P + c0];
            wv[j][1] = wp[r * P + c1];
        }
        if (has_gl) {
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                const int r = min(wave + HW_WAVES * j, P - 1);
                glv[j][0] = glp[r * P + c0];
                glv[j][1] = glp[r * P + c1];
            }
        }
        if (has_gc) {
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                const int r = min(wave + HW_WAVES * j, P - 1);
                gcv[j][0] = gcp[r * P + c0];
                gcv[j][1] = gcp[r * P + c1];
            }
        }
        load_codes(0);
    }
    if (tid < 2 * TP) nrm[tid] = prm.nrm[(size_t)(tid < TP ? sA : sB) * TP + (tid & (TP - 1))];
    // ---- G (rows r = wave + HW_WAVES j, columns 2 lane, 2 lane + 1) in registers; its largest entry picks the tile's scale
    float gmax = 0.f;
#pragma unroll
    for (int j = 0; j < RB; ++j) {
        const int r = wave + HW_WAVES * j;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // the forward left the clamp pass-mask 1[cmin <= cd <= cmax] in the mantissa LSB of w
            const unsigned wb = __builtin_bit_cast(unsigned, wv[j][h]);
            const float w = __builtin_bit_cast(float, wb & ~1u);
            float g = (wb & 1u) ? -(w + om) * (has_gl ? glv[j][h] : gl_b) : 0.f;
            if (has_gc) g += gcv[j][h];
            g = (r < P && 2 * lane + h < P) ? g : 0.f;
            wv[j][h] = g;
            gmax = fmaxf(gmax, fabsf(g));
        }
    }
    float tmax;
    if constexpr (DENSE) {
        // the exact largest |g| of the tile: one wave reduction + 8 partials through LDS
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, m, 64));
        if (lane == 0) red[wave] = gmax;
        __syncthreads();
        tmax = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < HW_WAVES; ++w2) tmax = fmaxf(tmax, red[w2]);
    } else {
        // scalar upstream: |g| = |gl| |w + old_mean| < 4 |gl| (w = a cosine - a row mean - a shift): a bound is as good as the
        // maximum here (it only has to keep |g| s below fp16's range and within a few bits of 1), and costs no barrier
        tmax = 4.f * fabsf(gl_b);
    }
    // one power of two per tile brings the largest |g| into [0.5, 1): fp16 hi + lo then resolve 2^-25 of it absolutely
    float sc_tile = 1.f, inv_tile = 1.f;
    if (tmax > 0.f && tmax < 3.0e38f) {
        const int e = __builtin_amdgcn_frexp_expf(tmax);
        sc_tile = __builtin_ldexpf(1.f, -e);
        inv_tile = __builtin_ldexpf(1.f, e);
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) {
        const int r = wave + HW_WAVES * j;
        unsigned hi, lo;
        split_f16_pair(wv[j][0] * sc_tile, wv[j][1] * sc_tile, hi, lo);
        *reinterpret_cast<unsigned*>(Gh + r * HB_LDR + 2 * lane) = hi;
        *reinterpret_cast<unsigned*>(Gl + r * HB_LDR + 2 * lane) = lo;
    }

    // lane map of v_mfma_f32_16x16x16_f16: A[i = lane & 15][k = 4 (lane >> 4) + 0..3], B[k = 4 (lane >> 4) + 0..3][j = lane & 15]
    const int cl = lane & 15, kq = lane >> 4;
    half_t* ATh = CT;                                 // side 0 = anchors
    half_t* ATl = CT + CTR * HB_LDR;
    half_t* BTh = sameAB ? ATh : CT + 2 * CTR * HB_LDR;
    half_t* BTl = sameAB ? ATl : CT + 3 * CTR * HB_LDR;
    float* dtA = prm.dt + ((size_t)tile * 2 + 0) * TP * ldk;
    const bool skip = (prm.debug & 1) || tmax == 0.f; // (no upstream: G = 0, the gradients are zero)

    const bool early_dB = NG == 1 && !sameAB && !skip && !(prm.debug & 2048);      // (debug 2048: both sides at the end, plain)
    f32x4 dAt[NT][HW_NP], dBt[NT][HW_NP];
#pragma unroll
    for (int mc = 0; mc < NT; ++mc)
#pragma unroll
        for (int np = 0; np < HW_NP; ++np) { dAt[mc][np] = f32x4{0.f, 0.f, 0.f, 0.f}; dBt[mc][np] = f32x4{0.f, 0.f, 0.f, 0.f}; }

#pragma unroll
    for (int grp = 0; grp < NG; ++grp) {
        if (grp > 0) {
            __syncthreads();                          // everyone is done with the previous group's images
            load_codes(grp);
        }
        convert_codes(grp);
        __syncthreads();                              // CT of this group (and, the first time, G) complete
        if (stamp_on && grp == 0) ts[1] = __builtin_amdgcn_s_memrealtime();
        if (skip) continue;
        auto gemm_dA = [&]() {
            // ---- dAn^T = Bn^T . G^T: channel rows of CT(B) (A operand) against the ROWS 16 (HW_NP wave + np) + j of G (B operand:
            // B[k][j] = G[j][k], k-contiguous in a row-major G)
            {
                const int ra = cl * HB_LDR + 4 * kq;
                const int rg = (16 * HW_NP * wave + cl) * HB_LDR + 4 * kq;
#pragma unroll 2
                for (int kk = 0; kk < TP; kk += 16) {
                    f16x4 ah[NTG], al[NTG], bh[HW_NP], bl[HW_NP];
#pragma unroll
                    for (int mc = 0; mc < NTG; ++mc) {
                        const int ks = ((((kk >> 2) + kq) ^ mc) << 2) - 4 * kq;          // swizzled 4-point group (see convert_codes)
                        ah[mc] = *reinterpret_cast<const f16x4*>(BTh + ra + 16 * mc * HB_LDR + ks);
                        al[mc] = *reinterpret_cast<const f16x4*>(BTl + ra + 16 * mc * HB_LDR + ks);
                    }
#pragma unroll
                    for (int np = 0; np < HW_NP; ++np) {
                        bh[np] = *reinterpret_cast<const f16x4*>(Gh + rg + 16 * np * HB_LDR + kk);
                        bl[np] = *reinterpret_cast<const f16x4*>(Gl + rg + 16 * np * HB_LDR + kk);
                    }
                    // the three terms outermost: consecutive MFMAs never wait for each other's accumulator
#pragma unroll
                    for (int term = 0; term < 3; ++term)
#pragma unroll
                        for (int mc = 0; mc < NTG; ++mc)
#pragma unroll
                            for (int np = 0; np < HW_NP; ++np)
                                if (grp * NTG + mc < NT)
                                    dAt[grp * NTG + mc][np] = __builtin_amdgcn_mfma_f32_16x16x16f16(
                                        term == 2 ? al[mc] : ah[mc], term == 1 ? bl[np] : bh[np], dAt[grp * NTG + mc][np], 0, 0, 0);
                }
            }
        };
        auto gemm_dB = [&]() {
            // ---- dBn^T = An^T . G: channel rows of CT(A) against the COLUMNS 16 (HW_NP wave + np) + j of G (k = anchor point),
            // gathered as 4 x 2 bytes per lane and step
            {
                const int ra = cl * HB_LDR + 4 * kq;                          // CT(A): channel cl (+16 mc), k = kk + 4 kq ..
                // G: row kk + 4 kq + e, column 16 (HW_NP wave + np) + cl - four ROWS of one column per lane: the LDS transposes
                // (ds_read_b64_tr_b16: lane p of a 16-lane group passes the address of row p >> 2, columns 4 (p & 3) .. + 3 of a
                // [4 rows][16 columns] block and receives column p, rows 0..3; one read per fragment instead of four 2-byte reads)
                const int gt = (4 * kq + (cl >> 2)) * HB_LDR + 16 * HW_NP * wave + 4 * (cl & 3);
                typedef __fp16 fp16x4v __attribute__((__vector_size__(4 * sizeof(__fp16))));
                typedef __attribute__((address_space(3))) fp16x4v* lds_tr_ptr;
#pragma unroll 2
                for (int kk = 0; kk < TP; kk += 16) {
                    f16x4 ah[NTG], al[NTG], bh[HW_NP], bl[HW_NP];
#pragma unroll
                    for (int mc = 0; mc < NTG; ++mc) {
                        const int ks = ((((kk >> 2) + kq) ^ mc) << 2) - 4 * kq;
                        ah[mc] = *reinterpret_cast<const f16x4*>(ATh + ra + 16 * mc * HB_LDR + ks);
                        al[mc] = *reinterpret_cast<const f16x4*>(ATl + ra + 16 * mc * HB_LDR + ks);
                    }
#pragma unroll
                    for (int np = 0; np < HW_NP; ++np) {
                        const fp16x4v th = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_tr_ptr)(Gh + gt + kk * HB_LDR + 16 * np));
                        const fp16x4v tl = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_tr_ptr)(Gl + gt + kk * HB_LDR + 16 * np));
                        __builtin_memcpy(&bh[np], &th, 8);
                        __builtin_memcpy(&bl[np], &tl, 8);
                    }
#pragma unroll
                    for (int term = 0; term < 3; ++term)
#pragma unroll
                        for (int mc = 0; mc < NTG; ++mc)
#pragma unroll
                            for (int np = 0; np < HW_NP; ++np)
                                if (grp * NTG + mc < NT)
                                    dBt[grp * NTG + mc][np] = __builtin_amdgcn_mfma_f32_16x16x16f16(
                                        term == 2 ? al[mc] : ah[mc], term == 1 ? bl[np] : bh[np], dBt[grp * NTG + mc][np], 0, 0, 0);
                }
            }
        };
        if (early_dB) {
            // The B side first: its rows leave (normalised, written through) while the A side is multiplied, so half of the DT bytes are
            // out of the L2 before the launch ends - what is still dirty then is written back behind the last workgroup, in front of the
            // unsample launch.  The A-side rows stay plain: the unsample units of this anchor's image run on this XCD and hit them in L2.
            gemm_dB();
#pragma unroll
            for (int mc = 0; mc < NT; ++mc)
#pragma unroll
                for (int np = 0; np < HW_NP; ++np) dBt[mc][np] *= inv_tile;
            normalize_bwd_store_t<NT>(dBt, csB, ldk, nrm + TP, dtA + (size_t)TP * ldk, K, prm.KQ, lane, wave, true);
            gemm_dA();
        } else {
            gemm_dA();
            gemm_dB();
        }
    }
    if (stamp_on) ts[2] = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int mc = 0; mc < NT; ++mc)
#pragma unroll
        for (int np = 0; np < HW_NP; ++np) { dAt[mc][np] *= inv_tile; if (!early_dB) dBt[mc][np] *= inv_tile; }
    if (sameAB) {
        // c1 is c2: both adjoints hit the same samples (same register layout: just add)
#pragma unroll
        for (int mc = 0; mc < NT; ++mc)
#pragma unroll
            for (int np = 0; np < HW_NP; ++np) dAt[mc][np] += dBt[mc][np];
        normalize_bwd_store_t<NT>(dAt, csA, ldk, nrm, dtA, K, prm.KQ, lane, wave);
    } else {
        normalize_bwd_store_t<NT>(dAt, csA, ldk, nrm, dtA, K, prm.KQ, lane, wave);
        if (!early_dB) normalize_bwd_store_t<NT>(dBt, csB, ldk, nrm + TP, dtA + (size_t)TP * ldk, K, prm.KQ, lane, wave);
    }
    if (stamp_on) ts[3] = __builtin_amdgcn_s_memrealtime();
}

template <int NT, bool DENSE>
__global__ void __launch_bounds__(HW_THREADS) corr_bwd_tile_h_kernel(const BwdParams prm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bwd_tile_h_body<NT, DENSE>(prm, (int)blockIdx.x, smem);
}

// ------------------------------------------------------------------------------- unsample
// grid = n_dest(2) * B * n_bands ; block = 512 (8 waves); a band has RT <= 8 pixel rows and WAVE r OWNS ROW r.
// LDS: float acc[RT][W][K] | contribution table | per-row worklists | counters.
// Per round, 4 contributions x 128 points are tested lane-parallel; every tap row that falls in the band
// appends a 16-byte row-entry {DT row, x0|x1, w0, w1} to the list of that pixel row.  The owning wave then
// drains its list: 16 DT rows in flight per lane, plain LDS read-modify-write (LDS fp32 atomics measured
// ~1000 cycles per wave-instruction on gfx950, so they are used only on worklist overflow).
constexpr int UNS_THREADS = 512;
constexpr int UNS_SETS_PER_ROUND = UNS_THREADS / TP;      // 4
constexpr int UNS_MAX_RT = 8;
constexpr int UNS_ROW_CAP = 192;                          // row-entries per pixel row per round (expected ~35)
constexpr int UNS_MAX_CONTRIB = 1024;

struct UnsRowEntry {        // 16 bytes
    int dtoff;              // float offset of the DT row
    int x01;                // x0 | x1 << 16
    float wa, wb;           // tap weights at (row, x0), (row, x1)
};

__global__ void __launch_bounds__(UNS_THREADS) corr_unsample_kernel(const BwdParams prm, const int RT, const int n_bands,
                                                                   const int acc_bytes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* acc = reinterpret_cast<float*>(smem);
    int* ctab = reinterpret_cast<int*>(smem + acc_bytes);                               // [UNS_MAX_CONTRIB]
    UnsRowEntry* wl = reinterpret_cast<UnsRowEntry*>(smem + acc_bytes + UNS_MAX_CONTRIB * 4);   // [UNS_MAX_RT][UNS_ROW_CAP]
    int* cnt = reinterpret_cast<int*>(smem + acc_bytes + UNS_MAX_CONTRIB * 4 + UNS_MAX_RT * UNS_ROW_CAP * (int)sizeof(UnsRowEntry));
    int& s_nc = cnt[UNS_MAX_RT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int B = prm.B, P = prm.P, K = prm.K, W = prm.W, H = prm.H, ldk = prm.LDK;
    const bool direct = prm.mode == 1;
    int bid = blockIdx.x;
    const int band = bid % n_bands; bid /= n_bands;
    const int j = bid % B;
    const int dest = bid / B;                        // 0: d_code (d_c1), 1: d_code_pos (d_c2)
    const int r0 = band * RT, r1 = min(H, r0 + RT);
    const int band_elems = (r1 - r0) * W * K;
    const int side_elems = TP * ldk;
    for (int e = tid; e < band_elems; e += UNS_THREADS) acc[e] = 0.f;

    // ---- contribution table: every contribution is (taps of sample set s, DT matrix (tile, side)).
    //   own role of image j, entries [0, n_own):
    //       dest 0: anchor set j with the A side of EVERY pair-set tile (p, j)      -> n_own = n_sets
    //       dest 1: positive set B+j with the B side of tile (1, j); helper: set dest*B+j, tile j, side dest
    //   then (dest 0 only) the negative sets (2+i)*B+b with perm_i[b] == j, B side of their own tile
    //   (the index_put of orig_code[perm], modules.py:385).   ctab[c] holds the set id of entry c >= n_own.
    const int n_own = (!direct && dest == 0) ? prm.n_sets : 1;
    if (wave == 0) {
        int nc = n_own;
        if (!direct && dest == 0) {
            for (int i = 0; i < prm.n_neg; ++i)
                for (int b0 = 0; b0 < B; b0 += 64) {
                    const int bb = b0 + lane;
                    const bool m = bb < B && (int)prm.perms[(size_t)i * B + bb] == j;
                    const unsigned long long mask = __ballot(m);
                    const int pre = __builtin_popcountll(mask & ((1ull << lane) - 1ull));
                    if (m && nc + pre < UNS_MAX_CONTRIB) ctab[nc + pre] = (2 + i) * B + bb;
                    nc += __builtin_popcountll(mask);
                }
            if (nc > UNS_MAX_CONTRIB) nc = UNS_MAX_CONTRIB;        // (cannot happen: host checks n_sets + n_neg*B <= 1024)
        }
        if (lane == 0) s_nc = nc;
    }
    __syncthreads();
    const int NC = s_nc;
    

    for (int c0 = 0; c0 < NC; c0 += UNS_SETS_PER_ROUND) {
        if (tid < UNS_MAX_RT) cnt[tid] = 0;
        __syncthreads();
        // ---- phase 1: thread -> (contribution c0 + tid/128, point tid%128): band test, append row-entries
        {
            const int c = c0 + (tid >> 7), q = tid & (TP - 1);
            if (c < NC && q < P && !(prm.debug & 2)) {
                int s, tile0, side;
                if (c < n_own) {
                    if (direct) { s = dest * B + j; tile0 = j; side = dest; }
                    else if (dest == 1) { s = B + j; tile0 = B + j; side = 1; }
                    else { s = j; tile0 = c * B + j; side = 0; }
                } else { s = ctab[c]; tile0 = s; side = 1; }
                const int4 yx = prm.tapyx[(size_t)s * TP + q];
                const float4 w = prm.tapw[(size_t)s * TP + q];
                const int dtoff = (tile0 * 2 + side) * side_elems + q * ldk;
#pragma unroll
                for (int hrow = 0; hrow < 2; ++hrow) {
                    const int y = (hrow ? yx.z : yx.x) >> 16;
                    const float wa = hrow ? w.z : w.x, wb = hrow ? w.w : w.y;
                    if (y < r0 || y >= r1 || (wa == 0.f && wb == 0.f)) continue;
                    if (hrow == 1 && y == (yx.x >> 16)) continue;          // clamped second row: weights are 0 anyway
                    const int xa = (hrow ? yx.z : yx.x) & 0xffff, xb = (hrow ? yx.w : yx.y) & 0xffff;
                    const int row = y - r0;
                    const int slot = atomicAdd(&cnt[row], 1);
                    if (slot < UNS_ROW_CAP) {
                        UnsRowEntry e;
                        e.dtoff = dtoff; e.x01 = xa | (xb << 16); e.wa = wa; e.wb = wb;
                        wl[row * UNS_ROW_CAP + slot] = e;
                    } else {
                        // overflow (pathological coords, e.g. every point on one row): slow but correct path
                        for (int ch = 0; ch < K; ++ch) {
                            const float v = prm.dt[dtoff + ch];
                            if (wa != 0.f) atomicAdd(&acc[(row * W + xa) * K + ch], wa * v);
                            if (wb != 0.f) atomicAdd(&acc[(row * W + xb) * K + ch], wb * v);
                        }
                    }
                }
            }
        }
        __syncthreads();
        // ---- phase 2: wave r drains the list of pixel row r (lanes = channels, second pass for channels >= 64)
        if (wave < r1 - r0) {
            const int count = min(cnt[wave], UNS_ROW_CAP);
            const UnsRowEntry* list = wl + wave * UNS_ROW_CAP;
            float* arow = acc + wave * W * K;
            constexpr int UB = 16;
            // branch-free: lanes without a channel (and clamped second taps) are redirected to a per-lane
            // dummy cell behind the band, so every load / read-modify-write is unconditional
            const int cA = min(lane, K - 1), cB = min(64 + lane, K - 1);
            const bool hasA = lane < K, hasB = 64 + lane < K;
            float* dummy = acc + (acc_bytes >> 2) - 4 * 64 + lane;
            for (int e0 = 0; e0 < count; e0 += UB) {
                float v0[UB], v1[UB], wa[UB], wb[UB];
                int x01[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const UnsRowEntry en = list[min(e0 + u, count - 1)];
                    const bool on = e0 + u < count;
                    x01[u] = en.x01; wa[u] = on ? en.wa : 0.f; wb[u] = on ? en.wb : 0.f;
                    v0[u] = prm.dt[en.dtoff + cA];
                    v1[u] = prm.dt[en.dtoff + cB];
                }
                if (!(prm.debug & 16)) {
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        // the (up to) 4 cells of one row-entry are distinct: read all, then write all
                        const int xa = x01[u] & 0xffff, xb = x01[u] >> 16;
                        const bool two = xb != xa;                 // clamped tap (xb == xa) carries weight 0
                        float* pa0 = hasA ? arow + xa * K + lane : dummy;
                        float* pb0 = (hasA && two) ? arow + xb * K + lane : dummy + 64;
                        float* pa1 = hasB ? arow + xa * K + 64 + lane : dummy + 128;
                        float* pb1 = (hasB && two) ? arow + xb * K + 64 + lane : dummy + 192;
                        const float a0 = *pa0, b0 = *pb0, a1 = *pa1, b1 = *pb1;
                        *pa0 = a0 + wa[u] * v0[u];
                        *pb0 = b0 + wb[u] * v0[u];
                        *pa1 = a1 + wa[u] * v1[u];
                        *pb1 = b1 + wb[u] * v1[u];
                    }
                }
            }
        }
        __syncthreads();
    }
    float* out = (dest == 0 ? prm.d_code : prm.d_code_pos) + ((size_t)j * H + r0) * W * K;
    for (int e = tid; e < band_elems; e += UNS_THREADS) out[e] = acc[e];
}


// ------------------------------------------------------------------------------- unsample, row-owned
// The fast path (W <= 64): ONE PIXEL ROW of one destination image is owned by a wave (or by the 3 waves of a
// workgroup) and lives in REGISTERS as MFMA accumulators.  The bilinear adjoint of a row is a tiny sparse GEMM
//        row[x][ch] += sum_e  onehot_e[x] * DT_e[ch],      onehot_e[x] = wa_e (x == x0_e), wb_e (x == x0_e+1), 0
// and it is issued exactly like that on v_mfma_f32_16x16x4_f32 (M = 16 pixels, N = 16 channels, K = 4
// row-entries): no dynamic register indexing, no branches, no LDS read-modify-write chain (the LDS-band
// version measured ~500 cycles per entry; a wave-uniform switch over register accumulators made the
// structurizer emit 15 k accumulator copies).  No atomics, deterministic summation order.
//
// The kernel is bound by DEPENDENT LOAD ROUND TRIPS (~2 us each with 4 k waves in flight, measured with
// s_memrealtime stamps), neither by bytes nor by flops, so it is shaped to need three of them:
//   1. perms  -> items : the DT matrices that touch this image (anchor role: A side of every pair-set tile of
//                        the image; negatives: the (i,b) with perm_i[b] == image, found with ballots)
//   2. scan           : the (y0 << 16 | x0) word of every point of <= 6 items; points whose upper or lower
//                        tap row is MY row are compacted (ballot + mbcnt) into a 4-byte worklist in LDS
//   3. drain          : 40 entries per step; the 16 lanes of entry k = lane / 16 load its DT row (B operand,
//                        5 dwords per lane) AND its 32-byte tap record (one dword per lane, spread to the group
//                        with ds_bpermute) in the same round; the A operand is the one-hot weight of pixel lane % 16
// and every row has to be in flight at once (register budget: 128 VGPRs -> 4 waves per SIMD -> 4096 waves).
// Rows of the anchor image (forward mode, dest 0) collect ~12 items, the others one: a HEAVY row gets a
// workgroup whose 3 waves take every 3rd item and reduce-scatter their partial rows through LDS at the end;
// light rows are one wave each.
constexpr int UR_WAVES = 3;
constexpr int UR_GROUP = 6;          // items scanned per step; a point hits a row at most once -> <= 6*128 entries
constexpr int UR_CAP = UR_GROUP * TP;
constexpr int UR_ITEMCAP = 96;       // a pass looks at 256 candidates; only heavy rows have > 1, split over 3 waves
// UR_NT (template): 16-channel tiles, 5 for K <= 80, 8 for K <= 128.  UR_EG: groups of 4 entries per drain step
// (NT = 5: 9 groups = 54 loads in flight per lane; NT = 8: 5 groups = 45)

__device__ __forceinline__ int lane_prefix(unsigned long long mask)
{
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

template <int MT, int UR_NT>        // 16-pixel tiles per row: W <= 16 * MT; 16-channel tiles: K <= 16 * UR_NT
__global__ void __launch_bounds__(UR_WAVES * 64, (MT <= 2 && UR_NT <= 5) ? 4 : 2) corr_unsample_row_kernel(const BwdParams prm)
{
    constexpr int UR_EG = UR_NT <= 5 ? 9 : 5;
    constexpr int RED_BYTES = UR_WAVES * UR_NT * 64 * (int)sizeof(f32x4);          // reduction buffer, aliases the worklists
    constexpr int WL_BYTES = UR_WAVES * MT * UR_CAP * 4;      // one worklist per 16-pixel tile
    __shared__ __attribute__((aligned(16))) unsigned char wl_raw[RED_BYTES > WL_BYTES ? RED_BYTES : WL_BYTES];
    __shared__ __attribute__((aligned(8))) int2 items_s[UR_WAVES][UR_ITEMCAP];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int B = prm.B, P = prm.P, K = prm.K, W = prm.W, H = prm.H, ldk = prm.LDK;
    const bool direct = prm.mode == 1;
    const int n_heavy = direct ? 0 : B * H;                 // units are (dest, image j, row r); heavy dest 0 first
    const bool heavy = (int)blockIdx.x < n_heavy;
    const int unit = heavy ? (int)blockIdx.x : n_heavy + ((int)blockIdx.x - n_heavy) * UR_WAVES + wave;
    const int sub = heavy ? wave : 0, nsub = heavy ? UR_WAVES : 1;
    if (unit >= 2 * B * H) return;
    const int r = unit % H;
    const int j = (unit / H) % B;
    const int dest = unit / (H * B);
    int* wl = reinterpret_cast<int*>(wl_raw) + wave * MT * UR_CAP;
    int2* items = items_s[wave];
    const int side_elems = TP * ldk;
    const int n_own = (!direct && dest == 0) ? prm.n_sets : 1;
    const int n_cand = n_own + ((!direct && dest == 0) ? prm.n_neg * B : 0);
    const int l16 = lane & 15, k4 = lane >> 4;
    int cch[UR_NT];                                          // channel of this lane in each N tile, kept inside the DT row
#pragma unroll
    for (int nt = 0; nt < UR_NT; ++nt) cch[nt] = min(16 * nt + l16, ldk - 1);
    // word of the 32-byte tap record this lane fetches for its entry: lanes 0-3 the pixel words, 4-7 the weights
    // (32-bit offsets from uniform bases: one address VGPR per load instead of two)
    const int* tap_words = reinterpret_cast<const int*>(prm.tapyx);
    const unsigned meta_off = l16 < 4 ? (unsigned)l16
                                      : (unsigned)(reinterpret_cast<const int*>(prm.tapw) - tap_words) + (unsigned)(l16 & 3);
    const int grp_lane0 = lane & 48;

    f32x4 acc[MT][UR_NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < UR_NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // debug bit 8: start / end of every workgroup on the 100 MHz global clock (tools/stamps_bwd.py)
    unsigned long long* ts = reinterpret_cast<unsigned long long*>(prm.dt + (size_t)prm.n_sets * B * 2 * side_elems);
    const bool stamp_on = (prm.debug & 8) && threadIdx.x == 0 && blockIdx.x < 4000;
    if (stamp_on) ts[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();

    int n_seen = 0;
    static_assert(UR_ITEMCAP * UR_WAVES >= 256 && UR_ITEMCAP <= 128, "item table vs candidate pass / 7-bit item index");
    for (int u0 = 0; u0 < n_cand; u0 += 256) {
        // ---- item table from the next 256 candidates (loads issued together)
        int n_items = 0;
        const int n_before = (n_seen + nsub - 1 - sub) / nsub;      // items this wave took in earlier passes
        {
            long long pv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int t = u0 + k * 64 + lane - n_own;
                pv[k] = (t >= 0 && t < n_cand - n_own) ? prm.perms[t] : -1;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int u = u0 + k * 64 + lane;
                bool m = false;
                int s = 0, tile0 = 0, side = 0;
                if (u < n_own) {
                    m = true;
                    if (direct) { s = dest * B + j; tile0 = j; side = dest; }
                    else if (dest == 1) { s = B + j; tile0 = B + j; side = 1; }
                    else { s = j; tile0 = u * B + j; side = 0; }
                } else if (u < n_cand) {
                    m = (int)pv[k] == j;                     // the index_put of orig_code[perm], modules.py:385
                    s = 2 * B + (u - n_own); tile0 = s; side = 1;
                }
                const unsigned long long mask = __ballot(m);
                const int ord = n_seen + lane_prefix(mask);          // this wave takes every nsub-th item
                if (m && ord % nsub == sub) items[ord / nsub - n_before] = make_int2(s, (tile0 * 2 + side) * side_elems);
                n_seen += __builtin_popcountll(mask);
            }
            n_items = (n_seen + nsub - 1 - sub) / nsub - n_before;
        }
        __builtin_amdgcn_wave_barrier();       // LDS is in-order per wave; this only pins the compiler

        for (int it_pos = 0; it_pos < n_items; it_pos += UR_GROUP) {
            // ---- scan a group of items: which points have a tap row on row r?  Only the (y0 << 16 | x0) word
            // of the tap record is read here; the lower row is y0 + 1 (a clamped lower row has zero weights).
            // An entry only feeds the 16-pixel tile(s) its two taps fall in, so entries are binned per tile
            // (x0 % 16 == 15 goes to two bins) and a group of 4 entries costs 5 MFMAs instead of 5 * MT: the
            // MFMA pipe is what this kernel saturates first (ablation: scan 7.6 us, + loads 12.5 us, + MFMA 25 us
            // before binning).
            int cnt[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) cnt[m] = 0;
            {
                int yx0[UR_GROUP][2];
#pragma unroll
                for (int g = 0; g < UR_GROUP; ++g) {
                    const int2 it = items[min(it_pos + g, n_items - 1)];
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        yx0[g][h] = reinterpret_cast<const int*>(prm.tapyx + (size_t)it.x * TP + h * 64 + lane)[0];
                }
#pragma unroll
                for (int g = 0; g < UR_GROUP; ++g) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int q = h * 64 + lane;
                        const int y0 = yx0[g][h] >> 16, x0 = yx0[g][h] & 0xffff;
                        const bool hit = it_pos + g < n_items && q < P && (y0 == r || y0 + 1 == r);
#pragma unroll
                        for (int m = 0; m < MT; ++m) {
                            const bool mine = hit && ((x0 >> 4) == m || x0 == 16 * m - 1);
                            const unsigned long long mask = __ballot(mine);
                            if (mine) wl[m * UR_CAP + cnt[m] + lane_prefix(mask)] = (it_pos + g) << 7 | q;
                            cnt[m] += __builtin_popcountll(mask);
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            // virtual entry order: bin 0, padded to a multiple of 4, then bin 1, ...  (every group of 4 is one bin)
            int off[MT + 1];
            off[0] = 0;
#pragma unroll
            for (int m = 0; m < MT; ++m) off[m + 1] = off[m] + ((cnt[m] + 3) & ~3);
            int count = off[MT];

            // ---- drain: UR_EG groups of 4 entries per step, entry k = lane / 16 of each group
            if (prm.debug & 128) count = 0;                // ablation: no drain at all
            for (int e0 = 0; e0 < count; e0 += 4 * UR_EG) {
                float bv[UR_EG][UR_NT];
                int meta[UR_EG];
#pragma unroll
                for (int g = 0; g < UR_EG; ++g) {
                    const int eg = min(e0 + 4 * g, count - 4);               // group start (uniform), clamped
                    int m = 0;
#pragma unroll
                    for (int mm = 1; mm < MT; ++mm) m += eg >= off[mm] ? 1 : 0;
                    int base = 0, n = cnt[0], o = 0;
#pragma unroll
                    for (int mm = 1; mm < MT; ++mm) if (m == mm) { base = mm * UR_CAP; n = cnt[mm]; o = off[mm]; }
                    const int pk = wl[base + min(eg - o + k4, n - 1)];
                    const int2 it = items[pk >> 7];
                    const int q = pk & (TP - 1);
                    const unsigned rowoff = (unsigned)(it.y + q * ldk);
#pragma unroll
                    for (int nt = 0; nt < UR_NT; ++nt) bv[g][nt] = prm.dt[rowoff + (unsigned)cch[nt]];
                    meta[g] = tap_words[(unsigned)(it.x * TP + q) * 4u + meta_off];
                }
                __builtin_amdgcn_sched_barrier(0);          // every load is in flight before the first use waits
                if (prm.debug & 64) {                        // ablation: loads only
                    float sum = 0.f;
#pragma unroll
                    for (int g = 0; g < UR_EG; ++g) {
                        sum += __builtin_bit_cast(float, meta[g]);
#pragma unroll
                        for (int nt = 0; nt < UR_NT; ++nt) sum += bv[g][nt];
                    }
                    acc[0][0][0] += sum;
                    continue;
                }
#pragma unroll
                for (int g = 0; g < UR_EG; ++g) {
                    const int eg = e0 + 4 * g;
                    if (eg < count) {                                          // uniform: skip the tail groups
                        int m = 0;
#pragma unroll
                        for (int mm = 1; mm < MT; ++mm) m += eg >= off[mm] ? 1 : 0;
                        int n = cnt[0], o = 0;
#pragma unroll
                        for (int mm = 1; mm < MT; ++mm) if (m == mm) { n = cnt[mm]; o = off[mm]; }
                        const int yx = __shfl(meta[g], grp_lane0, 64);
                        const int x0 = yx & 0xffff;
                        const int low = (yx >> 16) != r ? 2 : 0;       // my row is the lower tap row of this point
                        float wa = __builtin_bit_cast(float, __shfl(meta[g], grp_lane0 + 4 + low, 64));
                        float wb = __builtin_bit_cast(float, __shfl(meta[g], grp_lane0 + 5 + low, 64));
                        if (eg - o + k4 >= n) { wa = 0.f; wb = 0.f; }
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            if (m == mt) {                                     // uniform
                                const int dx = 16 * mt + l16 - x0;
                                const float a = dx == 0 ? wa : (dx == 1 ? wb : 0.f);
#pragma unroll
                                for (int nt = 0; nt < UR_NT; ++nt)
                                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[g][nt], acc[mt][nt], 0, 0, 0);
                            }
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }

    // ---- the row, once, from the accumulators: acc[mt][nt][reg] is pixel 16 mt + 4 (lane / 16) + reg,
    //      channel 16 nt + lane % 16
    float* out = (dest == 0 ? prm.d_code : prm.d_code_pos) + ((size_t)j * H + r) * W * K;
    if (!heavy) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int x = 16 * mt + 4 * k4 + reg;
#pragma unroll
                for (int nt = 0; nt < UR_NT; ++nt) {
                    const int ch = 16 * nt + l16;
                    if (x < W && ch < K) __builtin_nontemporal_store(acc[mt][nt][reg], out + (size_t)x * K + ch);
                }
            }
        return;
    }
    // heavy row: reduce-scatter the partial rows, one pixel tile (5 accumulator tiles) per round; wave w sums
    // and stores tiles w, w + 3.  The buffer aliases the (now idle) worklists.
    f32x4* red = reinterpret_cast<f32x4*>(wl_raw);             // [wave][5 tiles][64 lanes]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < UR_NT; ++nt) red[(wave * UR_NT + nt) * 64 + lane] = acc[mt][nt];
        __syncthreads();
#pragma unroll
        for (int t = 0; t < (UR_NT + UR_WAVES - 1) / UR_WAVES; ++t) {
            const int nt = wave + UR_WAVES * t;                // wave-uniform
            if (nt < UR_NT) {
                f32x4 v = red[nt * 64 + lane];
#pragma unroll
                for (int w2 = 1; w2 < UR_WAVES; ++w2) v += red[(w2 * UR_NT + nt) * 64 + lane];
                const int ch = 16 * nt + l16;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int x = 16 * mt + 4 * k4 + reg;
                    if (x < W && ch < K) __builtin_nontemporal_store(v[reg], out + (size_t)x * K + ch);
                }
            }
        }
    }
    if (stamp_on) ts[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
}

// =========================================================================================== lists first: the training backward
// corr_bwd_tile_build_kernel + corr_unsample_list_kernel (scalar upstreams, split-fp16 GEMMs, maps up to 64 x 64).  The row kernel
// above spends three dependent load rounds per unit (perms -> items, tap words -> worklist, DT rows) of which only the last needs the
// tile kernel's output.  Here the first two run BESIDE the tiles, in the same launch: with B = 32 the 224 tile workgroups leave 32
// CUs free, and B "builder" workgroups at the front of the grid turn perms + the forward's tap tables into the entry list of every
// (destination, image, pixel row, 16-pixel bin) unit - which DT rows land there, with which weights.  The second launch is then ONE
// wave per unit pair: its lists arrive in one load (fixed slots), every DT row of the unit is in flight at once (16-byte loads), the
// rows are accumulated as one-hot GEMMs on the f32 MFMA in registers, and the pixels leave as 16-byte stores.  One wave per unit, the
// list order fixed by a stable counting sort: no atomics on data, no reduction between waves, bitwise repeatable.
// (A version with the unsample INSIDE the same launch, handing the DT rows over through device counters, was measured slower than the
// kernel boundary it removed: profiles/r04b_bwd_one_launch_attempt.txt.)
constexpr int BF_ITEM_CAP = 1032;                 // items of one destination image: n_sets + n_neg B <= 1026 (fill_bwd_ctx)
constexpr int BF_MAXH = 64, BF_MAXMT = 4;         // maps up to 64 x 64
constexpr int BF_NKEY = 2 * BF_MAXH * BF_MAXMT;   // list keys of one image: (destination, pixel row, 16-pixel bin)
constexpr int UL_CAP0 = 63, UL_CAP1 = 15;         // entries in the slot of a dest-0 / dest-1 unit, behind a 16-byte header {count, overflow start}
constexpr int UL_SLOT0 = 16 * (UL_CAP0 + 1), UL_SLOT1 = 16 * (UL_CAP1 + 1);
constexpr int BFS_ITEMS = 0;                                          // int2[BF_ITEM_CAP]  {sample set, DT base (floats)}
constexpr int BFS_HIST = BFS_ITEMS + BF_ITEM_CAP * 8;                 // unsigned[HW_WAVES][BF_NKEY] counts, then running positions
constexpr int BFS_OVF = BFS_HIST + HW_WAVES * BF_NKEY * 4;            // unsigned[BF_NKEY] overflow start of every list (pool entries)
constexpr int BFS_SCAN = BFS_OVF + BF_NKEY * 4;                       // unsigned[HW_WAVES + 1] partial sums of the block scan
constexpr int BFS_MISC = BFS_SCAN + 64;                               // int[4]: items of dest 0, overflow slab of dest 0 / dest 1
constexpr int BF_LDS_BYTES = BFS_MISC + 64;
static_assert(BF_NKEY % HW_THREADS == 0 && BF_NKEY % NTHREADS == 0, "whole keys per thread in the prefix");

typedef unsigned int bu32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t bf_rsrc(const void* p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// ---- builder role: the entry lists of destination images bi, bi + n_build, ... (dest 0 = orig_code with its anchor and negative
// items, dest 1 = orig_code_pos with its one item, handled together).  A stable counting sort of the (item, point, tap row) sequence
// by list key, in LDS: wave w owns the item halves w, w + 8, ...; counts per (wave, key), a block-wide prefix, then every element
// takes its slot with a returning LDS add on its wave's OWN counter - lanes of one instruction that hit the same counter are
// served in an order the hardware fixes (no other wave touches it), so the lists come out the same launch after launch.
template <int NW>        // waves of the workgroup: 8 beside the split-fp16 tiles, 4 beside the fp32 ones
__device__ __forceinline__ void bwd_builder_role(const BwdParams& prm, const int bi, unsigned char* smem)
{
    int2* items = reinterpret_cast<int2*>(smem + BFS_ITEMS);
    unsigned* hist = reinterpret_cast<unsigned*>(smem + BFS_HIST);
    unsigned* ovf = reinterpret_cast<unsigned*>(smem + BFS_OVF);
    unsigned* scan = reinterpret_cast<unsigned*>(smem + BFS_SCAN);
    int* misc = reinterpret_cast<int*>(smem + BFS_MISC);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int B = prm.B, P = prm.P, H = prm.H, W = prm.W, ldk = prm.LDK;
    const int MT = (W + 15) >> 4;
    const int side_elems = TP * ldk;
    const unsigned EPI = 4u * (unsigned)P;                   // entry slots per item: two rows, at most two bins each
    const unsigned light0 = (unsigned)(B * H * MT) * UL_SLOT0;       // byte offset of the dest-1 slots
    const __amdgpu_buffer_rsrc_t pool = bf_rsrc(prm.upool, prm.upool_bytes);
    const __amdgpu_buffer_rsrc_t slots = bf_rsrc(prm.uslots, prm.uslots_bytes);
    unsigned long long* ts = reinterpret_cast<unsigned long long*>(prm.dt + (size_t)prm.n_sets * B * 2 * side_elems) + 2048 + 2 * bi;
    const bool stamp_on = (prm.debug & 8) && tid == 0 && bi < 512;
    if (stamp_on) ts[0] = __builtin_amdgcn_s_memrealtime();
    unsigned* myhist = hist + wave * BF_NKEY;

    for (int j = bi; j < B; j += prm.n_build) {
        __syncthreads();                                     // the previous image's LDS tables are dead
        // ---- items of dest 0 (the anchor's own tiles, then the negatives whose perm picked image j: modules.py:385), then dest 1's one item
        if (wave == 0) {
            const int n_own = prm.n_sets;
            for (int u = lane; u < n_own; u += 64) items[u] = make_int2(j, ((u * B + j) * 2 + 0) * side_elems);
            int n_seen = 0, below = 0;
            const int n_negc = prm.n_neg * B;
            for (int t0 = 0; t0 < n_negc; t0 += 64) {
                const int t = t0 + lane;
                const long long pv = t < n_negc ? prm.perms[t] : -1;
                const bool m = (int)pv == j && t < n_negc;
                const unsigned long long mask = __ballot(m);
                const int s = 2 * B + t;
                if (m) items[n_own + n_seen + lane_prefix(mask)] = make_int2(s, (s * 2 + 1) * side_elems);
                n_seen += __builtin_popcountll(mask);
                below += __builtin_popcountll(__ballot(t < n_negc && (int)pv < j));
            }
            const int n0 = n_own + n_seen;
            if (lane == 0) {
                items[n0] = make_int2(B + j, ((B + j) * 2 + 1) * side_elems);
                misc[0] = n0;
                misc[1] = (int)(EPI * (unsigned)(prm.n_sets * j + below));          // overflow slabs: room for every entry of the image
                misc[2] = (int)(EPI * (unsigned)(prm.n_sets * B + n_negc + j));
            }
        }
        for (int i = tid; i < NW * BF_NKEY; i += (64 * NW)) hist[i] = 0u;
        __syncthreads();
        const int n0 = misc[0], n_ih = 2 * (n0 + 1);         // item halves: 64 points each
        const unsigned slab0 = (unsigned)misc[1], slab1 = (unsigned)misc[2];
        // the (<= 4) list keys of a point: tap rows y0 and y0 + 1 (a clamped lower row carries zero weights: no entry), in each
        // the bin of x0 and - when x0 is the last pixel of its bin - the bin of x0 + 1
        auto point_keys = [&](int dest, int q, int wd, int (&key)[4]) {
            const int y0 = wd >> 16, x0 = wd & 0xffff;
            const int b0 = x0 >> 4, b1 = (x0 & 15) == 15 && b0 + 1 < MT ? b0 + 1 : -1;
            const bool v = q < P;
            const int r0 = (dest * H + y0) * BF_MAXMT, r1 = r0 + BF_MAXMT;
            key[0] = v ? r0 + b0 : -1;
            key[1] = v && b1 >= 0 ? r0 + b1 : -1;
            key[2] = v && y0 + 1 < H ? r1 + b0 : -1;
            key[3] = v && y0 + 1 < H && b1 >= 0 ? r1 + b1 : -1;
        };
        // ---- pass 0: counts per (wave, key)
        for (int ih0 = wave; ih0 < n_ih; ih0 += 4 * NW) {
            int wd[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ih = min(ih0 + u * NW, n_ih - 1), it = ih >> 1, q = (ih & 1) * 64 + lane;
                wd[u] = reinterpret_cast<const int*>(prm.tapyx + (size_t)items[it].x * TP + q)[0];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ih = ih0 + u * NW;
                if (ih < n_ih) {                             // uniform
                    int key[4];
                    point_keys((ih >> 1) >= n0, (ih & 1) * 64 + lane, wd[u], key);
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (key[c] >= 0) __hip_atomic_fetch_add(myhist + key[c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        __syncthreads();
        // ---- prefix: thread t owns the keys t KPT .. t KPT + KPT - 1.  A list's first UL_CAP entries live in the unit's slot, the rest in
        // the image's overflow slab (lists in key order per destination); inside a list the waves come in order
        {
            constexpr int KPT = BF_NKEY / (64 * NW);
            const int kd1 = H * BF_MAXMT;                    // first key of dest 1
            unsigned c[KPT][NW], tot[KPT], over[KPT], osum = 0u;
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const int key = tid * KPT + i;
                tot[i] = 0u;
#pragma unroll
                for (int w = 0; w < NW; ++w) { c[i][w] = hist[w * BF_NKEY + key]; tot[i] += c[i][w]; }
                const unsigned cap = key >= kd1 ? UL_CAP1 : UL_CAP0;
                over[i] = tot[i] > cap ? tot[i] - cap : 0u;
                osum += over[i];
            }
            unsigned inc = osum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const unsigned o = (unsigned)__shfl_up((int)inc, d, 64);
                if (lane >= d) inc += o;
            }
            if (lane == 63) scan[wave] = inc;
            __syncthreads();
            unsigned base = inc - osum;
#pragma unroll
            for (int w = 0; w < NW; ++w) base += w < wave ? scan[w] : 0u;
            unsigned kb[KPT];                                // overflow entries in front of each of my keys
#pragma unroll
            for (int i = 0; i < KPT; ++i) { kb[i] = base; base += over[i]; }
#pragma unroll
            for (int i = 0; i < KPT; ++i)
                if (tid * KPT + i == kd1) scan[NW] = kb[i];  // dest 1's overflow restarts in its own slab
            __syncthreads();
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const int key = tid * KPT + i;
                const bool d1 = key >= kd1;
                const unsigned ostart = d1 ? slab1 + (kb[i] - scan[NW]) : slab0 + kb[i];
                ovf[key] = ostart;
                const int rm = key - (d1 ? kd1 : 0), m = rm & (BF_MAXMT - 1);     // rm = row * 4 + bin
                if (key < 2 * kd1 && m < MT) {
                    const unsigned ui = (unsigned)((j * H + (rm >> 2)) * MT + m);
                    __builtin_amdgcn_raw_buffer_store_b128(bu32x4{tot[i], ostart, 0u, 0u}, slots, d1 ? light0 + ui * UL_SLOT1 : ui * UL_SLOT0, 0, 0);
                }
                unsigned run = 0u;                           // list-relative running positions per (wave, key)
#pragma unroll
                for (int w = 0; w < NW; ++w) { hist[w * BF_NKEY + key] = run; run += c[i][w]; }
            }
        }
        __syncthreads();
        // ---- pass 1: every element takes its position and leaves its entry {DT row, x0, weight left, weight right}
        const int kd1 = H * BF_MAXMT;
        for (int ih0 = wave; ih0 < n_ih; ih0 += 4 * NW) {
            int wd[4];
            float4 w4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ih = min(ih0 + u * NW, n_ih - 1), it = ih >> 1, q = (ih & 1) * 64 + lane;
                const size_t e = (size_t)items[it].x * TP + q;
                wd[u] = reinterpret_cast<const int*>(prm.tapyx + e)[0];
                w4[u] = prm.tapw[e];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ih = ih0 + u * NW;
                if (ih < n_ih) {                             // uniform
                    const int it = ih >> 1, q = (ih & 1) * 64 + lane;
                    const bool d1 = it >= n0;
                    int key[4];
                    point_keys(d1, q, wd[u], key);
                    bu32x4 ent;
                    ent[0] = (unsigned)(items[it].y + q * ldk);
                    ent[1] = (unsigned)(wd[u] & 0xffff);
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (key[c] >= 0) {
                            const unsigned pos = __hip_atomic_fetch_add(myhist + key[c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            ent[2] = __builtin_bit_cast(unsigned, c >= 2 ? w4[u].z : w4[u].x);
                            ent[3] = __builtin_bit_cast(unsigned, c >= 2 ? w4[u].w : w4[u].y);
                            const int rm = key[c] - (d1 ? kd1 : 0);
                            const unsigned ui = (unsigned)((j * H + (rm >> 2)) * MT + (rm & (BF_MAXMT - 1)));
                            const unsigned cap = d1 ? UL_CAP1 : UL_CAP0;
                            if (pos < cap) __builtin_amdgcn_raw_buffer_store_b128(ent, slots, (d1 ? light0 + ui * UL_SLOT1 : ui * UL_SLOT0) + 16u * (1u + pos), 0, 0);
                            else __builtin_amdgcn_raw_buffer_store_b128(ent, pool, (ovf[key[c]] + (pos - cap)) * 16u, 0, 0);
                        }
                }
            }
        }
    }
    if (stamp_on) ts[1] = __builtin_amdgcn_s_memrealtime();
}

template <int NT>
__global__ void __launch_bounds__(HW_THREADS) corr_bwd_tile_build_kernel(const BwdParams prm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int w = blockIdx.x;
    if (w < prm.n_build) bwd_builder_role<HW_WAVES>(prm, w, smem);
    else bwd_tile_h_body<NT, false>(prm, w - prm.n_build, smem);
}

// the same beside the exact-fp32 tiles (F32 mode): 256-thread workgroups, four builder waves
template <int NT>
__global__ void __launch_bounds__(NTHREADS) corr_bwd_tile32_build_kernel(const BwdParams prm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int w = blockIdx.x;
    if (w < prm.n_build) bwd_builder_role<NTHREADS / 64>(prm, w, smem);
    else bwd_tile_body<NT>(prm, w - prm.n_build, smem);
}

// ---- second launch: wave k takes the pair of units {dest 0, dest 1} x (image, row, bin) k
// Channels: NQ x 64 as 16-byte loads (lane l16 holds channels 64 qd + 4 l16 .. + 3: accumulator tile 4 qd + c is channel 64 qd + 4 l16 + c,
// so a pixel leaves as one 16-byte store per lane), + TAIL 16-channel tiles as 4-byte loads (channel 64 NQ + l16).
constexpr int UL_WAVES = 4;
template <int NQ, int TAIL>
__global__ void __launch_bounds__(64 * UL_WAVES) corr_unsample_list_kernel(const BwdParams prm)
{
    constexpr int NA = 4 * NQ + TAIL;                        // accumulator tiles
    constexpr int EG0 = (UL_CAP0 + 1) / 4, EG1 = (UL_CAP1 + 1) / 4;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int B = prm.B, H = prm.H, W = prm.W, K = prm.K, ldk = prm.LDK;
    const int MT = (W + 15) >> 4;
    const int n_pairs = B * H * MT;
    // every unit of image j on XCD j % 8 (block b runs on XCD b % 8: observed, used for speed only): the builder of image j and the tiles
    // of anchor j ran there, and plain stores leave their lines in that L2 across the kernel boundary (measured -1 us per step)
    int k;
    {
        const int upi = H * MT, wpi = (upi + UL_WAVES - 1) / UL_WAVES;
        const int x = blockIdx.x & 7, sl = blockIdx.x >> 3;
        const int j = (sl / wpi) * 8 + x, u = (sl % wpi) * UL_WAVES + wave;
        if (j >= B || u >= upi) return;
        k = j * upi + u;
    }
    if (k >= n_pairs) return;
    const int l16 = lane & 15, k4 = lane >> 4;
    const __amdgpu_buffer_rsrc_t pool = bf_rsrc(prm.upool, prm.upool_bytes);
    const __amdgpu_buffer_rsrc_t slots = bf_rsrc(prm.uslots, prm.uslots_bytes);
    const __amdgpu_buffer_rsrc_t dtr = bf_rsrc(prm.dt, prm.dt_bytes);
    const unsigned light0 = (unsigned)n_pairs * UL_SLOT0;
    unsigned long long* ts = reinterpret_cast<unsigned long long*>(prm.dt + (size_t)prm.n_sets * B * 2 * TP * ldk) + 2 * blockIdx.x;
    const bool stamp_on = (prm.debug & 8) && threadIdx.x == 0 && blockIdx.x < 1024;
    if (stamp_on) ts[0] = __builtin_amdgcn_s_memrealtime();
    const int m = k % MT, jr = k / MT;                       // jr = image * H + row

    // both slots in one round: lane 0 the header, lane i entry i - 1
    bu32x4 s0 = __builtin_amdgcn_raw_buffer_load_b128(slots, (unsigned)k * UL_SLOT0 + 16u * lane, 0, 0);
    bu32x4 s1 = __builtin_amdgcn_raw_buffer_load_b128(slots, light0 + (unsigned)k * UL_SLOT1 + 16u * (lane & 15), 0, 0);
    const unsigned n0 = __builtin_amdgcn_readfirstlane(s0[0]), o0 = __builtin_amdgcn_readfirstlane(s0[1]);
    const unsigned n1 = __builtin_amdgcn_readfirstlane(s1[0]), o1 = __builtin_amdgcn_readfirstlane(s1[1]);

    f32x4 acc[2][NA];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int a = 0; a < NA; ++a) acc[u][a] = f32x4{0.f, 0.f, 0.f, 0.f};

    // the DT rows of NG groups of 4 entries held by the lanes first .. first + n - 1 of `ent` (group g, lane group k4: entry 4 g + k4)
    struct Rows { f32x4 q[NQ]; float t[TAIL > 0 ? TAIL : 1]; };
    auto issue = [&](const bu32x4& ent, int first, int n, auto& rows, auto ng_tag) {
        constexpr int NG = decltype(ng_tag)::value;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (4 * g < n) {                                  // uniform: groups beyond the list issue nothing
                const int e = min(4 * g + k4, n - 1);
                const unsigned row = (unsigned)__shfl((int)ent[0], first + e, 64);
#pragma unroll
                for (int qd = 0; qd < NQ; ++qd)
                    rows[g].q[qd] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dtr, (row + 64u * qd + 4u * l16) * 4u, 0, 0));
#pragma unroll
                for (int t = 0; t < TAIL; ++t)
                    rows[g].t[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dtr, (row + (unsigned)min(64 * NQ + 16 * t + l16, ldk - 1)) * 4u, 0, 0));
            }
        }
    };
    auto consume = [&](const bu32x4& ent, int first, int n, auto& rows, f32x4 (&ac)[NA], auto ng_tag) {
        constexpr int NG = decltype(ng_tag)::value;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (4 * g < n) {                                  // uniform
                const int e = min(4 * g + k4, n - 1);
                const int x0 = __shfl((int)ent[1], first + e, 64);
                float wa = __builtin_bit_cast(float, __shfl((int)ent[2], first + e, 64));
                float wb = __builtin_bit_cast(float, __shfl((int)ent[3], first + e, 64));
                if (4 * g + k4 >= n) { wa = 0.f; wb = 0.f; }
                const int dx = 16 * m + l16 - x0;
                const float a = dx == 0 ? wa : (dx == 1 ? wb : 0.f);
#pragma unroll
                for (int qd = 0; qd < NQ; ++qd)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        ac[4 * qd + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, rows[g].q[qd][c], ac[4 * qd + c], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < TAIL; ++t)
                    ac[4 * NQ + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, rows[g].t[t], ac[4 * NQ + t], 0, 0, 0);
            }
        }
    };
    {
        Rows r0[EG0], r1[EG1];
        const int c0 = (int)min(n0, (unsigned)UL_CAP0), c1 = (int)min(n1, (unsigned)UL_CAP1);
        issue(s0, 1, c0, r0, std::integral_constant<int, EG0>());
        issue(s1, 1, c1, r1, std::integral_constant<int, EG1>());
        __builtin_amdgcn_sched_barrier(0);                   // every DT row is in flight before the first use waits
        consume(s0, 1, c0, r0, acc[0], std::integral_constant<int, EG0>());
        consume(s1, 1, c1, r1, acc[1], std::integral_constant<int, EG1>());
    }
    // lists longer than their slot (a few percent of the units at the training shape): 64 entries per round from the overflow slab
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int rest = u ? (int)n1 - UL_CAP1 : (int)n0 - UL_CAP0;
        const unsigned ob = u ? o1 : o0;
        for (int e0 = 0; e0 < rest; e0 += 64) {
            const int n = min(64, rest - e0);
            const bu32x4 ent = __builtin_amdgcn_raw_buffer_load_b128(pool, (ob + (unsigned)(e0 + min(lane, n - 1))) * 16u, 0, 0);
            Rows r[16];
            issue(ent, 0, n, r, std::integral_constant<int, 16>());
            __builtin_amdgcn_sched_barrier(0);
            consume(ent, 0, n, r, acc[u], std::integral_constant<int, 16>());
        }
    }
    // acc[u][a][reg] is pixel 16 m + 4 k4 + reg of row jr of destination u; channel 64 qd + 4 l16 + c (a = 4 qd + c), 64 NQ + 16 t + l16 (tail)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        float* out = (u == 0 ? prm.d_code : prm.d_code_pos) + (size_t)jr * W * K;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int x = 16 * m + 4 * k4 + reg;
            if (x < W) {
                float* px = out + (size_t)x * K;
#pragma unroll
                for (int qd = 0; qd < NQ; ++qd) {
                    const int ch = 64 * qd + 4 * l16;
                    const f32x4 v = f32x4{acc[u][4 * qd][reg], acc[u][4 * qd + 1][reg], acc[u][4 * qd + 2][reg], acc[u][4 * qd + 3][reg]};
                    if (ch + 3 < K) {
                        typedef f32x4 f32x4_a4 __attribute__((aligned(4)));
                        __builtin_nontemporal_store(v, reinterpret_cast<f32x4_a4*>(px + ch));
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (ch + c < K) __builtin_nontemporal_store(v[c], px + ch + c);
                    }
                }
#pragma unroll
                for (int t = 0; t < TAIL; ++t) {
                    const int ch = 64 * NQ + 16 * t + l16;
                    if (ch < K) __builtin_nontemporal_store(acc[u][4 * NQ + t][reg], px + ch);
                }
            }
        }
    }
    if (stamp_on) ts[1] = __builtin_amdgcn_s_memrealtime();
}

// the lists-first backward covers: forward() semantics, scalar upstreams, split-fp16 GEMMs, maps up to 64 x 64
bool bwd_lists_supported(const BwdParams& prm)
{
    const int nt = (prm.KQ + 15) / 16;
    const bool dense = prm.g_intra_cd || prm.g_inter_cd || prm.g_neg_cd || (prm.g_neg_loss && prm.g_neg_loss_stride > 0);
    if ((prm.debug & 512) && nt > 5) return false;       // (the fp32-MFMA tile kernel ends at K = 80)
    return prm.uslots && prm.upool && prm.mode == 0 && !dense && prm.H <= BF_MAXH && prm.W <= 16 * BF_MAXMT &&
           !(prm.debug & (32 | 1024)) && (size_t)prm.n_sets * prm.B * 2 * TP * prm.LDK < ((size_t)1 << 30);
}

hipError_t launch_corr_bwd_lists(const BwdParams& prm_in, hipStream_t stream)
{
    BwdParams prm = prm_in;
    const int nt = (prm.KQ + 15) / 16;
    const int ntg = nt <= 5 ? nt : (nt + 1) / 2;
    // split-fp16 GEMMs in F16X3 mode and - whatever the mode - for code dimensions above 80 (launch_corr_bwd has the same rule)
    const bool split = !(prm.debug & 512) && (prm.precision == PREC_F16X3 || nt > 5);
    int lds = split ? SMH_CT + (4 * 16 * ntg + 2 * TP) * HB_LDR * 2 : SMB_AN + 2 * TP * prm.LDK * 4 + TP * LDG * 4;
    if (lds < BF_LDS_BYTES) lds = BF_LDS_BYTES;
    const int n_tiles = prm.n_sets * prm.B;
    prm.n_build = prm.B < 32 ? prm.B : 32;
    {
        const dim3 grid(prm.n_build + n_tiles);
#define STEGO_BWDL_CASE(N)                                                                                        \
    case N: {                                                                                                     \
        hipError_t ea = ensure_dynamic_lds(reinterpret_cast<const void*>(&corr_bwd_tile_build_kernel<N>), lds);   \
        if (ea != hipSuccess) return ea;                                                                          \
        hipLaunchKernelGGL((corr_bwd_tile_build_kernel<N>), grid, dim3(HW_THREADS), lds, stream, prm);            \
        break;                                                                                                    \
    }
#define STEGO_BWDL32_CASE(N)                                                                                      \
    case N: {                                                                                                     \
        hipError_t ea = ensure_dynamic_lds(reinterpret_cast<const void*>(&corr_bwd_tile32_build_kernel<N>), lds); \
        if (ea != hipSuccess) return ea;                                                                          \
        hipLaunchKernelGGL((corr_bwd_tile32_build_kernel<N>), grid, dim3(NTHREADS), lds, stream, prm);            \
        break;                                                                                                    \
    }
        if (split) {
            switch (nt) {
                STEGO_BWDL_CASE(1)
                STEGO_BWDL_CASE(2)
                STEGO_BWDL_CASE(3)
                STEGO_BWDL_CASE(4)
                STEGO_BWDL_CASE(5)
                STEGO_BWDL_CASE(6)
                STEGO_BWDL_CASE(7)
                default:
                STEGO_BWDL_CASE(8)
            }
        } else {
            switch (nt) {
                STEGO_BWDL32_CASE(1)
                STEGO_BWDL32_CASE(2)
                STEGO_BWDL32_CASE(3)
                STEGO_BWDL32_CASE(4)
                default:
                STEGO_BWDL32_CASE(5)
            }
        }
#undef STEGO_BWDL_CASE
#undef STEGO_BWDL32_CASE
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    const int upi = prm.H * ((prm.W + 15) >> 4), wpi = (upi + UL_WAVES - 1) / UL_WAVES;
    const dim3 grid(((prm.B + 7) / 8) * 8 * wpi), block(64 * UL_WAVES);
    if (prm.K <= 64) hipLaunchKernelGGL((corr_unsample_list_kernel<1, 0>), grid, block, 0, stream, prm);
    else if (prm.K <= 80) hipLaunchKernelGGL((corr_unsample_list_kernel<1, 1>), grid, block, 0, stream, prm);
    else hipLaunchKernelGGL((corr_unsample_list_kernel<2, 0>), grid, block, 0, stream, prm);
    return hipGetLastError();
}

hipError_t launch_corr_bwd(const BwdParams& prm, hipStream_t stream)
{
    if (bwd_lists_supported(prm)) return launch_corr_bwd_lists(prm, stream);
    // ---- tile kernel
    {
        const int cside = TP * prm.LDK * 4;
        const int nt = (prm.KQ + 15) / 16;
        // forward() semantics: split-fp16 GEMMs in F16X3 mode, and - whatever the mode - for code dimensions above 80, whose
        // fp32 operand images no longer fit LDS (the split kernel walks the channel tiles in two groups)
        const bool split = prm.mode == 0 && !(prm.debug & 512) && (prm.precision == PREC_F16X3 || nt > 5);  // (debug 512: fp32 MFMA kernel)
        const int ntg = nt <= 5 ? nt : (nt + 1) / 2;
        const bool dense = prm.g_intra_cd || prm.g_inter_cd || prm.g_neg_cd || (prm.g_neg_loss && prm.g_neg_loss_stride > 0);
        const int lds = split ? SMH_CT + (4 * 16 * ntg + 2 * TP) * HB_LDR * 2 : SMB_AN + 2 * cside + TP * LDG * 4;
        const dim3 grid(prm.n_sets * prm.B), block(NTHREADS);
#define STEGO_BWD_CASE(N)                                                                                         \
    case N: {                                                                                                     \
        if (split && dense) {                                                                                     \
            hipError_t ea = ensure_dynamic_lds(reinterpret_cast<const void*>(&corr_bwd_tile_h_kernel<N, true>), lds);   \
            if (ea != hipSuccess) return ea;                                                                      \
            hipLaunchKernelGGL((corr_bwd_tile_h_kernel<N, true>), grid, dim3(HW_THREADS), lds, stream, prm);      \
        } else if (split) {                                                                                       \
            hipError_t ea = ensure_dynamic_lds(reinterpret_cast<const void*>(&corr_bwd_tile_h_kernel<N, false>), lds);  \
            if (ea != hipSuccess) return ea;                                                                      \
            hipLaunchKernelGGL((corr_bwd_tile_h_kernel<N, false>), grid, dim3(HW_THREADS), lds, stream, prm);     \
        } else {                                                                                                  \
            hipError_t ea = ensure_dynamic_lds(reinterpret_cast<const void*>(&corr_bwd_tile_kernel<N>), lds);    \
            if (ea != hipSuccess) return ea;                                                                      \
            hipLaunchKernelGGL((corr_bwd_tile_kernel<N>), grid, block, lds, stream, prm);                         \
        }                                                                                                         \
        break;                                                                                                    \
    }
        if (nt > 5 && !split) return hipErrorInvalidValue;       // (helper() is limited to K <= 72 by check_desc)
        if (nt > 5) {
#define STEGO_BWD_WIDE(N)                                                                                         \
    case N: {                                                                                                     \
        if (dense) {                                                                                              \
            hipError_t ea = ensure_dynamic_lds(reinterpret_cast<const void*>(&corr_bwd_tile_h_kernel<N, true>), lds);   \
            if (ea != hipSuccess) return ea;                                                                      \
            hipLaunchKernelGGL((corr_bwd_tile_h_kernel<N, true>), grid, dim3(HW_THREADS), lds, stream, prm);      \
        } else {                                                                                                  \
            hipError_t ea = ensure_dynamic_lds(reinterpret_cast<const void*>(&corr_bwd_tile_h_kernel<N, false>), lds);  \
            if (ea != hipSuccess) return ea;                                                                      \
            hipLaunchKernelGGL((corr_bwd_tile_h_kernel<N, false>), grid, dim3(HW_THREADS), lds, stream, prm);     \
        }                                                                                                         \
        break;                                                                                                    \
    }
            switch (nt) {
                STEGO_BWD_WIDE(6)
                STEGO_BWD_WIDE(7)
                default:
                STEGO_BWD_WIDE(8)
            }
#undef STEGO_BWD_WIDE
        } else
        switch (nt) {
            STEGO_BWD_CASE(1)
            STEGO_BWD_CASE(2)
            STEGO_BWD_CASE(3)
            STEGO_BWD_CASE(4)
            default:
            STEGO_BWD_CASE(5)
        }
#undef STEGO_BWD_CASE
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    // ---- unsample: row-owned register kernel for W <= 64, LDS band kernel beyond
    if (prm.W <= 64 && (size_t)prm.n_sets * prm.B * 2 * TP * prm.LDK < ((size_t)1 << 31) && !(prm.debug & 32)) {
        const int n_heavy = prm.mode == 1 ? 0 : prm.B * prm.H;
        const dim3 grid(n_heavy + (2 * prm.B * prm.H - n_heavy + UR_WAVES - 1) / UR_WAVES), block(UR_WAVES * 64);
        const bool wide_k = prm.KQ > 80;
        if (prm.W <= 32) {
            if (wide_k) hipLaunchKernelGGL((corr_unsample_row_kernel<2, 8>), grid, block, 0, stream, prm);
            else hipLaunchKernelGGL((corr_unsample_row_kernel<2, 5>), grid, block, 0, stream, prm);
        } else {
            if (wide_k) hipLaunchKernelGGL((corr_unsample_row_kernel<4, 8>), grid, block, 0, stream, prm);
            else hipLaunchKernelGGL((corr_unsample_row_kernel<4, 5>), grid, block, 0, stream, prm);
        }
        return hipGetLastError();
    }
    // ---- unsample kernel: bands of RT <= 8 rows (one per wave) with RT*W*K floats <= ~96 KB of LDS
    {
        const int row_bytes = prm.W * prm.K * 4;
        int RT = (96 * 1024) / row_bytes;
        if (RT > UNS_MAX_RT) RT = UNS_MAX_RT;
        if (RT < 1) RT = 1;
        if (RT > prm.H) RT = prm.H;
        const int n_bands = (prm.H + RT - 1) / RT;
        RT = (prm.H + n_bands - 1) / n_bands;                 // even bands
        const int acc_bytes = ((RT * row_bytes + 15) & ~15) + 4 * 64 * 4 + 256;     // band + per-lane dummy cells
        const int lds = acc_bytes + UNS_MAX_CONTRIB * 4 + UNS_MAX_RT * UNS_ROW_CAP * (int)sizeof(UnsRowEntry) + 64;
        hipError_t ea = ensure_dynamic_lds(reinterpret_cast<const void*>(&corr_unsample_kernel), lds);
        if (ea != hipSuccess) return ea;
        const dim3 grid(2 * prm.B * n_bands), block(UNS_THREADS);
        hipLaunchKernelGGL(corr_unsample_kernel, grid, block, lds, stream, prm, RT, n_bands, acc_bytes);
    }
    return hipGetLastError();
}

}  // namespace stego
