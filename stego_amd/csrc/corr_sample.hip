// Stage 1 of the forward: bilinear sampling + L2 normalisation of every (role, image) "set"
// exactly once, written in the LDS-image layout the tile kernel copies with global_load_lds.
//
//   reference: sample() modules.py:287-288 (grid_sample bilinear/border/align_corners) and
//              norm() :275-276, applied at :369-373 (anchor @coords1, positive @coords2) and
//              :384-385 (negatives: orig_feats[perm] / orig_code[perm] @coords2).
//
// Sets: s = role*B + b.  role 0 = anchor (feats/code[b] @ coords1[b]); role 1 = positive
// (feats_pos/code_pos[b] @ coords2[b]); role 2+i = negative i (feats/code[perm_i[b]] @ coords2[b]).
// helper mode: role 0 = (f1,c1), role 1 = (f2,c2), both taken pixel-for-pixel (already sampled).
//
// Why a separate pass: in the fused-gather kernel every negative tile pulled its source image into
// a different XCD's L2 (166 MB of L2 misses for 91 MB of input).  Here the work is placed so that all
// sets whose SOURCE image is j run on XCD j%8 (blockIdx%8 -> XCD is the observed dispatch order; it
// only affects speed): each image is fetched from HBM once and re-read from that XCD's L2.
//
// One half-wave (32 lanes) owns one sample point: lane hl holds channels 128*j + 4*hl .. +3, so a
// wave instruction is two coalesced 512 B runs; the row norm is a 5-step shuffle reduction.
#include "corr_common.h"

namespace stego {

struct PointTaps {
    int4 yx;
    float4 w;
    bool valid;
};

template <int PREC>
__device__ __forceinline__ void store_feat4(const SampleParams& prm, int s, int q, int c, f32x4 v)
{
    const int chunk = c >> 6, col = c & 63;
    if constexpr (PREC == PREC_F32) {
        float* d = static_cast<float*>(prm.fs) + (((size_t)s * prm.NCH + chunk) * TP + q) * LDA + col;
        *reinterpret_cast<f32x4*>(d) = v;
    } else {
        __bf16* dh = static_cast<__bf16*>(prm.fs) + ((((size_t)s * prm.NCH + chunk) * 2) * TP + q) * LDH + col;
        __bf16* dl = dh + TP * LDH;
        unsigned h0, l0, h1, l1;
        split_bf16_pair(v[0], v[1], h0, l0);
        split_bf16_pair(v[2], v[3], h1, l1);
        *reinterpret_cast<u32x2*>(dh) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(dl) = u32x2{l0, l1};
    }
}

template <int PREC>
__device__ __forceinline__ void store_feat1(const SampleParams& prm, int s, int q, int c, float v)
{
    const int chunk = c >> 6, col = c & 63;
    if constexpr (PREC == PREC_F32) {
        static_cast<float*>(prm.fs)[(((size_t)s * prm.NCH + chunk) * TP + q) * LDA + col] = v;
    } else {
        __bf16* dh = static_cast<__bf16*>(prm.fs) + ((((size_t)s * prm.NCH + chunk) * 2) * TP + q) * LDH + col;
        unsigned h0, l0;
        split_bf16_pair(v, 0.f, h0, l0);
        *reinterpret_cast<unsigned short*>(dh) = (unsigned short)(h0 & 0xffffu);
        *reinterpret_cast<unsigned short*>(dh + TP * LDH) = (unsigned short)(l0 & 0xffffu);
    }
}

__device__ __forceinline__ float half_wave_sum(float v)
{
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);   // xor masks < 32 stay inside the half-wave
    return v;
}

// NJ > 0: channels-last, 16-byte aligned, C == 128*NJ: everything stays in registers.
// NJ == 0: any strides / any C: two gather passes (norm, then normalised write).
template <int NJ, int PREC>
__global__ void __launch_bounds__(NTHREADS) sample_norm_kernel(const SampleParams prm)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hl = lane & 31, hw = lane >> 5;
    const int B = prm.B, P = prm.P;
    const int nsets = prm.n_roles * B;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    const bool direct = prm.mode == 1;

    // Work assignment without any division: walk (role, b) with counters; the units of this block's
    // XCD class are numbered in walk order and block `slot` takes numbers slot, slot+nslots, ...
    int ordinal = 0, next_mine = slot;
    (void)nsets;
    for (int role = 0; role < prm.n_roles; ++role) {
        for (int b0 = 0; b0 < B; b0 += 64) {
            bool match = false;
            {
                const int bb = b0 + lane;
                if (bb < B) {
                    const int src = (!direct && role >= 2) ? (int)prm.perms[(size_t)(role - 2) * B + bb] : bb;
                    match = (src & 7) == xcd;
                }
            }
            unsigned long long mask = __ballot(match);
            while (mask) {
                const int b = b0 + __builtin_ctzll(mask);
                mask &= mask - 1;
                if (next_mine >= ordinal + 4) { ordinal += 4; continue; }      // none of this set's 4 units is mine
                for (int qt = 0; qt < 4; ++qt, ++ordinal) {
                    if (ordinal != next_mine) continue;
                    next_mine += nslots;
                    // ---------------- one unit: set s = role*B + b, points [32*qt, 32*qt+32)
                    const int s = role * B + b;
                    const int src = (!direct && role >= 2) ? (int)prm.perms[(size_t)(role - 2) * B + b] : b;
                    const bool pos = role == 1;
                    const MapV mf = pos ? prm.feats_pos : prm.feats;
                    const MapV mc = pos ? prm.code_pos : prm.code;
                    const float* fimg = mf.p + (long long)src * mf.sn;
                    const float* cimg = mc.p + (long long)src * mc.sn;
                    const float* coords = direct ? nullptr : ((role == 0 ? prm.coords1 : prm.coords2) + (size_t)b * P * 2);
                // the 4 points this half-wave owns in the unit: fetch their coords up front
                int4 yxs[4];
                float4 ws[4];
#pragma unroll
                for (int it = 0; it < 4; ++it)
                    tap_for_point(32 * qt + 8 * it + 2 * wave + hw, P, prm.S, prm.H, prm.W, direct, coords, yxs[it], ws[it]);
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int q = 32 * qt + 8 * it + 2 * wave + hw;
                    const int4 yx = yxs[it];
                    const float4 w = ws[it];
                    const bool valid = q < P;
                    if (hl == 0 && prm.tapyx) {
                        prm.tapyx[(size_t)s * TP + q] = yx;
                        prm.tapw[(size_t)s * TP + q] = w;
                    }
                    // ---- features
                    const int4 of = taps_to_offsets(yx, mf.sh, mf.sw);
                    if constexpr (NJ > 0) {
                        f32x4 v[NJ];
                        f32x4 t[NJ][4];
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            const float* base = fimg + 128 * j + 4 * hl;
                            t[j][0] = *reinterpret_cast<const f32x4*>(base + of.x);
                            t[j][1] = *reinterpret_cast<const f32x4*>(base + of.y);
                            t[j][2] = *reinterpret_cast<const f32x4*>(base + of.z);
                            t[j][3] = *reinterpret_cast<const f32x4*>(base + of.w);
                        }
                        float ss = 0.f;
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float r = w.x * t[j][0][e] + w.y * t[j][1][e] + w.z * t[j][2][e] + w.w * t[j][3][e];
                                r = valid ? r : 0.f;
                                v[j][e] = r;
                                ss += r * r;
                            }
                        ss = half_wave_sum(ss);
                        const float inv = 1.f / fmaxf(sqrtf(ss), 1e-10f);       // F.normalize eps (modules.py:276)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) store_feat4<PREC>(prm, s, q, 128 * j + 4 * hl, v[j] * inv);
                    } else {
                        const int cpad = prm.NCH * 64;
                        float ss = 0.f;
                        for (int c = hl; c < prm.C; c += 32) {
                            const float* base = fimg + (long long)c * mf.sc;
                            const float r = w.x * base[of.x] + w.y * base[of.y] + w.z * base[of.z] + w.w * base[of.w];
                            ss += valid ? r * r : 0.f;
                        }
                        ss = half_wave_sum(ss);
                        const float inv = 1.f / fmaxf(sqrtf(ss), 1e-10f);
                        for (int c = hl; c < cpad; c += 32) {
                            float r = 0.f;
                            if (c < prm.C && valid) {
                                const float* base = fimg + (long long)c * mf.sc;
                                r = (w.x * base[of.x] + w.y * base[of.y] + w.z * base[of.z] + w.w * base[of.w]) * inv;
                            }
                            store_feat1<PREC>(prm, s, q, c, r);
                        }
                    }
                    // ---- code: lane hl holds channels hl + 32*m
                    {
                        const int4 oc = taps_to_offsets(yx, mc.sh, mc.sw);
                        constexpr int MK = 5;           // K <= 144 (host-checked)
                        float cv[MK];
                        float ss = 0.f;
                        float ct[MK][4];
#pragma unroll
                        for (int m = 0; m < MK; ++m) {          // branch-free: out-of-range channels re-read channel 0
                            const int c = hl + 32 * m;
                            const float* base = cimg + (long long)(c < prm.K ? c : 0) * mc.sc;
                            if (32 * m < prm.K) {               // (wave-uniform) skip whole groups beyond K
                                ct[m][0] = base[oc.x]; ct[m][1] = base[oc.y]; ct[m][2] = base[oc.z]; ct[m][3] = base[oc.w];
                            } else {
                                ct[m][0] = ct[m][1] = ct[m][2] = ct[m][3] = 0.f;
                            }
                        }
#pragma unroll
                        for (int m = 0; m < MK; ++m) {
                            const int c = hl + 32 * m;
                            float r = w.x * ct[m][0] + w.y * ct[m][1] + w.z * ct[m][2] + w.w * ct[m][3];
                            r = (valid && c < prm.K) ? r : 0.f;
                            cv[m] = r;
                            ss += r * r;
                        }
                        ss = half_wave_sum(ss);
                        const float nr = sqrtf(ss);
                        const float inv = 1.f / fmaxf(nr, 1e-10f);
                        float* crow = prm.cs + ((size_t)s * TP + q) * prm.LDK;
#pragma unroll
                        for (int m = 0; m < MK; ++m) {
                            const int c = hl + 32 * m;
                            if (c < prm.KQ) crow[c] = cv[m] * inv;
                        }
                        if (hl == 0) prm.nrm[(size_t)s * TP + q] = nr;
                    }
                    }
                }
            }
        }
    }
}

hipError_t launch_corr_sample(const SampleParams& prm, int precision, hipStream_t stream)
{
    auto cl4 = [&](const MapV& m) {
        return m.sc == 1 && (m.sn % 4) == 0 && (m.sh % 4) == 0 && (m.sw % 4) == 0 &&
               (reinterpret_cast<uintptr_t>(m.p) % 16) == 0;
    };
    int nj = 0;
    if (prm.C % 128 == 0 && prm.C <= 768 && cl4(prm.feats) && cl4(prm.feats_pos)) nj = prm.C / 128;
    const int units = prm.n_roles * prm.B * 4;
    int nslots = (units + 7) / 8;
    if (nslots > 256) nslots = 256;
    if (nslots < 1) nslots = 1;
    const dim3 grid(8 * nslots), block(NTHREADS);
#define STEGO_SAMPLE_CASE(N)                                                                                       \
    case N:                                                                                                        \
        if (precision == PREC_BF16X3) hipLaunchKernelGGL((sample_norm_kernel<N, PREC_BF16X3>), grid, block, 0, stream, prm); \
        else hipLaunchKernelGGL((sample_norm_kernel<N, PREC_F32>), grid, block, 0, stream, prm);                   \
        break;
    switch (nj) {
        STEGO_SAMPLE_CASE(1)
        STEGO_SAMPLE_CASE(2)
        STEGO_SAMPLE_CASE(3)
        STEGO_SAMPLE_CASE(4)
        STEGO_SAMPLE_CASE(6)
        default:
        STEGO_SAMPLE_CASE(0)
    }
#undef STEGO_SAMPLE_CASE
    return hipGetLastError();
}

}  // namespace stego
