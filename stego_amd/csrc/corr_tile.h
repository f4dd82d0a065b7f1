// Device-side building blocks of the correlation tile kernels (gfx950), shared by the three-launch forward
// (corr_fwd.hip: pre-sampled anchors, generic shapes, helper mode) and the fused forward (corr_fused.hip).
#pragma once
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
// AUX = 16 (sc1): the copy bypasses this CU's L1 - for data another workgroup of the same launch has published.
template <int AUX = 0>
__device__ __forceinline__ void issue_copy(const unsigned char* __restrict__ gsrc, unsigned char* lds_dst, int npieces,
                                           int wave, int lane)
{
    for (int pc = wave; pc < npieces; pc += 4) {
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(gsrc + (size_t)pc * 1024 + lane * 16),
            (__attribute__((address_space(3))) void*)(lds_dst + pc * 1024), 16, 0, AUX);
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
    // Branch-free: elements of the padding rows / columns go to a per-lane dummy word in the unused tail of the tile region
    // (P * P + 6 <= TP * LDT - 72).  With a branch per element the compiler put every ds_write into its own block behind an
    // s_waitcnt lgkmcnt(0): 64 serialised LDS round trips, 2.1 us per tile (measured, tools/stamps_fused.py).
    const int dummy = TP * LDT - 72 + lane;          // (T may be shifted by up to 3 floats)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int col = 64 * wc + 32 * ni + (lane & 31);
        const float sc = colscale ? colscale[col] : 1.f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 64 * wr + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                T[(row < P && col < P) ? row * P + col : dummy] = acc[mi][ni][r] * sc;
            }
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

// bytes of one stage buffer: max(feature chunk pair, code operand pair)
inline int dense_stage_bytes(int precision, int LDK)
{
    const int f = 2 * (precision == PREC_F32 ? FEAT_SIDE_F32 : FEAT_SIDE_F16);
    const int c = 2 * TP * LDK * 4;
    return f > c ? f : c;
}

inline int dense_lds_bytes(int precision, int LDK)
{
    const int st = 2 * dense_stage_bytes(precision, LDK);
    return SD_BIG + (st > SM_TILES_BYTES ? st : SM_TILES_BYTES);
}

}  // namespace stego
