// ROUND 6 ATTEMPT, NOT BUILT INTO THE LIBRARY (kept for the record; the product kernel is csrc/dense_stream.hip).
// One workgroup per 128 x 128 tile, two workgroups per compute unit, BOTH maps converted fp32 -> split fp16 by the multiplying workgroup.
// Measured on MI355X with a stand-alone harness (HIP events, B = 32, C = 384): 28 x 28 maps 154 us (dense_stream: 98), 32 x 32 maps (whole
// tiles only) 167 us; with the conversion compiled out 78 us, with the MFMAs compiled out 98 us, with neither conversion nor loads 70 us
// (= the MFMA time of 2048 tiles at ~22 ns per v_mfma_f32_32x32x16_f16 when every compute unit multiplies): conversion and MFMAs of the two
// co-resident workgroups ADD UP instead of overlapping, and a tile that re-converts its A rows for every column block does seven times
// the conversion work of a kernel that keeps them in registers.  What the attempt taught: (i) VALU work beside MFMAs is not free on this
// part - keep it out of the multiplying kernel or keep it small; (ii) an unrolled chunk loop lets the scheduler hoist fragment reads and
// conversions of several chunks over each other (300-470 spilled registers at a 256-register budget) - roll it, peel first and last.
// Dense feature correspondence in ONE launch (round 6)  out[n,h,w,i,j] = sum_c a[n,c,h,w] * b[n,c,i,j]  (gfx950).
//
//   reference: tensor_correlation() src/modules.py:283-284 = einsum("nchw,ncij->nhwij") on norm()'ed maps (:275-276); the full-resolution
//              callers are plot_dino_correspondence.py:39-58 / plot_pr_curves.py:108-121.  SURVEY.md 8(f) rank 3.
//
// dense_corr.hip needs a prep launch that writes split-fp16 operand images (read 38.5 MB + write 43 MB per map: 28 us at [32,784,784],
// C = 384) and then either keeps a row block's fragments in 192 registers of ONE wave per SIMD (row-block kernel: every LDS round trip,
// conversion and store of that wave is time its matrix core idles - 93 us) or streams prepared tiles (tile kernel).  Here:
//   * one workgroup (4 waves, 64 x 64 each) per 128 x 128 output tile, TWO workgroups per compute unit (<= 256 registers, 74 KB of LDS):
//     while one converts and waits, the other's MFMAs own the matrix cores - the overlap is the hardware's, not a schedule's;
//   * both maps are read as they lie (fp32, channels-last): per 64-channel chunk a lane reads 16 bytes of eight rows of either side
//     (whole 256-byte runs), scales, splits into fp16 hi / lo and stores into the ONE LDS stage; the next chunk's loads fly under the MFMAs;
//   * no pass over the maps in front: a row's staging scale is a power of two from its FIRST chunk (largest magnitude there in
//     [1/16, 1/8): later chunks may be 2^19 times larger before an fp16 overflows; the lo halves keep 2^-21 of the row's scale), its
//     1 / ||.|| is summed while the chunks pass and applied with 1 / the staging scale when the tile is parked
//     (a . b / (||a|| ||b||) instead of (a / ||a||) . (b / ||b||));
//   * hi*hi + lo*hi + hi*lo on v_mfma_f32_32x32x16_f16, fp32 accumulate (22-bit products: the arithmetic of the loss kernels);
//   * the tile leaves through LDS: 16-byte stores, 256-byte runs per quarter wave.
// Tiles of image n run on XCD n % 8 (they share its two maps in that L2), whole tiles first, the strips of a map whose size is not a
// multiple of 128 at the end of the grid (they are short: the launch's tail is theirs).
#include "corr_common.h"
#include "host_util.h"
#include <type_traits>

namespace stego {

struct DenseFusedParams {
    MapV a, b;                  // [B,C,H1,W1], [B,C,H2,W2], channels-last (sc == 1), 16-byte aligned pixels
    float* out;                 // [B][M][N]
    int B, C, M, N, W1, W2, normalize;
    int nfa, nfb;               // whole 128-pixel blocks of a / b
    int nsa, nsb;               // 1 if a / b has a partial last block
    int groups;                 // ceil(B / 8)
};

#ifndef DF_ABL
#define DF_ABL 0                                    // (tools/ubench/dense_fused_bench.hip: compile-time timing ablations: 2 no conversion, 4 no MFMAs, 8 no loads)
#endif
constexpr int DF_SIDE = 2 * TP * LDH * 2;           // bytes of one operand side of the stage: hi[128][72] + lo[128][72] fp16
constexpr int DF_PKS = 68;                          // floats per parked row of a wave's 64 x 64 tile (272 B: conflict-free 16-byte reads)
constexpr int DF_LDS = 2 * DF_SIDE + 2 * TP * 4;    // + row scales, column scales
static_assert(4 * 64 * DF_PKS * 4 <= 2 * DF_SIDE, "the four parked wave tiles fit the stage");

template <int NCH>
__global__ void __launch_bounds__(NTHREADS, 2) dense_fused_kernel(const DenseFusedParams prm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* rowsc = reinterpret_cast<float*>(smem + 2 * DF_SIDE);          // [128] what a row of the tile is multiplied by at the park
    float* colsc = rowsc + TP;                                            // [128] ... a column
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int r = lane & 31, half = lane >> 5;
    const int q4 = lane >> 4, s16 = lane & 15;
    const int C = prm.C;

    // ---- my tile: block b runs on XCD b % 8 (observed; speed only)
    const int x = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int nfull = prm.nfa * prm.nfb;
    const int nstrip = (prm.nsb ? prm.nfa : 0) + (prm.nsa ? prm.nfb + prm.nsb : 0);
    int il, mi, nj;
    if (seq < prm.groups * nfull) {
        il = seq / nfull;
        const int t = seq - il * nfull;
        mi = t / prm.nfb; nj = t - mi * prm.nfb;
    } else {
        const int s2 = seq - prm.groups * nfull;
        il = s2 / nstrip;
        int t = s2 - il * nstrip;
        if (prm.nsb && t < prm.nfa) { mi = t; nj = prm.nfb; }
        else { t -= prm.nsb ? prm.nfa : 0; mi = prm.nfa; nj = t; }
    }
    const int n = il * 8 + x;
    if (n >= prm.B) return;
    const int rowsA = min(TP, prm.M - mi * TP), rowsB = min(TP, prm.N - nj * TP);      // pixels of either side that exist

    // ---- my eight pixels of either side: byte offsets inside the image (an image is < 2^31 bytes, host-checked); beyond the map: its last pixel
    const char* aimg = reinterpret_cast<const char*>(prm.a.p + (long long)n * prm.a.sn);
    const char* bimg = reinterpret_cast<const char*>(prm.b.p + (long long)n * prm.b.sn);
    // (dense pixel stride, host-checked: pixel p lies at p * sw)
    const unsigned asw4 = 4u * (unsigned)prm.a.sw, bsw4 = 4u * (unsigned)prm.b.sw;
    const int pa0 = mi * TP + 32 * wave + q4, pb0 = nj * TP + 32 * wave + q4;
    f32x4 ra[8], rb[8];
    auto load_chunk = [&](int c) __attribute__((always_inline)) {
        if ((DF_ABL & 8) && c > 0) return;
        const unsigned cho = 4u * (unsigned)min(64 * c + 4 * s16, C - 4);
#pragma unroll
        for (int i = 0; i < 8; ++i) ra[i] = *reinterpret_cast<const f32x4*>(aimg + ((unsigned)min(pa0 + 4 * i, prm.M - 1) * asw4 + cho));
#pragma unroll
        for (int i = 0; i < 8; ++i) rb[i] = *reinterpret_cast<const f32x4*>(bimg + ((unsigned)min(pb0 + 4 * i, prm.N - 1) * bsw4 + cho));
    };
    float ass[8], bss[8];                            // running sums of squares of my eight pixels of either side (their staging scales wait in rowsc / colsc)
    // chunk c of one side: scale, split, two 8-byte LDS stores per row; the first chunk fixes the row's staging scale, the last one leaves
    // 1 / (||row|| x staging scale) where the park finds it
    auto convert = [&](const f32x4 (&R)[8], float (&ss)[8], unsigned char* side, float* scale_out, const int rows,
                       const bool FIRST, const bool LAST) __attribute__((always_inline)) {
        if ((DF_ABL & 2) && !LAST) return;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            f32x4 v = R[i];
            if (LAST && 64 * (NCH - 1) + 4 * s16 >= C) v = f32x4{0.f, 0.f, 0.f, 0.f};          // channels beyond C (the last chunk only)
            float sc;
            if (FIRST) {
                float m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
#pragma unroll
                for (int d = 8; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
                const float s = m > 0.f ? __builtin_ldexpf(1.f, -3 - __builtin_amdgcn_frexp_expf(m)) : 1.f;
                sc = 32 * wave + 4 * i + q4 < rows ? s : 0.f;
                if (s16 == 0 && !LAST) scale_out[32 * wave + 4 * i + q4] = sc;      // (read back by my own 16-lane group only: no barrier needed)
                ss[i] = 0.f;
            } else {
                sc = scale_out[32 * wave + 4 * i + q4];
            }
            ss[i] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            unsigned h0, l0, h1, l1;
            split_f16_pair(v[0] * sc, v[1] * sc, h0, l0);
            split_f16_pair(v[2] * sc, v[3] * sc, h1, l1);
            half_t* dh = reinterpret_cast<half_t*>(side) + (32 * wave + 4 * i + q4) * LDH + 4 * s16;
            *reinterpret_cast<u32x2*>(dh) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(dh + TP * LDH) = u32x2{l0, l1};
            if (LAST) {
                float t = ss[i];
#pragma unroll
                for (int d = 8; d >= 1; d >>= 1) t += __shfl_xor(t, d, 64);
                const float inv = prm.normalize ? 1.f / fmaxf(sqrtf(t), 1e-10f) : 1.f;              // norm(), modules.py:276
                if (s16 == 0) scale_out[32 * wave + 4 * i + q4] = sc > 0.f ? inv / sc : 0.f;
            }
        }
    };

    // ---- which of my wave's 2 x 2 groups of 32 x 32 exist
    const int gi = min(2, max(0, (rowsA - 64 * wr + 31) >> 5)), gj = min(2, max(0, (rowsB - 64 * wc + 31) >> 5));
    const bool whole = gi == 2 && gj == 2;
    auto run = [&](auto wholec) __attribute__((always_inline)) {
    constexpr bool WHOLE = decltype(wholec)::value;      // all four 32 x 32 groups of my wave exist (the two forms share no registers: see dense_stream.hip)
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    constexpr int LO = TP * LDH;
    const half_t* ap = reinterpret_cast<const half_t*>(smem) + (64 * wr + r) * LDH + 8 * half;
    const half_t* bp = reinterpret_cast<const half_t*>(smem + DF_SIDE) + (64 * wc + r) * LDH + 8 * half;
    auto multiply = [&]() __attribute__((always_inline)) {
        if (DF_ABL & 4) return;
        if constexpr (WHOLE) {
#pragma unroll
            for (int ks = 0; ks < KC / 16; ++ks) {
                __builtin_amdgcn_sched_barrier(0);   // (one k-step's fragments at a time: the scheduler otherwise hoists all 32 reads)
                f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    ah[i] = *reinterpret_cast<const f16x8*>(ap + i * 32 * LDH + 16 * ks);
                    bh[i] = *reinterpret_cast<const f16x8*>(bp + i * 32 * LDH + 16 * ks);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    al[i] = *reinterpret_cast<const f16x8*>(ap + i * 32 * LDH + LO + 16 * ks);
                    bl[i] = *reinterpret_cast<const f16x8*>(bp + i * 32 * LDH + LO + 16 * ks);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (i < gi && j < gj) {
#pragma unroll
                        for (int ks = 0; ks < KC / 16; ++ks) {
                            const f16x8 ah = *reinterpret_cast<const f16x8*>(ap + i * 32 * LDH + 16 * ks);
                            const f16x8 al = *reinterpret_cast<const f16x8*>(ap + i * 32 * LDH + LO + 16 * ks);
                            const f16x8 bh = *reinterpret_cast<const f16x8*>(bp + j * 32 * LDH + 16 * ks);
                            const f16x8 bl = *reinterpret_cast<const f16x8*>(bp + j * 32 * LDH + LO + 16 * ks);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[i][j], 0, 0, 0);
                        }
                    }
                }
        }
    };
    // the chunk loop is NOT unrolled (first and last chunk peeled): unrolled, the scheduler hoisted fragment reads and conversions of several
    // chunks over each other and spilled 300-470 registers
    load_chunk(0);
    convert(ra, ass, smem, rowsc, rowsA, true, NCH == 1);
    convert(rb, bss, smem + DF_SIDE, colsc, rowsB, true, NCH == 1);
    if (NCH > 1) load_chunk(1);                      // flies under this chunk's MFMAs
    __syncthreads();                                 // the stage is whole
    multiply();
#pragma unroll 1
    for (int c = 1; c < NCH - 1; ++c) {
        __syncthreads();                             // everybody has multiplied chunk c - 1: the stage is free
        convert(ra, ass, smem, rowsc, rowsA, false, false);
        convert(rb, bss, smem + DF_SIDE, colsc, rowsB, false, false);
        load_chunk(c + 1);
        __syncthreads();
        multiply();
    }
    if (NCH > 1) {
        __syncthreads();
        convert(ra, ass, smem, rowsc, rowsA, false, true);
        convert(rb, bss, smem + DF_SIDE, colsc, rowsB, false, true);
        __syncthreads();
        multiply();
    }

    // ---- the way out: my 64 x 64 tile parked (scaled) in my quarter of the dead stage, then 16-byte stores: a lane owns 4 consecutive
    // columns of a row, 256 contiguous bytes per quarter wave.  C/D layout: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
    __syncthreads();                                 // everybody has multiplied the last chunk
    if (gi == 0 || gj == 0) return;
    float* park = reinterpret_cast<float*>(smem) + wave * (64 * DF_PKS);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float cs = colsc[64 * wc + 32 * j + r];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rl = 32 * i + (e & 3) + 8 * (e >> 2) + 4 * half;
                park[rl * DF_PKS + 32 * j + r] = acc[i][j][e] * (rowsc[64 * wr + rl] * cs);
            }
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // (one wave: its LDS operations execute in order)
    float* outn = prm.out + (size_t)n * prm.M * prm.N;
    const int row0 = mi * TP + 64 * wr, col = nj * TP + 64 * wc + 4 * s16;
    const bool v4 = (prm.N & 3) == 0;
    if (WHOLE && v4) {
        const unsigned o0 = 4u * ((unsigned)(row0 + q4) * (unsigned)prm.N + (unsigned)col), rstep = 16u * (unsigned)prm.N;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(park + (4 * k + q4) * DF_PKS + 4 * s16);
            __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(reinterpret_cast<char*>(outn) + (o0 + k * rstep)));
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int row = row0 + 4 * k + q4;
            const f32x4 v = *reinterpret_cast<const f32x4*>(park + (4 * k + q4) * DF_PKS + 4 * s16);
            if (row < prm.M) {
                float* o = outn + (size_t)row * prm.N + col;
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
    if (whole) run(std::true_type{}); else run(std::false_type{});
}

// the caller (launch_dense_corr) has checked: channels-last maps, 16-byte aligned pixels, C % 8 == 0, C <= 768, images below 2^31 bytes
hipError_t launch_dense_fused(const MapV& a, const MapV& b, int B, int C, int H1, int W1, int H2, int W2, int normalize, float* out, hipStream_t stream)
{
    DenseFusedParams prm{};
    prm.a = a; prm.b = b; prm.out = out;
    prm.B = B; prm.C = C; prm.M = H1 * W1; prm.N = H2 * W2; prm.W1 = W1; prm.W2 = W2;
    prm.normalize = normalize;
    prm.nfa = prm.M / TP; prm.nfb = prm.N / TP;
    prm.nsa = prm.M % TP ? 1 : 0; prm.nsb = prm.N % TP ? 1 : 0;
    prm.groups = (B + 7) / 8;
    const int tiles = (prm.nfa + prm.nsa) * (prm.nfb + prm.nsb);
    const dim3 grid((unsigned)(8 * prm.groups * tiles));
    const int NCH = (C + KC - 1) / KC;
#define STEGO_DF(N_)                                                                                                    \
    case N_: {                                                                                                          \
        hipError_t e_ = ensure_dynamic_lds(reinterpret_cast<const void*>(&dense_fused_kernel<N_>), DF_LDS);             \
        if (e_ != hipSuccess) return e_;                                                                                \
        hipLaunchKernelGGL(dense_fused_kernel<N_>, grid, dim3(NTHREADS), DF_LDS, stream, prm);                          \
        break;                                                                                                          \
    }
    switch (NCH) {
#ifndef DF_ONLY6
        STEGO_DF(1) STEGO_DF(2) STEGO_DF(3) STEGO_DF(4) STEGO_DF(5) STEGO_DF(7) STEGO_DF(8) STEGO_DF(9) STEGO_DF(10) STEGO_DF(11) STEGO_DF(12)
#endif
        STEGO_DF(6)
        default: return hipErrorInvalidValue;
    }
#undef STEGO_DF
    return hipGetLastError();
}

}  // namespace stego
