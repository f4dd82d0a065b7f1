// Dense feature correspondence  out[n,h,w,i,j] = sum_c a[n,c,h,w] * b[n,c,i,j]  (gfx950).
//
//   reference: tensor_correlation() src/modules.py:283-284 = einsum("nchw,ncij->nhwij") - the literal north-star
//              tensor; used at full map resolution by plot_dino_correspondence.py:39-58 / plot_pr_curves.py:108-121
//              (optionally on norm()'ed maps, :275-276).  SURVEY.md 8(f) rank 3: the MFMA-bound regime
//              ([B, 784, 784] per pair at 224^2 / 8).
//
// Two launches:
//   dense_prep_kernel - any-stride [B,C,H,W] map -> per image, per 128-pixel block, per 64-channel chunk an LDS
//                       operand image [hi|lo][128][72] of fp16 halves (optional L2 normalisation over C, zero padded):
//                       the same split-fp16 layout the loss and KNN kernels multiply.
//   dense_tile_kernel - one workgroup per 128x128 output tile of one image: both operands are streamed with
//                       global_load_lds (double-buffered), hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 (22-bit
//                       products, fp32 accumulate), the tile is stored straight from the accumulators (each wave
//                       instruction writes 128-byte row segments).  Output-write bound at C = 384.
#include "corr_common.h"
#include "host_util.h"

namespace stego {

hipError_t launch_rowsum(const float* x, float* out, long long rows, int P, hipStream_t stream);          // loss_pointwise.hip
hipError_t launch_dense_stream(const MapV& a, const MapV& b, int B, int C, int H1, int W1, int H2, int W2, int normalize, float* out, void* ws,
                               hipStream_t stream);                                                    // dense_stream.hip

constexpr int DC_SIDE = 2 * TP * LDH * 2;          // bytes of one chunk image: hi[128][72] + lo[128][72] fp16
constexpr int DT_PKS = 68;                         // floats per row of a wave's parked 32 x 64 half quadrant (272 B: conflict-free 16-byte reads along a row)

struct DenseParams {
    MapV a, b;                  // [B,C,H1,W1], [B,C,H2,W2]
    void* imgA;                 // [B][nbA][NCH][2][128][LDH] fp16
    void* imgB;
    float* out;                 // [B][M][N]
    float* rsA;                 // [B][nbA*128] 1 / (power-of-two row scale) of the fp16 staging; rsB likewise
    float* rsB;
    int B, C, M, N, W1, W2, nbA, nbB, NCH, normalize;
    float* out1;                // seg > 0: images [seg, 2 seg) go to out1, [2 seg, B) to out2 (the three cd outputs of the loss)
    float* out2;
    int seg;
    float* rowsum;              // optional [B][M]: sum_j out[n][i][j] (row-block kernel with prepared A operands only: a workgroup sees whole rows)
    int a_mod;                  // > 0: pair n multiplies the A image n % a_mod (one set of A operands against several sets of B: the pair-sets of the loss)
};

// grid = (B * (nbA + nbB)), block = 256 = 8 half-waves; a half-wave owns 16 of the block's 128 pixel rows, one at a time.
// Channels-last maps (channel stride 1, 16-byte aligned pixels, C % 4 == 0, C <= 1024): lane hl holds channels
// 4 hl + 128 j of the row in registers (16-byte loads, one coalesced run per row), the norm is a 5-step shuffle
// reduction, hi/lo halves go out as 8-byte stores.  Anything else: scalar loads, two passes over the row.
__global__ void __launch_bounds__(NTHREADS) dense_prep_kernel(const DenseParams prm, const int vec)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int hl = lane & 31, hw = tid >> 5;                // half-wave 0..7
    const int per_img = prm.nbA + prm.nbB;
    const int n = blockIdx.x / per_img;
    int blk = blockIdx.x - n * per_img;
    const bool isB = blk >= prm.nbA;
    if (isB) blk -= prm.nbA;
    const MapV m = isB ? prm.b : prm.a;
    const int P = isB ? prm.N : prm.M, W = isB ? prm.W2 : prm.W1;
    const int C = prm.C;
    half_t* base = static_cast<half_t*>(isB ? prm.imgB : prm.imgA) +
                   ((size_t)n * (isB ? prm.nbB : prm.nbA) + blk) * prm.NCH * (2 * TP * LDH);
    constexpr int MAXJ = 8;
    for (int it = 0; it < 16; ++it) {
        const int rl = it * 8 + hw;
        const int pix = blk * TP + rl;
        const bool rv = pix < P;
        const int hh = rv ? pix / W : 0, ww = rv ? pix - hh * W : 0;
        const float* x = m.p + (long long)n * m.sn + (long long)hh * m.sh + (long long)ww * m.sw;
        if (vec) {
            f32x4 v[MAXJ];
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < MAXJ; ++j) {
                const int c = 128 * j + 4 * hl;
                v[j] = (rv && c < C) ? *reinterpret_cast<const f32x4*>(x + c) : f32x4{0.f, 0.f, 0.f, 0.f};
                ss += v[j][0] * v[j][0] + v[j][1] * v[j][1] + v[j][2] * v[j][2] + v[j][3] * v[j][3];
            }
#pragma unroll
            for (int s2 = 16; s2 >= 1; s2 >>= 1) ss += __shfl_xor(ss, s2, 64);
            float inv = prm.normalize ? 1.f / fmaxf(sqrtf(ss), 1e-10f) : 1.f;            // norm(), modules.py:276
            float mx = 0.f;
#pragma unroll
            for (int j = 0; j < MAXJ; ++j) mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[j][0]), fabsf(v[j][1])), fmaxf(fabsf(v[j][2]), fabsf(v[j][3]))));
#pragma unroll
            for (int s2 = 16; s2 >= 1; s2 >>= 1) mx = fmaxf(mx, __shfl_xor(mx, s2, 64));
            // fp16 staging wants |x| ~ 1: a power-of-two row scale (exact), divided out again by the tile kernel
            const float rs = mx * inv > 0.f ? __builtin_ldexpf(1.f, -__builtin_amdgcn_frexp_expf(mx * inv)) : 1.f;
            inv *= rs;
            if (hl == 0) (isB ? prm.rsB : prm.rsA)[((size_t)n * (isB ? prm.nbB : prm.nbA) + blk) * TP + rl] = 1.f / rs;
#pragma unroll
            for (int j = 0; j < MAXJ; ++j) {
                const int c = 128 * j + 4 * hl;
                if (c < prm.NCH * KC) {                                                  // zero padding up to the chunk end
                    half_t* dh = base + (size_t)(c >> 6) * (2 * TP * LDH) + rl * LDH + (c & 63);
                    unsigned h0, l0, h1, l1;
                    split_f16_pair(v[j][0] * inv, v[j][1] * inv, h0, l0);
                    split_f16_pair(v[j][2] * inv, v[j][3] * inv, h1, l1);
                    *reinterpret_cast<u32x2*>(dh) = u32x2{h0, h1};
                    *reinterpret_cast<u32x2*>(dh + TP * LDH) = u32x2{l0, l1};
                }
            }
        } else {
            float ss = 0.f, mx = 0.f;
            if (rv) for (int c = hl; c < C; c += 32) { const float t = x[(long long)c * m.sc]; ss += t * t; mx = fmaxf(mx, fabsf(t)); }
#pragma unroll
            for (int s2 = 16; s2 >= 1; s2 >>= 1) { ss += __shfl_xor(ss, s2, 64); mx = fmaxf(mx, __shfl_xor(mx, s2, 64)); }
            float inv = prm.normalize ? 1.f / fmaxf(sqrtf(ss), 1e-10f) : 1.f;
            const float rs = mx * inv > 0.f ? __builtin_ldexpf(1.f, -__builtin_amdgcn_frexp_expf(mx * inv)) : 1.f;
            inv *= rs;
            if (hl == 0) (isB ? prm.rsB : prm.rsA)[((size_t)n * (isB ? prm.nbB : prm.nbA) + blk) * TP + rl] = 1.f / rs;
            for (int c = hl; c < prm.NCH * KC; c += 32) {
                const float t = (rv && c < C) ? x[(long long)c * m.sc] * inv : 0.f;
                unsigned h, l;
                split_f16_pair(t, 0.f, h, l);
                half_t* dh = base + (size_t)(c >> 6) * (2 * TP * LDH) + rl * LDH + (c & 63);
                *reinterpret_cast<unsigned short*>(dh) = (unsigned short)(h & 0xffffu);
                *reinterpret_cast<unsigned short*>(dh + TP * LDH) = (unsigned short)(l & 0xffffu);
            }
        }
    }
}

// grid = (nbB, nbA, B); block = 256 (4 waves, 2x2 quadrants of 64x64); LDS = 2 stages x (A chunk + B chunk)
__global__ void __launch_bounds__(NTHREADS) dense_tile_kernel(const DenseParams prm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int nj = blockIdx.x, mi = blockIdx.y, n = blockIdx.z;
    const int NCH = prm.NCH;
    const int na = prm.a_mod > 0 ? n % prm.a_mod : n;
    const unsigned char* A = static_cast<const unsigned char*>(prm.imgA) + ((size_t)na * prm.nbA + mi) * NCH * DC_SIDE;
    const unsigned char* Bm = static_cast<const unsigned char*>(prm.imgB) + ((size_t)n * prm.nbB + nj) * NCH * DC_SIDE;
    auto issue = [&](int c) {
        unsigned char* dst = smem;
        for (int pc = wave; pc < DC_SIDE / 1024; pc += 4) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A + (size_t)c * DC_SIDE + (size_t)pc * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(dst + pc * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Bm + (size_t)c * DC_SIDE + (size_t)pc * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(dst + DC_SIDE + pc * 1024), 16, 0, 0);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    constexpr int LO = TP * LDH;
    const int r = lane & 31, half = lane >> 5;
    for (int c = 0; c < NCH; ++c) {
        // ONE stage per workgroup (73.7 KB) and two workgroups per CU that cover each other's copies - as the backbone's GEMM does; with two
        // stages and one workgroup per CU a tile took 47 us for 7.7 us of MFMA time at C = 768 (round 5: 0.99 -> see profiles/r05i)
        if (c > 0) __syncthreads();                       // everyone is done reading chunk c - 1
        issue(c);
        sync_after_lds_dma();                             // chunk c landed
        const half_t* As = reinterpret_cast<const half_t*>(smem);
        const half_t* Bs = As + DC_SIDE / 2;
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
    // ---- store through LDS (the stage is free behind a barrier): a wave parks half of its 64 x 64 quadrant at a time and a lane then owns 4
    // consecutive columns of a row - 16-byte stores, 256 contiguous bytes per quarter wave (4-byte stores in 128-byte pieces before)
    float* out = prm.out + (size_t)n * prm.M * prm.N;
    const float* ra = prm.rsA + ((size_t)na * prm.nbA + mi) * TP;      // undo the rows' staging scales
    const float* rb = prm.rsB + ((size_t)n * prm.nbB + nj) * TP;
    __syncthreads();
    float* park = reinterpret_cast<float*>(smem) + wave * (32 * DT_PKS);
    const int c4 = 4 * (lane & 15);
    const bool v4 = (prm.N & 3) == 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float sb = rb[64 * wc + 32 * j + (lane & 31)];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rl = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                park[rl * DT_PKS + 32 * j + (lane & 31)] = acc[i][j][e] * (ra[64 * wr + 32 * i + rl] * sb);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // (one wave: its LDS operations execute in order)
        const int col = nj * TP + 64 * wc + c4;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int rl = 4 * k + (lane >> 4);
            const int row = mi * TP + 64 * wr + 32 * i + rl;
            const f32x4 v = *reinterpret_cast<const f32x4*>(park + rl * DT_PKS + c4);
            if (row < prm.M) {
                float* o = out + (size_t)row * prm.N + col;
                if (v4 && col < prm.N) {
                    *reinterpret_cast<f32x4*>(o) = v;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (col + q < prm.N) o[q] = v[q];
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // the reads are done before the next half overwrites the slab
    }
}

// ---------------------------------------------------------------------------------------------- round 4: row-block kernel
// dense_rowblock_kernel (channels-last A maps, C <= 384): one PERSISTENT workgroup per (image, 128-pixel block of A).  The tile kernel
// above stages both operands of every 128 x 128 tile through LDS with one chunk of look-ahead (6.1 rounds of one-tile workgroups, each
// chunk waiting for its copy: ~20 us per tile slot) after a prep pass over BOTH maps.  Here the A block never touches LDS or the
// workspace: the workgroup reads its 128 pixels straight from the fp32 map (each row by the two lanes that will own it in the MFMA A
// layout), normalises, applies the power-of-two row scale and splits into fp16 hi / lo MFMA fragments IN REGISTERS (192 VGPRs, as the
// KNN kernel's query block), then walks the B blocks of its image: chunks of the prepared B image through a three-stage LDS ring with
// two copies in flight (LDS-DMA from inline asm, counted waits), 48 MFMAs per chunk and wave, the 32 x 128 slab of a wave stored from
// the accumulators while the next block's copies fly.  Workgroups of image n sit on XCD n % 8 (they stream the same B image).  The prep
// kernel runs for the B side only.
constexpr int DR_NST = 3;
constexpr int DR_MAXCH = 6;
constexpr int DR_PKS = 68;        // floats per row of a wave's parked 32 x 64 half slab (272 B: conflict-free 16-byte reads along a row)
static_assert(4 * 32 * DR_PKS * 4 <= DC_SIDE, "the parked half slabs of four waves fit one ring slot");
static_assert(DC_SIDE % 4096 == 0, "whole 1 KB pieces per wave");
__device__ __forceinline__ void dense_dma_piece(const unsigned char* gsrc_lane, unsigned lds_addr)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc_lane), "s"(lds_addr) : "memory");
}

// APANELS (round 5): the A block is a prepared operand image too (stego_sample_panels wrote it: the sampled points of the loss) - the
// fragments are 16-byte loads of its hi / lo planes.
// NST = stages of the B ring: 3 (two copies in flight, one workgroup per CU) or 2 (one in flight, 74 KB: TWO workgroups per CU that hide each
// other's copies and store drains - the loss's point sets, where a workgroup has 2 x 2 .. 2 x 6 chunks between 128 KB of stores).
template <bool APANELS, int NST = 3>
__global__ void __launch_bounds__(NTHREADS) dense_rowblock_kernel(const DenseParams prm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* ra_s = reinterpret_cast<float*>(smem + NST * DC_SIDE);        // [128] 1 / row scale of the A block
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NCH = prm.NCH, C = prm.C;
    // image n on XCD n % 8 (block b runs on XCD b % 8: observed, speed only)
    const int x = blockIdx.x & 7, sl = blockIdx.x >> 3;
    const int n = (sl / prm.nbA) * 8 + x, mi = sl % prm.nbA;
    if (n >= prm.B) return;
    const int r = lane & 31, half = lane >> 5;

    // ---- my row of the A block -> MFMA fragments (lane (r, half) holds channels 16 ks + 8 half .. + 7 of every 64-channel chunk)
    f16x8 Ah[DR_MAXCH][KC / 16], Al[DR_MAXCH][KC / 16];
    if constexpr (APANELS) {
        const int na = prm.a_mod > 0 ? n % prm.a_mod : n;
        const half_t* ap = static_cast<const half_t*>(prm.imgA) + ((size_t)na * prm.nbA + mi) * NCH * (2 * TP * LDH) + (32 * wave + r) * LDH + 8 * half;
        if (half == 0) ra_s[32 * wave + r] = prm.rsA[((size_t)na * prm.nbA + mi) * TP + 32 * wave + r];
#pragma unroll
        for (int c = 0; c < DR_MAXCH; ++c)
#pragma unroll
            for (int ks = 0; ks < KC / 16; ++ks) {
                if (c < NCH && !(prm.normalize & 1)) {
                    Ah[c][ks] = *reinterpret_cast<const f16x8*>(ap + (size_t)c * (2 * TP * LDH) + 16 * ks);
                    Al[c][ks] = *reinterpret_cast<const f16x8*>(ap + (size_t)c * (2 * TP * LDH) + TP * LDH + 16 * ks);
                } else {
                    Ah[c][ks] = f16x8{};
                    Al[c][ks] = f16x8{};
                }
            }
    } else {
    // (round 6) The block comes in coalesced - lane (q, s) reads 16 bytes of eight rows per 64-channel chunk, four whole 256-byte runs per
    // wave instruction, every chunk in flight at once, nothing behind a branch -, gets its row statistics by shuffles inside the 16-lane
    // groups and passes through the (still idle) ring slots in the split-fp16 chunk layout, from which every lane takes its fragments with
    // conflict-free 16-byte reads.  Before, each lane read its own row in 32-byte pieces, twice, behind a branch and a full wait per
    // piece: ~25 serialized cold round trips and every line pulled into the L1 four times - 40 of the kernel's 93 us (stamps,
    // tools/exp/r6_dense_stamps.py on the streaming variant of this kernel).  Rows beyond M read the map's last pixel times zero.
    const int q4 = lane >> 4, s16 = lane & 15;
    const float* aimg = prm.a.p + (long long)n * prm.a.sn;
    f32x4 ar[DR_MAXCH][8];
    const float* arow[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int pc = min(mi * TP + 32 * wave + 4 * i + q4, prm.M - 1);
        const int hh = pc / prm.W1, ww = pc - hh * prm.W1;
        arow[i] = aimg + (long long)hh * prm.a.sh + (long long)ww * prm.a.sw;
    }
#pragma unroll
    for (int c = 0; c < DR_MAXCH; ++c) {
        const int ch = min(64 * c + 4 * s16, C - 4);           // (chunks / channels beyond C: a valid address, masked below)
#pragma unroll
        for (int i = 0; i < 8; ++i) ar[c][i] = *reinterpret_cast<const f32x4*>(arow[i] + ch);
    }
    float ainv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float ss = 0.f, mx = 0.f;
#pragma unroll
        for (int c = 0; c < DR_MAXCH; ++c) {
            const float m = 64 * c + 4 * s16 < C ? 1.f : 0.f;
            const f32x4 v = ar[c][i];
            ss += m * ((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]));
            mx = fmaxf(mx, m * fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
        }
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) { ss += __shfl_xor(ss, d, 64); mx = fmaxf(mx, __shfl_xor(mx, d, 64)); }
        float inv = prm.normalize ? 1.f / fmaxf(sqrtf(ss), 1e-10f) : 1.f;        // norm(), modules.py:276
        const float rs = mx * inv > 0.f ? __builtin_ldexpf(1.f, -__builtin_amdgcn_frexp_expf(mx * inv)) : 1.f;
        const int rl = 32 * wave + 4 * i + q4;
        ainv[i] = mi * TP + rl < prm.M ? inv * rs : 0.f;
        if (s16 == 0) ra_s[rl] = 1.f / rs;
    }
#pragma unroll
    for (int c = 0; c < DR_MAXCH; ++c) {
        unsigned char* stage = smem + (c & 1) * DC_SIDE;
        if (c < NCH) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float sc = 64 * c + 4 * s16 < C ? ainv[i] : 0.f;
                unsigned h0, l0, h1, l1;
                split_f16_pair(ar[c][i][0] * sc, ar[c][i][1] * sc, h0, l0);
                split_f16_pair(ar[c][i][2] * sc, ar[c][i][3] * sc, h1, l1);
                half_t* dh = reinterpret_cast<half_t*>(stage) + (32 * wave + 4 * i + q4) * LDH + 4 * s16;
                *reinterpret_cast<u32x2*>(dh) = u32x2{h0, h1};
                *reinterpret_cast<u32x2*>(dh + TP * LDH) = u32x2{l0, l1};
            }
        }
        __syncthreads();                             // the chunk is whole (and everybody has read the slot's previous tenant, two chunks back)
        const half_t* ap = reinterpret_cast<const half_t*>(stage) + (32 * wave + r) * LDH + 8 * half;
#pragma unroll
        for (int ks = 0; ks < KC / 16; ++ks) {
            if (c < NCH) {
                Ah[c][ks] = *reinterpret_cast<const f16x8*>(ap + 16 * ks);
                Al[c][ks] = *reinterpret_cast<const f16x8*>(ap + TP * LDH + 16 * ks);
            } else {
                Ah[c][ks] = f16x8{};
                Al[c][ks] = f16x8{};
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __syncthreads();                                       // ra_s
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the counted waits below start from zero

    // ---- the B blocks of image n: chunk stream g = nj * NCH + c
    const unsigned char* Bimg = static_cast<const unsigned char*>(prm.imgB) + (size_t)n * prm.nbB * NCH * DC_SIDE;
    const unsigned smem_addr = (unsigned)(uintptr_t)((__attribute__((address_space(3))) const unsigned char*)smem);
    const int nstage = prm.nbB * NCH;
    auto issue = [&](int g) {
        const unsigned char* src = Bimg + (size_t)g * DC_SIDE + lane * 16;
        const unsigned dst = smem_addr + (g % NST) * DC_SIDE;
        if constexpr (APANELS && NST == 2) {
            // a last block with few points (144 = 128 + 16): only the rows that exist, of the hi and of the lo plane (whole 1 KB pieces)
            const int rvb = min(TP, prm.N - (g / NCH) * TP);
            const int np = (rvb * LDH * 2 + 1023) >> 10;               // pieces per plane, <= 18
            for (int pc = wave; pc < 2 * np; pc += 4) {
                const int o = (pc < np ? pc : pc - np + DC_SIDE / 2048) * 1024;
                dense_dma_piece(src + o, dst + o);
            }
        } else {
#pragma unroll
            for (int i = 0; i < DC_SIDE / 4096; ++i) dense_dma_piece(src + (wave + 4 * i) * 1024, dst + (wave + 4 * i) * 1024);
        }
    };
    for (int s0 = 0; s0 < NST - 1 && s0 < nstage; ++s0) issue(s0);
    float* outn = prm.out + (size_t)n * prm.M * prm.N;
    if (prm.seg > 0 && n >= prm.seg) outn = n < 2 * prm.seg ? prm.out1 + (size_t)(n - prm.seg) * prm.M * prm.N : prm.out2 + (size_t)(n - 2 * prm.seg) * prm.M * prm.N;
    constexpr int LO = TP * LDH;
    int g = 0;
    float rsum[8];                     // (APANELS) row 4 k + (lane >> 4) of the wave's 32
#pragma unroll
    for (int e = 0; e < 8; ++e) rsum[e] = 0.f;
    for (int nj = 0; nj < prm.nbB; ++nj) {
        f32x16 acc[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ni][e] = 0.f;
#pragma unroll
        for (int c = 0; c < DR_MAXCH; ++c) {
            if (c < NCH) {
                {   // my pieces of stage g have landed: at most min(NST - 2, stages left) later stages (9 pieces per wave each) may still fly
                    const int ahead = NST >= 3 ? min(nstage - 1 - g, NST - 2) : 0;
                    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
                    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (g + NST - 1 < nstage) issue(g + NST - 1);
                const half_t* bp = reinterpret_cast<const half_t*>(smem + (g % NST) * DC_SIDE) + r * LDH + 8 * half;
                // (prepared operands: 32-column groups beyond the block's points and waves whose rows are all beyond M have nothing to multiply)
                // (round 6: the map variant too - a 784-pixel map ends 16 pixels into its seventh block: three of that block's four column
                // groups and three of the last row block's four waves multiplied padding, 20 % of the launch's MFMAs and fragment reads)
                const int nlive = (mi * TP + 32 * wave < prm.M) ? (min(TP, prm.N - nj * TP) + 31) >> 5 : 0;
                if (nlive == 4) {                                      // (the full block: nothing predicated between the MFMAs)
#pragma unroll
                    for (int ks = 0; ks < KC / 16; ++ks) {
                        if (APANELS && (prm.normalize & 4)) break;
                        f16x8 bh[4], bl[4];
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni) {
                            bh[ni] = *reinterpret_cast<const f16x8*>(bp + ni * 32 * LDH + 16 * ks);
                            bl[ni] = *reinterpret_cast<const f16x8*>(bp + ni * 32 * LDH + LO + 16 * ks);
                        }
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[c][ks], bh[ni], acc[ni], 0, 0, 0);
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[c][ks], bl[ni], acc[ni], 0, 0, 0);
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[c][ks], bh[ni], acc[ni], 0, 0, 0);
                    }
                } else {
                    for (int ni = 0; ni < nlive; ++ni) {
                        f32x16 a2 = ni == 0 ? acc[0] : ni == 1 ? acc[1] : acc[2];
#pragma unroll
                        for (int ks = 0; ks < KC / 16; ++ks) {
                            const f16x8 bh = *reinterpret_cast<const f16x8*>(bp + ni * 32 * LDH + 16 * ks);
                            const f16x8 bl = *reinterpret_cast<const f16x8*>(bp + ni * 32 * LDH + LO + 16 * ks);
                            a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[c][ks], bh, a2, 0, 0, 0);
                            a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[c][ks], bl, a2, 0, 0, 0);
                            a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[c][ks], bh, a2, 0, 0, 0);
                        }
                        if (ni == 0) acc[0] = a2; else if (ni == 1) acc[1] = a2; else acc[2] = a2;
                    }
                }
                ++g;
            }
        }
        // ---- my 32 x 128 slab.  C/D layout: col = lane & 31 (+ 32 ni), row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
        const float* rb = prm.rsB + ((size_t)n * prm.nbB + nj) * TP;
        {
            // through LDS (round 5; the map variant too: the 78.7 MB correspondence tensor of a 28 x 28 pair batch left as 4-byte stores before): a lane then owns 4 consecutive columns of a row - 16-byte stores, 256 contiguous bytes per quarter wave - instead of
            // 4-byte stores in 128-byte pieces (59 MB left at 2.2 TB/s that way: tools/exp/r5_rowblock_abl.py), and the row sums of the loss's
            // pointwise shift (modules.py:332) are one 4-step reduction per row.  The slab is parked in the ring slot the block's last chunk was
            // read from (free until the next copy is issued, behind the next barrier), in two passes of 64 columns
            __builtin_amdgcn_s_barrier();                                // everybody has read the last chunk
            float* park = reinterpret_cast<float*>(smem + ((g - 1) % NST) * DC_SIDE) + wave * (32 * DR_PKS);
            const int c4 = 4 * (lane & 15);
            const bool v4 = (prm.N & 3) == 0;
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                if (nj * TP + 64 * pass >= prm.N || mi * TP + 32 * wave >= prm.M) continue;          // nothing of this half slab exists
#pragma unroll
                for (int nh = 0; nh < 2; ++nh) {
                    const int ni = 2 * pass + nh;
                    const float sb = rb[32 * ni + (lane & 31)];
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int rl = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                        park[rl * DR_PKS + 32 * nh + (lane & 31)] = acc[ni][e] * (ra_s[32 * wave + rl] * sb);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // (one wave: its LDS operations execute in order)
                const int col = nj * TP + 64 * pass + c4;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int rl = 4 * k + (lane >> 4);
                    const int row = mi * TP + 32 * wave + rl;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(park + rl * DR_PKS + c4);
                    float sm = 0.f;
                    if (row < prm.M && !(prm.normalize & 2)) {
                        float* o = outn + (size_t)row * prm.N + col;
                        if (v4 && col < prm.N) {
                            __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(o));
                            sm = (v[0] + v[1]) + (v[2] + v[3]);
                        } else {
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                if (col + q < prm.N) { __builtin_nontemporal_store(v[q], o + q); sm += v[q]; }
                        }
                    }
#pragma unroll
                    for (int m = 8; m >= 1; m >>= 1) sm += __shfl_xor(sm, m, 64);
                    rsum[k] += sm;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the reads are done before the next pass overwrites the slab
            }
        }
        // (the stores and the rb loads above are vector-memory operations too: they complete in order IN FRONT of the copies issued
        // after them only if none of those is waited for by count - the copies of the next chunks were issued before them, so the
        // counted waits stay valid once these are drained)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if constexpr (APANELS) {
        if (prm.rowsum) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int row = mi * TP + 32 * wave + 4 * k + (lane >> 4);
                if ((lane & 15) == 0 && row < prm.M) prm.rowsum[(size_t)n * prm.M + row] = rsum[k];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- round 5: two row blocks per workgroup
// dense_rowpair_kernel (operands of <= MAXCH = 2 chunks: the codes - eight waves have 256 registers each, the 192 of a C = 384 row do not fit):
// the row-block kernel for prepared A operands with EIGHT waves - 256 rows of A in registers (two 128-row blocks, 32 rows
// per wave), the B chunks of the image streamed ONCE for both (the point sets of the loss have two row blocks per image: each B chunk was
// read by two workgroups before), two waves per SIMD that cover each other's LDS reads.  Two ring stages (one copy in flight), the output
// slabs of all eight waves parked in a region of their own (no barrier in front of the park), 16-byte stores, row sums as above.
constexpr int DR2_THREADS = 512;
template <int MAXCH>
__global__ void __launch_bounds__(DR2_THREADS) dense_rowpair_kernel(const DenseParams prm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* ra_s = reinterpret_cast<float*>(smem + 2 * DC_SIDE);              // [256] 1 / row scale of the two A blocks
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NCH = prm.NCH;
    const int nsb = (prm.nbA + 1) >> 1;                                      // row super-blocks per image
    const int x = blockIdx.x & 7, sl = blockIdx.x >> 3;
    const int n = (sl / nsb) * 8 + x, sb = sl % nsb;
    if (n >= prm.B) return;
    const int r = lane & 31, half = lane >> 5;
    const int mi = 2 * sb + (wave >> 2), wq = wave & 3;                      // my 128-row block and my 32 rows of it
    const bool live = mi < prm.nbA && mi * TP + 32 * wq < prm.M;            // (a wave whose rows are all beyond M only keeps the barriers)
    const int na = prm.a_mod > 0 ? n % prm.a_mod : n;
    f16x8 Ah[MAXCH][KC / 16], Al[MAXCH][KC / 16];
    {
        const int mia = mi < prm.nbA ? mi : 0;
        const half_t* ap = static_cast<const half_t*>(prm.imgA) + ((size_t)na * prm.nbA + mia) * NCH * (2 * TP * LDH) + (32 * wq + r) * LDH + 8 * half;
        if (half == 0) ra_s[32 * wave + r] = prm.rsA[((size_t)na * prm.nbA + mia) * TP + 32 * wq + r];
#pragma unroll
        for (int c = 0; c < MAXCH; ++c)
#pragma unroll
            for (int ks = 0; ks < KC / 16; ++ks) {
                if (c < NCH && live) {
                    Ah[c][ks] = *reinterpret_cast<const f16x8*>(ap + (size_t)c * (2 * TP * LDH) + 16 * ks);
                    Al[c][ks] = *reinterpret_cast<const f16x8*>(ap + (size_t)c * (2 * TP * LDH) + TP * LDH + 16 * ks);
                } else {
                    Ah[c][ks] = f16x8{};
                    Al[c][ks] = f16x8{};
                }
            }
    }
    __syncthreads();                                       // ra_s
    const unsigned char* Bimg = static_cast<const unsigned char*>(prm.imgB) + (size_t)n * prm.nbB * NCH * DC_SIDE;
    const unsigned smem_addr = (unsigned)(uintptr_t)((__attribute__((address_space(3))) const unsigned char*)smem);
    const int nstage = prm.nbB * NCH;
    auto issue = [&](int g) {
        const unsigned char* src = Bimg + (size_t)g * DC_SIDE + lane * 16;
        const unsigned dst = smem_addr + (g & 1) * DC_SIDE;
        const int rvb = min(TP, prm.N - (g / NCH) * TP);               // (a last block with few points: only the rows that exist, per plane)
        const int np = (rvb * LDH * 2 + 1023) >> 10;
        for (int pc = wave; pc < 2 * np; pc += DR2_THREADS / 64) {
            const int o = (pc < np ? pc : pc - np + DC_SIDE / 2048) * 1024;
            dense_dma_piece(src + o, dst + o);
        }
    };
    issue(0);
    float* outn = prm.out + (size_t)n * prm.M * prm.N;
    if (prm.seg > 0 && n >= prm.seg) outn = n < 2 * prm.seg ? prm.out1 + (size_t)(n - prm.seg) * prm.M * prm.N : prm.out2 + (size_t)(n - 2 * prm.seg) * prm.M * prm.N;
    constexpr int LO = TP * LDH;
    float* park = reinterpret_cast<float*>(smem + 2 * DC_SIDE + 1024) + wave * (32 * DR_PKS);
    int g = 0;
    float rsum[8];                     // row 4 k + (lane >> 4) of the wave's 32
#pragma unroll
    for (int e = 0; e < 8; ++e) rsum[e] = 0.f;
    for (int nj = 0; nj < prm.nbB; ++nj) {
        f32x16 acc[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ni][e] = 0.f;
#pragma unroll
        for (int c = 0; c < MAXCH; ++c) {
            if (c < NCH) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // chunk g landed (and this wave's stores of the previous block left)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (g + 1 < nstage) issue(g + 1);
                if (live) {
                    const half_t* bp = reinterpret_cast<const half_t*>(smem + (g & 1) * DC_SIDE) + r * LDH + 8 * half;
                    const int nlive = (min(TP, prm.N - nj * TP) + 31) >> 5;          // 32-column groups of this block that exist
                    if (nlive == 4) {
#pragma unroll
                        for (int ks = 0; ks < KC / 16; ++ks) {
                            f16x8 bh[4], bl[4];
#pragma unroll
                            for (int ni = 0; ni < 4; ++ni) {
                                bh[ni] = *reinterpret_cast<const f16x8*>(bp + ni * 32 * LDH + 16 * ks);
                                bl[ni] = *reinterpret_cast<const f16x8*>(bp + ni * 32 * LDH + LO + 16 * ks);
                            }
#pragma unroll
                            for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[c][ks], bh[ni], acc[ni], 0, 0, 0);
#pragma unroll
                            for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[c][ks], bl[ni], acc[ni], 0, 0, 0);
#pragma unroll
                            for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[c][ks], bh[ni], acc[ni], 0, 0, 0);
                        }
                    } else {
                        for (int ni = 0; ni < nlive; ++ni) {
                            f32x16 a2 = ni == 0 ? acc[0] : ni == 1 ? acc[1] : acc[2];
#pragma unroll
                            for (int ks = 0; ks < KC / 16; ++ks) {
                                const f16x8 bh = *reinterpret_cast<const f16x8*>(bp + ni * 32 * LDH + 16 * ks);
                                const f16x8 bl = *reinterpret_cast<const f16x8*>(bp + ni * 32 * LDH + LO + 16 * ks);
                                a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[c][ks], bh, a2, 0, 0, 0);
                                a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[c][ks], bl, a2, 0, 0, 0);
                                a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[c][ks], bh, a2, 0, 0, 0);
                            }
                            if (ni == 0) acc[0] = a2; else if (ni == 1) acc[1] = a2; else acc[2] = a2;
                        }
                    }
                }
                ++g;
            }
        }
        if (!live) continue;
        // ---- my 32 x 128 slab through the wave's own park region, 64 columns at a time: 16-byte stores, 256 contiguous bytes per quarter wave
        const float* rb = prm.rsB + ((size_t)n * prm.nbB + nj) * TP;
        const int c4 = 4 * (lane & 15);
        const bool v4 = (prm.N & 3) == 0;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if (nj * TP + 64 * pass >= prm.N) continue;                      // nothing of this half slab exists
#pragma unroll
            for (int nh = 0; nh < 2; ++nh) {
                const int ni = 2 * pass + nh;
                const float sb2 = rb[32 * ni + (lane & 31)];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int rl = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    park[rl * DR_PKS + 32 * nh + (lane & 31)] = acc[ni][e] * (ra_s[32 * wave + rl] * sb2);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // (one wave: its LDS operations execute in order)
            const int col = nj * TP + 64 * pass + c4;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int rl = 4 * k + (lane >> 4);
                const int row = mi * TP + 32 * wq + rl;
                const f32x4 v = *reinterpret_cast<const f32x4*>(park + rl * DR_PKS + c4);
                float sm = 0.f;
                if (row < prm.M) {
                    float* o = outn + (size_t)row * prm.N + col;
                    if (v4 && col < prm.N) {
                        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(o));
                        sm = (v[0] + v[1]) + (v[2] + v[3]);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (col + q < prm.N) { __builtin_nontemporal_store(v[q], o + q); sm += v[q]; }
                    }
                }
#pragma unroll
                for (int m = 8; m >= 1; m >>= 1) sm += __shfl_xor(sm, m, 64);
                rsum[k] += sm;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // the reads are done before the next pass overwrites the slab
        }
    }
    if (prm.rowsum && live) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int row = mi * TP + 32 * wq + 4 * k + (lane >> 4);
            if ((lane & 15) == 0 && row < prm.M) prm.rowsum[(size_t)n * prm.M + row] = rsum[k];
        }
    }
}

size_t dense_workspace_bytes(int B, int C, int M, int N)
{
    const size_t nbA = (M + TP - 1) / TP, nbB = (N + TP - 1) / TP, NCH = (C + KC - 1) / KC;
    return (size_t)B * (nbA + nbB) * NCH * DC_SIDE + (size_t)B * (nbA + nbB) * TP * sizeof(float) + 512;
}

hipError_t launch_dense_corr(const MapV& a, const MapV& b, int B, int C, int H1, int W1, int H2, int W2, int normalize,
                             float* out, void* ws, hipStream_t stream)
{
    DenseParams prm{};
    prm.a = a; prm.b = b; prm.out = out;
    prm.B = B; prm.C = C; prm.M = H1 * W1; prm.N = H2 * W2; prm.W1 = W1; prm.W2 = W2;
    prm.nbA = (prm.M + TP - 1) / TP; prm.nbB = (prm.N + TP - 1) / TP; prm.NCH = (C + KC - 1) / KC;
    prm.normalize = normalize;
    unsigned char* w = static_cast<unsigned char*>(ws);
    w += (256 - (reinterpret_cast<uintptr_t>(w) & 255)) & 255;
    prm.imgA = w;
    prm.imgB = w + (size_t)B * prm.nbA * prm.NCH * DC_SIDE;
    prm.rsA = reinterpret_cast<float*>(w + (size_t)B * (prm.nbA + prm.nbB) * prm.NCH * DC_SIDE);
    prm.rsB = prm.rsA + (size_t)B * prm.nbA * TP;
    auto cl = [&](const MapV& m) {
        return m.sc == 1 && C % 4 == 0 && C <= 1024 && (m.sn % 4) == 0 && (m.sh % 4) == 0 && (m.sw % 4) == 0 &&
               (reinterpret_cast<uintptr_t>(m.p) % 16) == 0;
    };
    const int vec = cl(a) && cl(b) ? 1 : 0;
    // round 6: the streaming kernel (dense_stream.hip) - B read as it lies and converted by the multiplying workgroups themselves, one small
    // statistics launch in front instead of the operand prep (debug bit 21: the row-block kernel below)
    if (vec && C % 8 == 0 && C > KC && prm.NCH <= DR_MAXCH && b.sh == W2 * b.sw && !(knob(KNOB_DEBUG) & (8192 | (1 << 21))))
        return launch_dense_stream(a, b, B, C, H1, W1, H2, W2, normalize, out, ws, stream);
    if (vec && C % 8 == 0 && prm.NCH <= DR_MAXCH && !(knob(KNOB_DEBUG) & 8192)) {          // (debug 8192: the tile kernel)
        // row-block kernel: the A map is consumed as it lies, only B is prepared (blocks [nbA, nbA + nbB) of every image)
        DenseParams pb = prm;
        pb.nbA = 0;                                    // the prep kernel's block index then runs over the B blocks only
        hipLaunchKernelGGL(dense_prep_kernel, dim3((unsigned)(B * prm.nbB)), dim3(NTHREADS), 0, stream, pb, vec);
        const int lds2 = DR_NST * DC_SIDE + 512;
        hipError_t e2 = ensure_dynamic_lds(reinterpret_cast<const void*>(&dense_rowblock_kernel<false>), lds2);
        if (e2 != hipSuccess) return e2;
        hipLaunchKernelGGL(dense_rowblock_kernel<false>, dim3((unsigned)(((B + 7) / 8) * 8 * prm.nbA)), dim3(NTHREADS), lds2, stream, prm);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(dense_prep_kernel, dim3((unsigned)(B * (prm.nbA + prm.nbB))), dim3(NTHREADS), 0, stream, prm, vec);
    const int lds = 2 * DC_SIDE;
    hipError_t ea = ensure_dynamic_lds(reinterpret_cast<const void*>(&dense_tile_kernel), lds);
    if (ea != hipSuccess) return ea;
    hipLaunchKernelGGL(dense_tile_kernel, dim3(prm.nbB, prm.nbA, B), dim3(NTHREADS), lds, stream, prm);
    return hipGetLastError();
}

// Both operands already prepared (stego_sample_panels): pair n = A image n % imagesA (M points) x B image n (N points).
hipError_t launch_dense_corr_panels_seg(const void* imgA, const float* rsA, int imagesA, const void* imgB, const float* rsB, int B, int C, int M, int N,
                                        float* out, float* out1, float* out2, int seg, float* rowsum, hipStream_t stream);
hipError_t launch_dense_corr_panels(const void* imgA, const float* rsA, int imagesA, const void* imgB, const float* rsB, int B, int C, int M, int N,
                                    float* out, float* rowsum, hipStream_t stream)
{
    return launch_dense_corr_panels_seg(imgA, rsA, imagesA, imgB, rsB, B, C, M, N, out, nullptr, nullptr, 0, rowsum, stream);
}

hipError_t launch_dense_corr_panels_seg(const void* imgA, const float* rsA, int imagesA, const void* imgB, const float* rsB, int B, int C, int M, int N,
                                        float* out, float* out1, float* out2, int seg, float* rowsum, hipStream_t stream)
{
    DenseParams prm{};
    prm.out = out;
    prm.B = B; prm.C = C; prm.M = M; prm.N = N; prm.W1 = 1; prm.W2 = 1;
    prm.nbA = (M + TP - 1) / TP; prm.nbB = (N + TP - 1) / TP; prm.NCH = (C + KC - 1) / KC;
    prm.imgA = const_cast<void*>(imgA); prm.imgB = const_cast<void*>(imgB);
    prm.rsA = const_cast<float*>(rsA); prm.rsB = const_cast<float*>(rsB);
    prm.a_mod = imagesA;
    prm.rowsum = rowsum;
    prm.out1 = out1; prm.out2 = out2; prm.seg = seg;
    prm.normalize = (knob(KNOB_DEBUG) >> 16) & 7;          // (tools: ablations of the row-block kernel)
    if (prm.NCH <= DR_MAXCH && !(knob(KNOB_DEBUG) & 8192)) {
        if (prm.nbA >= 2 && prm.NCH <= 2 && !(knob(KNOB_DEBUG) & (1 << 20))) {  // (debug bit 20: one row block per workgroup)
            const int ldsp = 2 * DC_SIDE + 1024 + (DR2_THREADS / 64) * 32 * DR_PKS * 4;
            hipError_t ep = ensure_dynamic_lds(reinterpret_cast<const void*>(&dense_rowpair_kernel<2>), ldsp);
            if (ep != hipSuccess) return ep;
            hipLaunchKernelGGL(dense_rowpair_kernel<2>, dim3((unsigned)(((B + 7) / 8) * 8 * ((prm.nbA + 1) / 2))), dim3(DR2_THREADS), ldsp, stream, prm);
            return hipGetLastError();
        }
        const dim3 grid((unsigned)(((B + 7) / 8) * 8 * prm.nbA));
        if (knob(KNOB_DEBUG) & (1 << 19)) {                     // (tools: four ring stages - three copies in flight -, one workgroup per CU)
            const int lds3 = 4 * DC_SIDE + 512;
            hipError_t e3 = ensure_dynamic_lds(reinterpret_cast<const void*>(&dense_rowblock_kernel<true, 4>), lds3);
            if (e3 != hipSuccess) return e3;
            hipLaunchKernelGGL((dense_rowblock_kernel<true, 4>), grid, dim3(NTHREADS), lds3, stream, prm);
            return hipGetLastError();
        }
        const int lds2 = 2 * DC_SIDE + 512;
        hipError_t e2 = ensure_dynamic_lds(reinterpret_cast<const void*>(&dense_rowblock_kernel<true, 2>), lds2);
        if (e2 != hipSuccess) return e2;
        hipLaunchKernelGGL((dense_rowblock_kernel<true, 2>), grid, dim3(NTHREADS), lds2, stream, prm);
        return hipGetLastError();
    }
    if (seg > 0) return hipErrorInvalidValue;                 // (segmented outputs: the row-block kernel only - codes, K <= 384)
    const int lds = 2 * DC_SIDE;
    hipError_t ea = ensure_dynamic_lds(reinterpret_cast<const void*>(&dense_tile_kernel), lds);
    if (ea != hipSuccess) return ea;
    hipLaunchKernelGGL(dense_tile_kernel, dim3(prm.nbB, prm.nbA, B), dim3(NTHREADS), lds, stream, prm);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && rowsum) e = launch_rowsum(out, rowsum, (long long)B * M, N, stream);        // (the tile kernel's workgroups see 128 columns)
    return e;
}

size_t dense_panel_image_bytes(int C, int P)
{
    return (size_t)((P + TP - 1) / TP) * ((C + KC - 1) / KC) * DC_SIDE;
}

}  // namespace stego
