// Fused forward of STEGO's ContrastiveCorrelationLoss for gfx950 (MI355X): ONE launch (+ a one-block scalar kernel).
//
// Reference path: src/modules.py:349-398 (forward), :325-347 (helper), :275-295 (norm / tensor_correlation / sample).
//
// Why one launch.  Measured on MI355X (tools/ubench/gather_bw.hip, profiles/r02_ubench_gather.txt): the 4-tap bilinear
// gather runs at ~62 GB/s per CU when the source image is shared through the XCD's L2 and at ~37 GB/s per CU when every
// image is pulled into ~7 different L2s (what the round-1 tile placement did to the negatives' source images) - the
// forward is bound by the ~6.5 TB/s the L2s can miss at, i.e. by FABRIC BYTES, not by its 113.7 MB of distinct tensors.
// The three-launch forward moved 239 MB through the fabric (feats read by the sampler AND by the tile kernel,
// 7.4 MB + 52 MB of anchor operands, a 27.5 MB finalize pass).  This kernel is organised around where bytes live:
//
//   * tile -> workgroup placement by the XCD of the tile's B-side SOURCE image (block b runs on XCD b % 8, observed):
//     all workgroups that gather from image j - the intra / inter tiles of anchor j and every negative tile (i, b) with
//     perm_i[b] = j - sit on XCD j % 8, so every feature / code map is fetched into exactly one L2, once;
//   * phase 1 (waves 0-3 of every workgroup, before their MFMA loop): the anchor sets are sampled + normalised ONCE,
//     in ~18-point units spread over all workgroups by the same XCD affinity (anchor b is sampled on XCD b % 8, so the
//     same fetch of image b serves its negatives), written as ready-made LDS operand images with write-through (sc1)
//     stores and published through one counter per anchor; a tile waits for its anchor's counter (bounded spin; on
//     timeout the waiting wave recomputes the anchor itself - same values, so the duplicate stores are benign) and then
//     streams the anchor operand with sc1 LDS-DMA copies.  The compact operand (196 KB) is what crosses XCDs, never the
//     raw taps of a second image (550 KB);
//   * the codes of a tile's own B set are sampled in-tile (they feed this tile only); the saved context of the
//     backward is written on the way;
//   * the batch-global old_mean (:331) - the reason the old forward had a third launch that re-read and re-wrote the
//     negative loss tensor - is a rendezvous INSIDE the launch: every negative tile publishes its sum(fd) as one
//     tagged 8-byte granule before it parks its tiles, and reads the B granules of its pair-set just before the
//     output sweep (measured 1.2 us mean / 1.7 us worst after the last publisher, hidden behind the parking).  The spin
//     is bounded: a tile that gives up writes the loss without the old_mean term and flags itself, and
//     corr_fused_scalars_kernel (which computes the three scalars from the per-tile sums in a fixed order anyway)
//     repairs flagged tiles.  Normally it repairs nothing.
//
// Arithmetic is that of corr_fwd.hip (same device functions): PREC_F16X3 split-fp16 feature products, fp32 code
// products, fp32 accumulation, fp32 epilogue.  No atomics on data, fixed summation orders: bitwise repeatable.
#include "corr_tile.h"
#include "host_util.h"

namespace stego {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int ANCHOR_CNT_STRIDE = 64;           // counters 256 bytes apart: pollers of different anchors hit different channels

__device__ __forceinline__ unsigned long long lanes_below(int lane) { return lane ? (~0ull >> (64 - lane)) : 0ull; }

// ------------------------------------------------------------------------------------------ tile placement
// XCD preference of tile t = (p, b): the image its B side is gathered from, modulo 8.
__device__ __forceinline__ int tile_pref(const FusedParams& prm, int t, int n_tiles)
{
    if (t >= n_tiles) return -1;
    const int B = prm.B;
    const int p = t / B, b = t - p * B;
    int src = b;
    if (p >= 2) src = (int)prm.perms[(size_t)(p - 2) * B + b];
    return src & 7;
}

// Bijection workgroups <-> tiles, computed redundantly by every wave (perms is device data: the host cannot build it
// without a sync).  Workgroup w sits on XCD w % 8 (slot w / 8).  The r-th tile (in tile order) that prefers XCD x takes
// slot r of x while x has slots; tiles beyond that ("overflow") fill the slots other XCDs leave free, in order.
// Tiles are visited in blocks of 4 x 64 whose perms loads are issued together (one round trip for <= 256 tiles).
constexpr int ASSIGN_NB = 4;

// the loads of the first block, issued at the top of the kernel so that they fly together with phase 1's
__device__ __forceinline__ void assign_prefetch(const FusedParams& prm, int lane, int (&pref)[ASSIGN_NB])
{
    const int n_tiles = prm.n_sets * prm.B;
#pragma unroll
    for (int i = 0; i < ASSIGN_NB; ++i) pref[i] = tile_pref(prm, 64 * i + lane, n_tiles);
}

__device__ int assign_tile(const FusedParams& prm, int me, int lane, const int (&pref0)[ASSIGN_NB])
{
    const int n_tiles = prm.n_sets * prm.B;
    constexpr int NB = ASSIGN_NB;
    int cnt[8];
#pragma unroll
    for (int x = 0; x < 8; ++x) cnt[x] = 0;
    int pref[NB];
    for (int c0 = 0; c0 < n_tiles; c0 += 64 * NB) {
#pragma unroll
        for (int i = 0; i < NB; ++i) pref[i] = c0 == 0 ? pref0[i] : tile_pref(prm, c0 + 64 * i + lane, n_tiles);
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int x = 0; x < 8; ++x) cnt[x] += __builtin_popcountll(__ballot(pref[i] == x));
    }
    const int xme = me & 7, rme = me >> 3;
    int cnt_me = 0, k = 0;
#pragma unroll
    for (int x = 0; x < 8; ++x) {
        const int nslot = (n_tiles - x + 7) >> 3;
        if (x == xme) cnt_me = cnt[x];
        if (x < xme) k += max(0, nslot - cnt[x]);
    }
    const bool direct = rme < cnt_me;
    k += rme - cnt_me;                                   // index among the free slots (used when !direct)
    int base[8];
#pragma unroll
    for (int x = 0; x < 8; ++x) base[x] = 0;
    int ovf_base = 0, result = 0;
    const unsigned long long below = lanes_below(lane);
    for (int c0 = 0; c0 < n_tiles; c0 += 64 * NB) {
#pragma unroll
        for (int i = 0; i < NB; ++i) pref[i] = c0 == 0 ? pref0[i] : tile_pref(prm, c0 + 64 * i + lane, n_tiles);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            int myrank = 0, myslots = 0;
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const unsigned long long m = __ballot(pref[i] == x);
                if (pref[i] == x) { myrank = base[x] + __builtin_popcountll(m & below); myslots = (n_tiles - x + 7) >> 3; }
                base[x] += __builtin_popcountll(m);
            }
            const bool ovf = pref[i] >= 0 && myrank >= myslots;
            const unsigned long long om = __ballot(ovf);
            const int ovf_idx = ovf_base + __builtin_popcountll(om & below);
            ovf_base += __builtin_popcountll(om);
            const bool hit = direct ? (pref[i] == xme && myrank == rme) : (ovf && ovf_idx == k);
            const unsigned long long hm = __ballot(hit);
            if (hm) result = c0 + 64 * i + __builtin_ctzll(hm);
        }
    }
    return __builtin_amdgcn_readfirstlane(result);
}

// ------------------------------------------------------------------------------------------ sampling (phase 1, own codes)
// Lane hl of a half-wave owns, of ONE sample point: feature channels 128 j + 4 hl .. + 3 (j < NJ) and code channels
// 2 hl, 2 hl + 1 and 64 + hl.
struct CodeTaps {
    f32x2 a[4];
    float b[4];
};

__device__ __forceinline__ void point_taps(const FusedParams& prm, const f32x2 cxy, int q, int4& yx, float4& w)
{
    yx = make_int4(0, 0, 0, 0);
    w = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < prm.P) make_taps(cxy[0], cxy[1], prm.H, prm.W, yx, w);
}

// index of point q = (h, w) in a coords image: it samples coords[w][h] (sample() permutes the grid, modules.py:288)
__device__ __forceinline__ int coord_index(const FusedParams& prm, int q)
{
    const int qq = q < prm.P ? q : 0;
    const int hh = qq / prm.S, ww = qq - hh * prm.S;
    return (ww * prm.S + hh) * 2;
}

__device__ __forceinline__ void code_issue(const FusedParams& prm, const float* cimg, const int4 oc, int hl, CodeTaps& ct)
{
    const int c1 = 64 + hl < prm.K ? 64 + hl : 0;
    const int c0 = 2 * hl < prm.K ? 2 * hl : 0;
    ct.a[0] = *reinterpret_cast<const f32x2*>(cimg + oc.x + c0);
    ct.a[1] = *reinterpret_cast<const f32x2*>(cimg + oc.y + c0);
    ct.a[2] = *reinterpret_cast<const f32x2*>(cimg + oc.z + c0);
    ct.a[3] = *reinterpret_cast<const f32x2*>(cimg + oc.w + c0);
    if (prm.K > 64) { ct.b[0] = cimg[oc.x + c1]; ct.b[1] = cimg[oc.y + c1]; ct.b[2] = cimg[oc.z + c1]; ct.b[3] = cimg[oc.w + c1]; }
    else { ct.b[0] = ct.b[1] = ct.b[2] = ct.b[3] = 0.f; }
}

// blend + L2-normalise (norm(), modules.py:275-276) the code vector of one point; returns the raw norm
__device__ __forceinline__ float code_finish(const FusedParams& prm, const CodeTaps& ct, const float4 wg, bool valid, int hl,
                                             f32x2& r0, float& r1)
{
    r0 = wg.x * ct.a[0] + wg.y * ct.a[1] + wg.z * ct.a[2] + wg.w * ct.a[3];
    r1 = wg.x * ct.b[0] + wg.y * ct.b[1] + wg.z * ct.b[2] + wg.w * ct.b[3];
    if (2 * hl >= prm.K) r0 = f32x2{0.f, 0.f};
    if (64 + hl >= prm.K) r1 = 0.f;
    float ss = r0[0] * r0[0] + r0[1] * r0[1] + r1 * r1;
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) ss += __shfl_xor(ss, m, 64);
    const float nr = valid ? sqrtf(ss) : 0.f;
    const float inv = valid ? __builtin_amdgcn_rcpf(fmaxf(nr, 1e-10f)) : 0.f;
    r0 = r0 * inv;
    r1 = r1 * inv;
    return nr;
}

// Phase 1: points [beg, end) of the flat list "anchors xa, xa + 8, ... , 128 points each" sampled by `nw` waves (this
// one is number `wv`), 2 G points per wave and pass.  Features and codes are normalised and laid out in a wave-private
// LDS area exactly as they sit in the operand images, then leave with coalesced 16-byte write-through (sc1) stores -
// scattered sc1 stores straight from the registers were measured at 0.4 TB/s chip-wide (22 us for 8.6 MB) - and each
// finished point is published on its anchor's counter.
template <int NJ, int PREC, int G>
__device__ __forceinline__ void sample_anchor_points(const FusedParams& prm, int xa, int beg, int end, int wv, int nw, int lane,
                                                     unsigned char* lds, __amdgpu_buffer_rsrc_t fs_rsrc,
                                                     __amdgpu_buffer_rsrc_t cs_rsrc, unsigned long long* tsd)
{
    constexpr int PW = 2 * G;                                         // points per wave and pass
    constexpr int NPL = 2 * NJ * (PREC == PREC_F16X3 ? 2 : 1);       // planes of an anchor's operand image
    constexpr int RB = PREC == PREC_F16X3 ? LDH * 2 : LDA * 4;       // bytes per row of a plane
    constexpr int UPR = RB / 16;                                      // 16-byte units per row
    const int hl = lane & 31, hw = lane >> 5;
    const int crow = prm.LDK * 4;                                     // bytes per code row (a multiple of 16)
    unsigned char* lds_code = lds + NPL * PW * RB;
    const MapV mf = prm.feats, mc = prm.code;
    for (int i0 = beg; i0 < end; i0 += nw * PW) {
        const int first = i0 + PW * wv;
        int4 yx[G];
        float4 w[G];
        int ba[G], q[G];
        bool act[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int idx = first + 2 * g + hw;
            act[g] = idx < end;
            ba[g] = min(xa + 8 * (idx >> 7), prm.B - 1);
            q[g] = idx & (TP - 1);
            const f32x2 cxy = *reinterpret_cast<const f32x2*>(prm.coords1 + (size_t)ba[g] * prm.P * 2 + coord_index(prm, q[g]));
            point_taps(prm, cxy, act[g] ? q[g] : TP, yx[g], w[g]);
            if (act[g] && hl == 0) {                                  // tap table of the saved context (backward only)
                prm.tapyx[(size_t)ba[g] * TP + q[g]] = yx[g];
                prm.tapw[(size_t)ba[g] * TP + q[g]] = w[g];
            }
        }
        if (tsd) tsd[0] = __builtin_amdgcn_s_memrealtime();
        f32x4 t[G][NJ][4];
        CodeTaps ct[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int4 of = taps_to_offsets(yx[g], mf.sh, mf.sw);
            const char* fb = reinterpret_cast<const char*>(mf.p + (long long)ba[g] * mf.sn);
            const f32x4* p0 = reinterpret_cast<const f32x4*>(fb + (unsigned)((of.x + 4 * hl) * 4));
            const f32x4* p1 = reinterpret_cast<const f32x4*>(fb + (unsigned)((of.y + 4 * hl) * 4));
            const f32x4* p2 = reinterpret_cast<const f32x4*>(fb + (unsigned)((of.z + 4 * hl) * 4));
            const f32x4* p3 = reinterpret_cast<const f32x4*>(fb + (unsigned)((of.w + 4 * hl) * 4));
#pragma unroll
            for (int j = 0; j < NJ; ++j) {       // inactive points re-read pixel (0,0) of a valid image: harmless
                t[g][j][0] = p0[32 * j]; t[g][j][1] = p1[32 * j]; t[g][j][2] = p2[32 * j]; t[g][j][3] = p3[32 * j];
            }
            code_issue(prm, mc.p + (long long)ba[g] * mc.sn, taps_to_offsets(yx[g], mc.sh, mc.sw), hl, ct[g]);
        }
        if (tsd) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            tsd[1] = __builtin_amdgcn_s_memrealtime();
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int r = 2 * g + hw;                                 // row inside the wave's staging area
            const bool valid = act[g] && q[g] < prm.P;
            const float4 wg = w[g];
            f32x4 v[NJ];
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = wg.x * t[g][j][0][e] + wg.y * t[g][j][1][e] + wg.z * t[g][j][2][e] + wg.w * t[g][j][3][e];
                    v[j][e] = x;
                    ss += x * x;
                }
#pragma unroll
            for (int m = 16; m >= 1; m >>= 1) ss += __shfl_xor(ss, m, 64);
            // F.normalize eps (modules.py:276); padding points (zero taps) are written as zeros
            const float inv = valid ? __builtin_amdgcn_rcpf(fmaxf(sqrtf(ss), 1e-10f)) : 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const f32x4 vn = v[j] * inv;
                if constexpr (PREC == PREC_F32) {
                    const int c = 128 * j + 4 * hl;
                    *reinterpret_cast<f32x4*>(lds + ((c >> 6) * PW + r) * RB + (c & 63) * 4) = vn;
                } else {
                    unsigned h0, l0, h1, l1;
                    split_f16_pair(vn[0], vn[1], h0, l0);
                    split_f16_pair(vn[2], vn[3], h1, l1);
                    const bool odd = hl & 1;                          // even lanes collect the hi halves of a lane pair, odd the lo
                    const unsigned r0 = __shfl_xor(odd ? h0 : l0, 1, 64);
                    const unsigned r1 = __shfl_xor(odd ? h1 : l1, 1, 64);
                    const u32x4 d = odd ? u32x4{r0, r1, l0, l1} : u32x4{h0, h1, r0, r1};
                    const int c = 128 * j + 4 * (hl & ~1);
                    *reinterpret_cast<u32x4*>(lds + (((c >> 6) * 2 + (odd ? 1 : 0)) * PW + r) * RB + (c & 63) * 2) = d;
                }
            }
            f32x2 c0;
            float c1;
            const float nr = code_finish(prm, ct[g], wg, valid, hl, c0, c1);
            if (2 * hl < prm.KQ) *reinterpret_cast<f32x2*>(lds_code + r * crow + 8 * hl) = c0;
            if (64 + hl < prm.KQ) *reinterpret_cast<float*>(lds_code + r * crow + 4 * (64 + hl)) = c1;
            if (act[g] && hl == 0) prm.nrm[(size_t)ba[g] * TP + q[g]] = nr;
        }
        // ---- out: every plane of the wave's rows is one contiguous run in the operand image (rows of one anchor)
        const int nact = min(max(end - first, 0), PW);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
            for (int u0 = 0; u0 < PW * UPR; u0 += 64) {
                const int u = u0 + lane;
                const int row = u / UPR;
                if (u < PW * UPR && row < nact) {
                    const int idx = first + row;
                    const int set = xa + 8 * (idx >> 7), qq = idx & (TP - 1);
                    const u32x4 d = *reinterpret_cast<const u32x4*>(lds + pl * PW * RB + u * 16);
                    const unsigned off = (unsigned)((((size_t)set * NPL + pl) * TP + qq) * RB + (u - row * UPR) * 16);
                    __builtin_amdgcn_raw_buffer_store_b128(d, fs_rsrc, off, 0, 16);
                }
            }
        }
        {
            const int upr = crow / 16;
            for (int u = lane; u < PW * upr; u += 64) {
                const int row = u / upr;
                if (row < nact) {
                    const int idx = first + row;
                    const int set = xa + 8 * (idx >> 7), qq = idx & (TP - 1);
                    const u32x4 d = *reinterpret_cast<const u32x4*>(lds_code + u * 16);
                    const unsigned off = (unsigned)(((size_t)set * TP + qq) * crow + (u - row * upr) * 16);
                    __builtin_amdgcn_raw_buffer_store_b128(d, cs_rsrc, off, 0, 16);
                }
            }
        }
        if (tsd) tsd[2] = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this wave's write-through stores have landed
        // publish: one add per anchor this wave contributed to (its rows span at most two)
        if (nact > 0 && lane < 2) {
            const int a0 = first >> 7, a1 = (first + nact - 1) >> 7;
            const int n0 = min(nact, ((a0 + 1) << 7) - first);
            const int n = lane == 0 ? n0 : nact - n0;
            if (n > 0 && (lane == 0 || a1 != a0))
                __hip_atomic_fetch_add(prm.anchor_cnt + (size_t)(xa + 8 * (lane == 0 ? a0 : a1)) * ANCHOR_CNT_STRIDE, (unsigned)n,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// bytes of the wave-private LDS staging area of sample_anchor_points
template <int NJ, int PREC, int G>
constexpr int anchor_stage_bytes(int LDK_max)
{
    return 2 * NJ * (PREC == PREC_F16X3 ? 2 : 1) * 2 * G * (PREC == PREC_F16X3 ? LDH * 2 : LDA * 4) + 2 * G * LDK_max * 4;
}

// ------------------------------------------------------------------------------------------ the kernel
template <int PREC, int NJ>
__global__ void __launch_bounds__(TILE_THREADS) corr_fused_kernel(const FusedParams prm, const int stage_bytes)
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
    constexpr int V = 4;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool mfma_team = wave8 < 4;            // wave-uniform role
    const int wave = wave8 & 3;
    const int gt = tid & (NTHREADS - 1);         // index inside the 256-thread team
    const int wr = wave >> 1, wc = wave & 1;
    const int B = prm.B, P = prm.P, NCH = prm.NCH;
    const int me = blockIdx.x;
    const int n_tiles = prm.n_sets * B;
    const int cside = TP * prm.LDK * 4;          // bytes of one code operand (a multiple of 1 KiB)

    unsigned long long* ts = reinterpret_cast<unsigned long long*>(prm.stats + (size_t)n_tiles * 4 + 256) + (size_t)me * 16;
    const bool stamp_on = (prm.debug & 256) && tid == 0;
    if (stamp_on) ts[0] = __builtin_amdgcn_s_memrealtime();

    const __amdgpu_buffer_rsrc_t fs_rsrc = __builtin_amdgcn_make_buffer_rsrc(prm.fs, 0, prm.fs_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t cs_rsrc = __builtin_amdgcn_make_buffer_rsrc(prm.cs, 0, prm.cs_bytes, 0x00020000);

    int pref0[ASSIGN_NB];
    assign_prefetch(prm, lane, pref0);           // (perms loads in flight under phase 1)

    // ---- phase 1 (all 8 waves): my share of the anchor sets of my XCD.  Wave-private staging areas live in stage
    // buffer 1, which nothing else touches before the first workgroup barrier B1(0).
    constexpr int G1 = NJ <= 3 ? 2 : 1;
    constexpr int P1_BYTES = anchor_stage_bytes<NJ, PREC, G1>(80);
    static_assert(8 * P1_BYTES <= 2 * FEAT_SIDE_F32, "phase-1 staging must fit stage buffer 1");
    unsigned char* p1_lds = stage + stage_bytes + wave8 * P1_BYTES;
    if (me < prm.n_owner && !(prm.debug & 64)) {
        const int x = me & 7, r = me >> 3;
        const int nb = x < B ? (B - x + 7) >> 3 : 0;
        const int nslot = (prm.n_owner - x + 7) >> 3;
        const long long L = (long long)nb * TP;
        const int beg = (int)(L * r / nslot), end = (int)(L * (r + 1) / nslot);
        sample_anchor_points<NJ, PREC, G1>(prm, x, beg, end, wave8, 8, lane, p1_lds, fs_rsrc, cs_rsrc, stamp_on ? ts + 8 : nullptr);
    }
    if (stamp_on) ts[1] = __builtin_amdgcn_s_memrealtime();

    // ---- which tile am I (every wave for itself: no barrier before the teams part ways)
    const int tile = assign_tile(prm, me, lane, pref0);
    if (stamp_on) ts[6] = __builtin_amdgcn_s_memrealtime();
    const int b = tile % B, p = tile / B;
    const bool sameAB = p == 0;
    const int sA = b;
    const int sB = p == 0 ? b : p * B + b;
    const bool usePos = p == 1;
    int src = b;
    if (p >= 2) src = (int)prm.perms[(size_t)(p - 2) * B + b];
    src = __builtin_amdgcn_readfirstlane(src);
    const unsigned char* fsA = prm.fs + (size_t)sA * NCH * FSIDE;
    const unsigned char* csA = reinterpret_cast<const unsigned char*>(prm.cs) + (size_t)sA * cside;
    const unsigned char* csB = reinterpret_cast<const unsigned char*>(prm.cs) + (size_t)sB * cside;
    const MapV mfB = usePos ? prm.feats_pos : prm.feats;
    const MapV mcB = usePos ? prm.code_pos : prm.code;
    const float* imgB = mfB.p + (long long)src * mfB.sn;
    const float* coordsB = prm.coords2 + (size_t)b * P * 2;
    const bool gatherB = !sameAB;

    auto copies = [&](int t) {                    // async LDS copies of stage t: A features / both code operands
        unsigned char* dst = stage + (t & 1) * stage_bytes;
        if (t < NCH) {
            if (prm.debug & 2048) issue_copy<0>(fsA + (size_t)t * FSIDE, dst, FSIDE / 1024, wave, lane);     // (timing experiment only)
            else issue_copy<16>(fsA + (size_t)t * FSIDE, dst, FSIDE / 1024, wave, lane);
        } else {
            issue_copy<16>(csA, dst, cside / 1024, wave, lane);
            if (!sameAB) issue_copy<16>(csB, dst + cside, cside / 1024, wave, lane);
        }
    };

    f32x16 accf[2][2], accc[2][2];
    if (mfma_team) {
        // ================================================================= MFMA team
        // Codes of my own B set (they feed this tile only; the backward's saved context on the way): 32 points per chunk
        // iteration, gathers issued before the chunk's MFMAs, blended after them - off the start-up critical path.
        const int hl = lane & 31, hw = lane >> 5;
        constexpr int G2 = 4;                     // passes (of 8 points) per iteration: 4 iterations cover the 128 points
        f32x2 own_xy[4 * G2];
        const float* cimgB = mcB.p + (long long)src * mcB.sn;
        if (!sameAB) {
#pragma unroll
            for (int i = 0; i < 4 * G2; ++i)
                own_xy[i] = *reinterpret_cast<const f32x2*>(coordsB + coord_index(prm, 8 * i + 2 * wave + hw));
        }
        if (stamp_on) ts[7] = __builtin_amdgcn_s_memrealtime();
        // Wait for my anchor set (published by its phase-1 owners).  ONE wave per workgroup polls (a relaxed load, a long
        // sleep): 900 waves polling 32 words every 0.25 us collapsed the channels holding them and delayed the very
        // stores and counter updates they were waiting for (phase 1 took 17-33 us instead of 9).  Wave 0 then issues the
        // whole first copy; the other waves only need the anchor after the barrier wave 0 arrives at.
        if (wave == 0) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            bool ready = false;
            for (;;) {
                const unsigned c = __hip_atomic_load(prm.anchor_cnt + (size_t)sA * ANCHOR_CNT_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__builtin_amdgcn_readfirstlane(c) >= (unsigned)TP) { ready = true; break; }
                if ((long long)(__builtin_amdgcn_s_memrealtime() - t0) > prm.timeout_ticks) break;
                __builtin_amdgcn_s_sleep(32);
            }
            if (!ready) {
                // The owners did not show up in time (not co-resident: a shared or over-subscribed device).  Sample the
                // whole anchor here, this wave alone: identical inputs give identical bytes, so racing with a late owner
                // is benign; nothing else in the launch depends on the counter being exact.
                sample_anchor_points<NJ, PREC, G1>(prm, sA, 0, TP, 0, 1, lane, p1_lds, fs_rsrc, cs_rsrc, nullptr);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        if (stamp_on) ts[2] = __builtin_amdgcn_s_memrealtime();
        zero_acc(accf);
        if (wave == 0) {                         // all pieces of stage 0 (the other waves have not polled)
            for (int pc = 0; pc < FSIDE / 1024; ++pc)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(fsA + (size_t)pc * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void*)(stage + pc * 1024), 16, 0, 16);
        }
        CodeTaps oc[G2];
        float4 ow[G2];
        for (int t = 0; t < NCH; ++t) {
            sync_after_lds_dma();                // B1(t): stage t complete (copies landed; gathered B: written before)
            copies(t + 1);                       // stage NCH = the code operands
            // own codes, group t-1: gathered during the previous iteration, blended + stored now (the stores drain at
            // B1(t+1); the last group's at B1(5) <= B1(NCH-1), before the code operands are copied back)
            if (!sameAB && t >= 1 && t <= 4 && !(prm.debug & 1024)) {
#pragma unroll
                for (int i = 0; i < G2; ++i) {
                    const int q = 8 * (G2 * (t - 1) + i) + 2 * wave + hw;
                    f32x2 c0;
                    float c1;
                    const float nr = code_finish(prm, oc[i], ow[i], q < P, hl, c0, c1);
                    float* crow = prm.cs + ((size_t)sB * TP + q) * prm.LDK;
                    if (2 * hl < prm.KQ) *reinterpret_cast<f32x2*>(crow + 2 * hl) = c0;
                    if (64 + hl < prm.KQ) crow[64 + hl] = c1;
                    if (hl == 0) prm.nrm[(size_t)sB * TP + q] = nr;
                }
            }
            if (!sameAB && t < 4 && !(prm.debug & 1024)) {
#pragma unroll
                for (int i = 0; i < G2; ++i) {
                    const int q = 8 * (G2 * t + i) + 2 * wave + hw;
                    int4 yx;
                    f32x2 xy = own_xy[0];
#pragma unroll
                    for (int k = 1; k < 4 * G2; ++k) xy = (k == G2 * t + i) ? own_xy[k] : xy;      // static register indexing
                    point_taps(prm, xy, q, yx, ow[i]);
                    if (hl == 0) {
                        prm.tapyx[(size_t)sB * TP + q] = yx;
                        prm.tapw[(size_t)sB * TP + q] = ow[i];
                    }
                    code_issue(prm, cimgB, taps_to_offsets(yx, mcB.sh, mcB.sw), hl, oc[i]);
                }
            }
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
        // tap table of the B points: every wave computes the 32 entries it will read itself (no early barrier)
        if (gatherB && lane < 32) {
            const int q = 16 * (lane >> 2) + 4 * wave + (lane & 3);
            int4 yx = make_int4(0, 0, 0, 0);
            float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < P) {
                const int hh = q / prm.S, ww = q - hh * prm.S;
                const f32x2 cxy = *reinterpret_cast<const f32x2*>(coordsB + (ww * prm.S + hh) * 2);
                make_taps(cxy[0], cxy[1], prm.H, prm.W, yx, w);
            }
            tapo[q] = taps_to_offsets(yx, mfB.sh, mfB.sw);
            tapw[q] = w;
        }
        GatherRegs<V> g;
        constexpr int ITEMS = GatherRegs<V>::ITEMS;
        float ss[ITEMS], bsc[ITEMS];
        const int gslot = gt % GatherRegs<V>::SLOTS, gprow = gt / GatherRegs<V>::SLOTS;
        const int lane_off = gslot * V;                                  // channels-last: this lane's 16 bytes of a chunk
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            ss[j] = 0.f;
            bsc[j] = 0.f;
        }
        auto chunk_ptr = [&](int t) { return imgB + (long long)min(t * KC, prm.C - KC); };
        if (gatherB) {
            gather_issue<V>(g, chunk_ptr(0), tapo, lane_off, gprow, 0, ITEMS);
            gather_commit<V, PREC>(g, tapw, true, stage + FSIDE, ss, bsc, gslot, gprow, 0, ITEMS);
            gather_issue<V>(g, chunk_ptr(1), tapo, lane_off, gprow, 0, ITEMS);
        }
        for (int t = 0; t < NCH; ++t) {
            __syncthreads();                     // B1(t): the MFMA team is done with stage t-1 = the buffer written next
            if (gatherB && t + 1 < NCH) {
                const float* nxt = chunk_ptr(t + 2);
                void* dstB = stage + ((t + 1) & 1) * stage_bytes + FSIDE;
#pragma unroll
                for (int j = 0; j < ITEMS; ++j) {
                    gather_commit<V, PREC>(g, tapw, true, dstB, ss, bsc, gslot, gprow, j, 1);
                    gather_issue<V>(g, nxt, tapo, lane_off, gprow, j, 1);
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
    const bool rendezvous = loss_out != nullptr && prm.pointwise && !(prm.debug & 32);
    __syncthreads();                             // E0: stage buffers are dead, csc complete
    if (stamp_on) ts[3] = __builtin_amdgcn_s_memrealtime();
    if (mfma_team) {
        // sum(fd) of the tile straight from the accumulators (fixed order), then park fd in the flat output layout
        float s = 0.f;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = 64 * wc + 32 * ni + (lane & 31);
            float sc = 0.f;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 64 * wr + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    sc += row < P ? accf[mi][ni][r] : 0.f;
                }
            s += col < P ? sc * csc[col] : 0.f;
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
        if (lane == 0) red[wave] = s;
        park_flat(accf, Tfd + a, P, csc, lane, wr, wc);
    }
    __syncthreads();                             // E1: Tfd and the four partial sums are complete
    if (tid == 0) {
        const float sfd = (red[0] + red[1]) + (red[2] + red[3]);
        prm.stats[(size_t)tile * 4 + 0] = sfd;
        if (rendezvous)                          // one aligned 8-byte write-through store: {tag, value}
            __hip_atomic_store(prm.gran + tile, (1ull << 32) | __builtin_bit_cast(unsigned, sfd), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
    float* omv = red + 8;                        // [0] old_mean, [1] applied
    if (mfma_team) {
        park_flat(accc, Tcd + a, P, nullptr, lane, wr, wc);
        if (wave == 0) {
            // old_mean of my pair-set = sum of its B tile sums / (B P^2), summed in image order (as the scalar kernel does)
            float om = 0.f, applied = 1.f;
            if (rendezvous) {
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                float acc = 0.f;
                bool ok_all = true;
                for (int i0 = 0; i0 < B && ok_all; i0 += 64) {
                    const int i = i0 + lane;
                    unsigned long long x = 0;
                    for (;;) {
                        bool ok = true;
                        if (i < B) {
                            x = __hip_atomic_load(prm.gran + (size_t)p * B + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            ok = (x >> 32) == 1ull;
                        }
                        if (__all(ok)) break;
                        if ((long long)(__builtin_amdgcn_s_memrealtime() - t0) > prm.timeout_ticks) { ok_all = false; break; }
                        __builtin_amdgcn_s_sleep(16);
                    }
                    const float v = i < B ? __builtin_bit_cast(float, (unsigned)x) : 0.f;
                    const int n = min(64, B - i0);
                    for (int l = 0; l < n; ++l) acc += __shfl(v, l, 64);
                }
                if (ok_all) om = acc * (1.f / ((float)B * (float)P2));     // same expression as the scalar kernel
                else applied = 0.f;
            } else if (loss_out != nullptr && prm.pointwise) {
                applied = 0.f;                   // debug & 32: leave it to the scalar kernel
            }
            if (lane == 0) { omv[0] = om; omv[1] = applied; }
        }
    } else {
        // row means of fd (fd.mean([3,4]), modules.py:332) by two lanes per row
        const int row = gt >> 1, half = gt & 1;
        const int hlf = (P + 1) >> 1;
        float s0 = 0.f, s1 = 0.f;
        if (row < P) {
            const float* srcr = Tfd + a + row * P;
            const int c1 = half ? P : hlf;
            int c = half ? hlf : 0;
            for (; c + 2 <= c1; c += 2) { s0 += srcr[c]; s1 += srcr[c + 1]; }
            if (c < c1) s0 += srcr[c];
        }
        float sfull = s0 + s1;
        sfull += __shfl_xor(sfull, 1, 64);
        if (half == 0 && row < TP) rowmean[row] = prm.pointwise ? sfull / (float)P : 0.f;
    }
    __syncthreads();                             // E2: Tcd, rowmean, old_mean
    if (stamp_on) ts[4] = __builtin_amdgcn_s_memrealtime();

    const float om = omv[0];
    const float cmin = prm.cmin, cmax = prm.cmax;
    const float invP = 1.f / (float)P;
    float loss_part = 0.f, clamp_part = 0.f;
    {
        const int nvec = (a + P2 + 3) >> 2;
        for (int v = tid; v < nvec; v += TILE_THREADS) {
            const int f0 = 4 * v;
            const f32x4 fd4 = *reinterpret_cast<const f32x4*>(Tfd + f0);
            const f32x4 cd4 = *reinterpret_cast<const f32x4*>(Tcd + f0);
            f32x4 w4, lo4;
            bool ok[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = f0 + k - a;
                ok[k] = e >= 0 && e < P2;
                const int r = min(max((int)(((float)e + 0.5f) * invP), 0), P - 1);
                const float w = fd4[k] - (rowmean[r] + shift);                 // fd_centred - shift
                const float cl = fminf(fmaxf(cd4[k], cmin), cmax);
                const float lp = -cl * w;                                      // loss without the old_mean term
                const unsigned pass = (cd4[k] >= cmin && cd4[k] <= cmax) ? 1u : 0u;
                w4[k] = __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, w) & ~1u) | pass);
                lo4[k] = __builtin_fmaf(-om, cl, lp);                          // loss = -clamp(cd) * (fd_centred + old_mean - shift)
                if (ok[k]) { loss_part += lp; clamp_part += cl; }
            }
            const int e0 = f0 - a;
            if (vec_ok && ok[0] && ok[3]) {
                *reinterpret_cast<f32x4*>(cd_out + e0) = cd4;
                if (loss_out) *reinterpret_cast<f32x4*>(loss_out + e0) = lo4;
                if (w_out) *reinterpret_cast<f32x4*>(w_out + e0) = w4;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (ok[k]) {
                        cd_out[e0 + k] = cd4[k];
                        if (loss_out) loss_out[e0 + k] = lo4[k];
                        if (w_out) w_out[e0 + k] = w4[k];
                    }
            }
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        loss_part += __shfl_xor(loss_part, m, 64);
        clamp_part += __shfl_xor(clamp_part, m, 64);
    }
    if (lane == 0) { red[16 + wave8 * 2] = loss_part; red[16 + wave8 * 2 + 1] = clamp_part; }
    __syncthreads();
    if (tid == 0) {
        float s1 = 0.f, s2 = 0.f;
        for (int w = 0; w < 8; ++w) { s1 += red[16 + w * 2]; s2 += red[16 + w * 2 + 1]; }
        float* st = prm.stats + (size_t)tile * 4;
        st[1] = s1; st[2] = s2; st[3] = omv[1];
    }
    if (stamp_on) ts[5] = __builtin_amdgcn_s_memrealtime();
}

// The three scalars and the saved means, from the per-tile sums in image order (modules.py:331,393,395):
//   old_mean_p = sum_b sum(fd) / (B P^2);   mean(loss_p) = (sum lp - old_mean_p * sum clamp) / (B P^2).
// Blocks 1.. : one per negative tile; a tile whose rendezvous gave up (stats[3] == 0) is repaired here:
//   loss = lp - old_mean * clamp(cd)   (the same fma the tile kernel uses).  Normally every block but 0 returns at once.
__global__ void __launch_bounds__(NTHREADS) corr_fused_scalars_kernel(const FusedParams prm)
{
    const int B = prm.B, P2 = prm.P * prm.P;
    const float inv_cnt = 1.f / ((float)B * (float)P2);
    if (blockIdx.x == 0) {
        if ((int)threadIdx.x < prm.n_sets) {
            const int p = threadIdx.x;
            float fs = 0.f, ls = 0.f, cs = 0.f;
            for (int b = 0; b < B; ++b) {
                const float* st = prm.stats + ((size_t)p * B + b) * 4;
                fs += st[0]; ls += st[1]; cs += st[2];
            }
            const float om = prm.pointwise ? fs * inv_cnt : 0.f;
            if (prm.saved_mean) prm.saved_mean[p] = om;
            if (p < 2) prm.loss_means[p] = (ls - om * cs) * inv_cnt;
        }
        return;
    }
    const int tile = 2 * B + (int)blockIdx.x - 1;            // negative tiles only carry a loss tensor
    if (!prm.pointwise || prm.stats[(size_t)tile * 4 + 3] != 0.f) return;
    const int p = tile / B;
    float fs = 0.f;
    for (int b = 0; b < B; ++b) fs += prm.stats[((size_t)p * B + b) * 4];
    const float om = fs * inv_cnt;
    const float cmin = prm.cmin, cmax = prm.cmax;
    float* loss = prm.neg_loss + (size_t)(tile - 2 * B) * P2;
    const float* cd = prm.neg_cd + (size_t)(tile - 2 * B) * P2;
    for (int e = threadIdx.x; e < P2; e += NTHREADS) {
        const float cl = fminf(fmaxf(cd[e], cmin), cmax);
        loss[e] = __builtin_fmaf(-om, cl, loss[e]);
    }
}

// ------------------------------------------------------------------------------------------------------ launch
bool fused_supported(const FusedParams& prm, int precision)
{
    auto cl4 = [&](const MapV& m) {
        return m.sc == 1 && (m.sn % 4) == 0 && (m.sh % 4) == 0 && (m.sw % 4) == 0 &&
               (reinterpret_cast<uintptr_t>(m.p) % 16) == 0 &&
               ((long long)(prm.H - 1) * m.sh + (long long)(prm.W - 1) * m.sw + prm.C) * 4 < (1ll << 31);
    };
    auto cl2 = [&](const MapV& m) {
        return m.sc == 1 && (m.sn % 2) == 0 && (m.sh % 2) == 0 && (m.sw % 2) == 0 && (reinterpret_cast<uintptr_t>(m.p) % 8) == 0;
    };
    if (!(prm.C % 128 == 0 && (prm.C == 384 || prm.C == 768))) return false;   // NJ instantiations below
    if (!cl4(prm.feats) || !cl4(prm.feats_pos) || !cl2(prm.code) || !cl2(prm.code_pos)) return false;
    if (prm.K % 2 != 0 || prm.K > 72) return false;
    (void)precision;
    return true;
}

hipError_t launch_corr_fused(const FusedParams& prm_in, int precision, size_t sync_bytes, hipStream_t stream,
                             hipEvent_t* ev /* null or [4]: before the memset, before / after the tile kernel, end */)
{
    FusedParams prm = prm_in;
    const int n_tiles = prm.n_sets * prm.B;
    const int cus = device_cu_count();
    prm.n_owner = n_tiles < (cus & ~7) ? n_tiles : (cus & ~7);
    if (prm.n_owner < 1) prm.n_owner = 1;
    prm.timeout_ticks = (prm.debug & 64) ? 100 : 20000;           // 200 us of the 100 MHz clock
    const int stage = dense_stage_bytes(precision, prm.LDK);
    const int lds = dense_lds_bytes(precision, prm.LDK);
    if (ev) (void)hipEventRecord(ev[0], stream);
    hipError_t e = hipMemsetAsync(prm.anchor_cnt, 0, sync_bytes, stream);      // counters + granules of this call
    if (e != hipSuccess) return e;
    if (ev) (void)hipEventRecord(ev[1], stream);
    const dim3 grid(n_tiles), block(TILE_THREADS);
#define STEGO_FUSED_LAUNCH(PR, N)                                                                      \
    do {                                                                                               \
        e = ensure_dynamic_lds(reinterpret_cast<const void*>(&corr_fused_kernel<PR, N>), lds);         \
        if (e != hipSuccess) return e;                                                                 \
        hipLaunchKernelGGL((corr_fused_kernel<PR, N>), grid, block, lds, stream, prm, stage);          \
    } while (0)
    if (precision == PREC_F32) { if (prm.C == 384) STEGO_FUSED_LAUNCH(PREC_F32, 3); else STEGO_FUSED_LAUNCH(PREC_F32, 6); }
    else { if (prm.C == 384) STEGO_FUSED_LAUNCH(PREC_F16X3, 3); else STEGO_FUSED_LAUNCH(PREC_F16X3, 6); }
#undef STEGO_FUSED_LAUNCH
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (ev) (void)hipEventRecord(ev[2], stream);
    hipLaunchKernelGGL(corr_fused_scalars_kernel, dim3(1 + prm.n_neg * prm.B), dim3(NTHREADS), 0, stream, prm);
    if (ev) (void)hipEventRecord(ev[3], stream);
    return hipGetLastError();
}

}  // namespace stego
