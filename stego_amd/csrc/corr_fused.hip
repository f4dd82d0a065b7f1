// Fused forward of STEGO's ContrastiveCorrelationLoss for gfx950 (MI355X): ONE launch, nothing before or after it.
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
//   * phase 1 (the 4 waves of every workgroup's MFMA team, first thing; the launch covers EVERY CU, workgroups beyond the
//     tiles do nothing else): the anchor sets are sampled + normalised ONCE, 16 points per workgroup by the same XCD
//     affinity (anchor b is sampled on XCD b % 8, so the same fetch of image b serves its negatives), staged through the
//     A sides of the idle ring into ready-made operand stages and written with coalesced write-through (sc1) stores; one
//     counter per anchor publishes them.  The team syncs through an LDS counter, not s_barrier, because the gather team
//     does not take part: its first wave works out which tile the workgroup owns (ballots over perms), every gather wave
//     builds its taps and the team fills the B sides of the first four ring slots - none of which needs an anchor - while
//     phase 1 runs.  A tile waits for its anchor's counter (one polling wave, bounded spin; on timeout that wave recomputes
//     the anchor itself - same values, so duplicate stores are benign) and then streams the anchor operand with sc1
//     LDS-DMA copies.  The compact operand (196 KB) is what crosses XCDs, never the raw taps of a second image (550 KB);
//   * the main loop is a RING of four 32-channel stages (A side: LDS-DMA issued three stages ahead from inline asm, so
//     that neither the compiler nor a barrier drains it; B side: gathered by a team of 8 waves two stages ahead into
//     registers - tap offsets and weights register-resident - and committed one stage ahead); one raw s_barrier per
//     stage, which is also a scheduling fence.  With two 64-channel stages every barrier waited for a whole round trip
//     (period = latency + transfer);
//   * the codes go through the same ring as K-chunks of fp32 operands (format F) AHEAD of the features.  PREC_F32 multiplies
//     them on the fp32 MFMA (VALU rate: 2.0 us per chunk); PREC_F16X3 has the MFMA team split each fragment into fp16 hi / lo
//     halves in registers and use the fp16 matrix cores like the features (1.3 us per chunk; the B-side codes get a
//     power-of-two prescale per point).  The B-side codes are gathered by the gather team with the feature taps (same points),
//     their norm becomes a column scale of cd, the backward's context is written on the way;
//   * the batch-global old_mean (:331) - the reason the old forward had a third launch that re-read and re-wrote the
//     negative loss tensor - is a rendezvous INSIDE the launch: every negative tile publishes its sum(fd) as one
//     tagged 8-byte granule before it parks its tiles, and reads the B granules of its pair-set between the two parts of
//     the output sweep (measured 1.2 us mean / 1.7 us worst after the last publisher).  The spin
//     is bounded: a tile that gives up writes the loss without the old_mean term and flags itself;
//   * every workgroup ends with a ticket; the LAST one computes the three scalars from the per-tile sums in a fixed
//     order, repairs flagged tiles (normally none) and writes the hand-off words back to zero, so that a prepared
//     workspace serves launch after launch without a memset (stego_corr_workspace_prepare / stego_corr_fwd_prepared);
//   * the way out is branch-free: the accumulators are parked in the flat output layout with predicated ADDRESSES (a
//     branch per element had put each ds_write behind its own s_waitcnt: 64 serialised LDS round trips, 2.1 us per tile),
//     the row means are taken from the parked tile by the idle gather waves with independent loads, and the sweep runs in
//     two parts around the old_mean rendezvous (cd and the backward's w first, the negative loss after it).
//
// Arithmetic: PREC_F16X3 split-fp16 products (hi*hi + hi*lo + lo*hi, fp32 accumulate) for features and codes, PREC_F32 exact
// fp32 products, fp32 epilogue.  No atomics on data, fixed summation orders: bitwise repeatable.
#include "corr_tile.h"
#include "host_util.h"

// This file is compiled three times (the 44 instantiations of the kernel take two minutes in one translation unit): as itself - the host
// side and the even-K kernels of C = 384 / 768 (the BASELINE configs) - and, included by corr_fused_odd.hip / corr_fused_c192.hip with
// STEGO_FUSED_PART = 1 / 2, for the odd-K kernels and the C = 192 kernels with their two launch functions.  Same template, same code.
// STEGO_FUSED_PART = 3 (corr_fused_half.hip): the device functions only - the column-half kernel of small batches is built from them.
#ifndef STEGO_FUSED_PART
#define STEGO_FUSED_PART 0
#endif

namespace stego {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef f32x4 f32x4_u __attribute__((aligned(4)));      // 16-byte gather of an 8-byte aligned code pixel (K % 4 == 2)
typedef f32x2 f32x2_u __attribute__((aligned(4)));      // 8-byte gather of a code pixel that is only 4-byte aligned (odd K)

constexpr int ANCHOR_CNT_STRIDE = 64;           // counters 256 bytes apart: pollers of different anchors hit different channels

// ---- ring geometry
constexpr int KC2 = 32;                          // channels per stage
constexpr int RS_SIDE = 16384;                   // bytes of one operand side of a stage: H = [hi 128 x 64 B][lo 128 x 64 B], F = [128 x 128 B]
constexpr int RS_STAGE = 2 * RS_SIDE;            // A side, B side
constexpr int RS_NS = 4;                         // stages in the ring
// dynamic LDS carve
constexpr int RD_ROWMEAN = 0;                    // float[128]
constexpr int RD_RED = RD_ROWMEAN + TP * 4;      // float[64]
constexpr int RD_CSC = RD_RED + 64 * 4;          // float[128]  1 / ||b_j|| (features)
constexpr int RD_CSCC = RD_CSC + TP * 4;         // float[128]  1 / ||b_j|| (codes)
constexpr int RD_TAPOF = RD_CSCC + TP * 4;       // int4[128]   element offsets of the 4 taps in the B feature image
constexpr int RD_TAPOC = RD_TAPOF + TP * 16;     // int4[128]   ... in the B code image
constexpr int RD_TAPW = RD_TAPOC + TP * 16;      // float4[128]
constexpr int RD_RING = 8192;
static_assert(RD_TAPW + TP * 16 <= RD_RING, "table carve");
constexpr int RING_LDS_BYTES = RD_RING + (RS_NS * RS_STAGE > SM_TILES_BYTES ? RS_NS * RS_STAGE : SM_TILES_BYTES);

// Conflict-free ds_read_b128 without padding (LDS-DMA needs linear rows): the 16-byte unit u of row r sits at
// u ^ f(r).  H: 4 units per 64-byte row; F: 8 units per 128-byte row.
__device__ __forceinline__ int swz_h(int r, int u) { return r * 64 + ((u ^ ((r >> 2) & 3)) << 4); }
__device__ __forceinline__ int swz_f(int r, int u) { return r * 128 + ((u ^ ((r >> 1) & 7)) << 4); }

__device__ __forceinline__ unsigned long long lanes_below(int lane) { return lane ? (~0ull >> (64 - lane)) : 0ull; }

// ------------------------------------------------------------------------------------------ tile placement
// XCD preference of tile t = (p, b): the image its B side is gathered from, modulo 8.
template <int HS = 0>                                   // HS = 1: work items are column halves, item t = 2 * tile + half (corr_fused_half.hip)
__device__ __forceinline__ int tile_pref(const FusedParams& prm, int t_, int t_end)
{
    if (t_ >= t_end) return -1;
    const int t = t_ >> HS;
    const int B = prm.B;
    const int p = t / B, b = t - p * B;
    int src = b;
    if (p >= 2) src = (int)prm.perms[(size_t)(p - 2) * B + b];
    return src & 7;
}

// Bijection workgroups <-> tiles, computed redundantly by every wave (perms is device data: the host cannot build it
// without a sync) - per ROUND: the workgroups [w0, w0 + n_tiles) take the tiles [w0, w0 + n_tiles) (whole pair-sets, see the kernel).
// Workgroup w sits on XCD w % 8 (its slot: its rank among the round's workgroups of that XCD).  The r-th tile (in tile order) that prefers XCD x takes
// slot r of x while x has slots; tiles beyond that ("overflow") fill the slots other XCDs leave free, in order.
// Tiles are visited in blocks of 4 x 64 whose perms loads are issued together (one round trip for <= 256 tiles).
constexpr int ASSIGN_NB = 4;

// the loads of the first block, issued at the top of the kernel so that they fly together with phase 1's
template <bool WINDOW, int HS = 0>
__device__ __forceinline__ void assign_prefetch(const FusedParams& prm, int lane, int w0_, int n_tiles, int (&pref)[ASSIGN_NB])
{
    const int w0 = WINDOW ? w0_ : 0;
#pragma unroll
    for (int i = 0; i < ASSIGN_NB; ++i) pref[i] = tile_pref<HS>(prm, w0 + 64 * i + lane, w0 + n_tiles);
}

template <bool WINDOW, int HS = 0>                       // (false: one round, w0 = 0 folded - the common case keeps its old code)
__device__ int assign_tile(const FusedParams& prm, int me, int lane, int w0_, int n_tiles, const int (&pref0)[ASSIGN_NB])
{
    constexpr int NB = ASSIGN_NB;
    const int w0 = WINDOW ? w0_ : 0;
    const int x0 = w0 & 7;                               // XCD of the round's first workgroup
    auto nslots = [&](int x) { return (n_tiles - ((x - x0) & 7) + 7) >> 3; };      // workgroups of the round on XCD x
    int cnt[8];
#pragma unroll
    for (int x = 0; x < 8; ++x) cnt[x] = 0;
    int pref[NB];
    for (int c0 = 0; c0 < n_tiles; c0 += 64 * NB) {
#pragma unroll
        for (int i = 0; i < NB; ++i) pref[i] = c0 == 0 ? pref0[i] : tile_pref<HS>(prm, w0 + c0 + 64 * i + lane, w0 + n_tiles);
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int x = 0; x < 8; ++x) cnt[x] += __builtin_popcountll(__ballot(pref[i] == x));
    }
    const int xme = me & 7, rme = (me - w0) >> 3;        // ((me - w0) - ((xme - x0) & 7)) / 8: the remainder is < 8
    int cnt_me = 0, k = 0;
#pragma unroll
    for (int x = 0; x < 8; ++x) {
        const int nslot = nslots(x);
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
        for (int i = 0; i < NB; ++i) pref[i] = c0 == 0 ? pref0[i] : tile_pref<HS>(prm, w0 + c0 + 64 * i + lane, w0 + n_tiles);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            int myrank = 0, myslots = 0;
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const unsigned long long m = __ballot(pref[i] == x);
                if (pref[i] == x) { myrank = base[x] + __builtin_popcountll(m & below); myslots = nslots(x); }
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
    return w0 + __builtin_amdgcn_readfirstlane(result);
}

// ------------------------------------------------------------------------------------------ phase 1: anchor sets
// Lane hl of a half-wave owns, of ONE sample point: feature channels 128 j + 4 hl .. + 3 (j < NJ) and code channels
// 2 hl, 2 hl + 1, 64 + hl and 96 + hl (K <= 128).
struct CodeTaps {
    f32x2 a[4];
    float b[4];
    float c[4];
};

// index of point q = (h, w) in a coords image: it samples coords[w][h] (sample() permutes the grid, modules.py:288)
__device__ __forceinline__ int coord_index(const FusedParams& prm, int q)
{
    const int qq = q < prm.P ? q : 0;
    const int hh = qq / prm.S, ww = qq - hh * prm.S;
    return (ww * prm.S + hh) * 2;
}

__device__ __forceinline__ void point_taps(const FusedParams& prm, const f32x2 cxy, int q, int4& yx, float4& w)
{
    yx = make_int4(0, 0, 0, 0);
    w = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < prm.P) make_taps(cxy[0], cxy[1], prm.H, prm.W, yx, w);
}

// Staging area of phase 1, rows = the points of this pass in image order.  Phase 1 belongs to the MFMA team (round 2b): the gather
// team fills the B sides of the four ring slots meanwhile, so the staging lives in the four A sides (16 KB pieces, RS_STAGE apart),
// which nothing touches before the team's first LDS-DMA:
//   pieces 0-1: features [stage s2][plane][ROWS][row bytes]   (H: 2 planes of 64-byte rows, F: 1 plane of 128-byte rows)
//   piece 2:    codes, ring format F [stage][ROWS][128 B]     (operand of the forward)
//   piece 3:    codes, context rows  [ROWS][LDK floats]       (what the backward reads)
// so that every plane leaves as ONE contiguous run per anchor (scattered sc1 stores straight from the registers ran at
// 0.4 TB/s chip-wide: 22 us for 8.6 MB).
template <int NJ, int PREC, bool LIGHT = false>
struct P1Layout {
    static constexpr int NCH2 = NJ == 2 ? 6 : 4 * NJ;          // feature stages of 32 channels: C / 32 (NJ = 2 is C = 192: the second 128-channel
                                                               // group of a half-wave is half empty - lanes 16..31 carry zeros and store nothing)
    static constexpr int PLANES = PREC == PREC_F16X3 ? 2 : 1;
    static constexpr int RB = PREC == PREC_F16X3 ? 64 : 128;
    static constexpr int G = NJ <= 3 ? 2 : 1;                  // points per half-wave and pass (register budget: 48 NJ G)
    static constexpr int MROWS = 4 * 2 * G;                    // rows the four waves of the MFMA team take per pass
    static constexpr int XW = NJ <= 3 ? 2 : 0;                 // gather waves (8, 9) that take two overflow rows each in the FIRST pass
    // LIGHT (round 4): a workgroup WITHOUT a gather stream of its own - a self-correlation tile or a tile-less helper - samples with all
    // twelve waves (2 G rows each; the eight gather waves a second chunk of 2 when the registers allow G = 2) and stages in the WHOLE
    // ring, linearly; the context rows of the backward go straight to memory (nobody reads them in this launch: plain stores).  It
    // carries phase 1 for its XCD (see the kernel)
    static constexpr int XB = NJ <= 3 ? 1 : 0;                 // second chunk of the gather waves
    static constexpr int ROWS = LIGHT ? 12 * 2 * G + 8 * 2 * XB : MROWS + 2 * XW;     // rows of a pass = geometry of the staging area
    static constexpr int STAGE_BYTES = ROWS * 128;             // one feature stage of a pass (either format)
    static constexpr int SPP = LIGHT ? 1 : RS_SIDE / STAGE_BYTES;        // feature stages per piece
    static_assert(LIGHT || NCH2 <= 2 * SPP, "feature stages fit pieces 0-1");
    static_assert(LIGHT || (4 * ROWS * 128 <= RS_SIDE && ROWS * (128 + 4) * 4 <= RS_SIDE), "code stages fit piece 2, context rows piece 3");
    static constexpr int CF = LIGHT ? NCH2 * STAGE_BYTES : 2 * RS_STAGE;
    static constexpr int CX = LIGHT ? 0 : 3 * RS_STAGE;        // (LIGHT: not staged)
    static_assert(!LIGHT || CF + 4 * ROWS * 128 <= RS_NS * RS_STAGE, "the light staging fits the ring");
    __device__ static __forceinline__ int feat_plane(int s2, int pp)
    {
        if constexpr (LIGHT) return s2 * STAGE_BYTES + pp * ROWS * RB;
        else return (s2 / SPP) * RS_STAGE + (s2 % SPP) * STAGE_BYTES + pp * ROWS * RB;
    }
};

// bilinear blend of the four taps with a FIXED rounding sequence (one multiply, three fmas)
__device__ __forceinline__ float blend4(const float4 w, float t0, float t1, float t2, float t3)
{
    return __builtin_fmaf(w.w, t3, __builtin_fmaf(w.z, t2, __builtin_fmaf(w.y, t1, w.x * t0)));
}

// Sum over the 32 lanes of a half-wave, in every lane: four DPP steps inside each 16-lane row + ONE exchange between the two rows.  The
// butterfly of five __shfl_xor is five dependent ds_bpermute round trips through the LDS crossbar - with the split's lane swaps 32 per wave
// and pass, 5.5 us between "taps landed" and "rows staged" under the load of phase 1 (stamps, profiles/r04h).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float half_wave_sum(float v, bool butterfly)
{
    if (butterfly) {                                  // (debug 4096: the reduction of rounds 2-3)
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        return v;
    }
    v += dpp_mov<0xB1>(v);                            // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);                            // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);                           // row_half_mirror: quads 0 <-> 1, 2 <-> 3
    v += dpp_mov<0x140>(v);                           // row_mirror: the two halves of a row
    return v + __shfl_xor(v, 16, 64);                 // the other row of the half-wave
}

// Sum over the 64 lanes of a wave in ONE fixed order (the tree of half_wave_sum in each half, then the two halves): what a tile's
// rendezvous and the last workgroup's tail both use for the per-tile sums of a pair-set, 64 images at a time - the two must agree to the bit
// (saved_mean is what the backward adds to w; the repair path rewrites losses with the tail's old_mean), and a sequential sum of 32 values
// was 32 dependent readlane + add steps on every negative tile's way out and 0.7 us of the tail (round 5).
__device__ __forceinline__ float wave_tree_sum(float v)
{
    v = half_wave_sum(v, false);
    return v + __shfl_xor(v, 32, 64);
}

// Samples the rows [lr0, lr0 + 2 G) of the current pass (this wave's share; global point index = blk0 + row) into the
// staging area.
// CH (the column-half kernel in F16X3 mode): the code operand leaves in format H like the features - split into fp16 hi / lo planes of
// 32-channel stages (kper channels + zero padding) - so that the code stages of that kernel are plain fragment reads + MFMAs (the
// in-register split of format F costs its MFMA team 1 us per code stage: profiles/r06c_half_ablations.txt).
template <int NJ, int PREC, int NKCT, int G, bool LIGHT = false, bool ODDK = false, bool CH = false>
__device__ __forceinline__ void p1_sample_rows(const FusedParams& prm, int xa, int blk0, int end, int lr0, int lane,
                                               unsigned char* lds, unsigned long long* tsd)
{
    typedef P1Layout<NJ, PREC, LIGHT> LY;
    constexpr int ROWS = LY::ROWS;
    const int hl = lane & 31, hw = lane >> 5;
    const int crow = prm.LDK * 4;
    const bool bfly = (prm.debug & 4096) != 0;
    unsigned char* lds_cf = lds + LY::CF;
    unsigned char* lds_cx = lds + LY::CX;
    const MapV mf = prm.feats, mc = prm.code;
    int4 yx[G];
    float4 w[G];
    int ba[G], q[G];
    bool act[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int idx = blk0 + lr0 + 2 * g + hw;
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
    const int c1 = 64 + hl < prm.K ? 64 + hl : 0;
    const int c2 = 96 + hl < prm.K ? 96 + hl : 0;
    const int c0 = 2 * hl < prm.K ? 2 * hl : 0;
    const int cshift = (ODDK && 2 * hl < prm.K && 2 * hl + 1 >= prm.K) ? 1 : 0;        // my pair straddles the end of an odd-K pixel (K >= 3)
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
            if (NJ == 2 && j == 1 && hl >= 16) {      // C = 192: channels 192.. do not exist (the next pixel's, past the tensor for the last one)
                t[g][j][0] = t[g][j][1] = t[g][j][2] = t[g][j][3] = f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
                t[g][j][0] = p0[32 * j]; t[g][j][1] = p1[32 * j]; t[g][j][2] = p2[32 * j]; t[g][j][3] = p3[32 * j];
            }
        }
    }
    auto code_loads = [&](int g) {
        const int4 oc = taps_to_offsets(yx[g], mc.sh, mc.sw);
        const float* cimg = mc.p + (long long)ba[g] * mc.sn;
        // (odd K: the last pair (K - 1, K) would read one float past the pixel - past the TENSOR for its last pixel: that lane loads
        // (K - 2, K - 1) and keeps the second; pixels of an odd K are only 4-byte aligned)
        if constexpr (ODDK) {
            ct[g].a[0] = *reinterpret_cast<const f32x2_u*>(cimg + oc.x + c0 - cshift);
            ct[g].a[1] = *reinterpret_cast<const f32x2_u*>(cimg + oc.y + c0 - cshift);
            ct[g].a[2] = *reinterpret_cast<const f32x2_u*>(cimg + oc.z + c0 - cshift);
            ct[g].a[3] = *reinterpret_cast<const f32x2_u*>(cimg + oc.w + c0 - cshift);
            if (cshift) {
#pragma unroll
                for (int tq = 0; tq < 4; ++tq) ct[g].a[tq] = f32x2{ct[g].a[tq][1], 0.f};
            }
        } else {
            ct[g].a[0] = *reinterpret_cast<const f32x2*>(cimg + oc.x + c0);
            ct[g].a[1] = *reinterpret_cast<const f32x2*>(cimg + oc.y + c0);
            ct[g].a[2] = *reinterpret_cast<const f32x2*>(cimg + oc.z + c0);
            ct[g].a[3] = *reinterpret_cast<const f32x2*>(cimg + oc.w + c0);
        }
        // (K > 64 needs three K-chunks, K > 96 four: the registers of the unused groups do not exist)
        if constexpr (NKCT > 2) { ct[g].b[0] = cimg[oc.x + c1]; ct[g].b[1] = cimg[oc.y + c1]; ct[g].b[2] = cimg[oc.z + c1]; ct[g].b[3] = cimg[oc.w + c1]; }
        else { ct[g].b[0] = ct[g].b[1] = ct[g].b[2] = ct[g].b[3] = 0.f; }
        if constexpr (NKCT > 3) { ct[g].c[0] = cimg[oc.x + c2]; ct[g].c[1] = cimg[oc.y + c2]; ct[g].c[2] = cimg[oc.z + c2]; ct[g].c[3] = cimg[oc.w + c2]; }
        else { ct[g].c[0] = ct[g].c[1] = ct[g].c[2] = ct[g].c[3] = 0.f; }
    };
#pragma unroll
    for (int g = 0; g < G; ++g) code_loads(g);
    if (tsd) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tsd[1] = __builtin_amdgcn_s_memrealtime();
    }
    auto feat_part = [&](int g) {
        const int lr = lr0 + 2 * g + hw;                          // row inside the staging area
        const int qq = q[g];
        const bool valid = act[g] && qq < prm.P;
        const float4 wg = w[g];
        f32x4 v[NJ];
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // (explicit fma chain: the two inlined instances of this code - g = 0, 1 - must round alike, or an anchor row's bits
                // would depend on which wave slot sampled it)
                const float x = blend4(wg, t[g][j][0][e], t[g][j][1][e], t[g][j][2][e], t[g][j][3][e]);
                v[j][e] = x;
                ss = __builtin_fmaf(x, x, ss);
            }
        ss = half_wave_sum(ss, bfly);
        // F.normalize eps (modules.py:276); padding points (zero taps) are written as zeros
        const float inv = valid ? __builtin_amdgcn_rcpf(fmaxf(sqrtf(ss), 1e-10f)) : 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const f32x4 vn = v[j] * inv;
            if constexpr (PREC == PREC_F32) {
                const int c = 128 * j + 4 * hl;
                const int u = ((c & 31) >> 2) ^ ((qq >> 1) & 7);
                if (NJ != 2 || c < 32 * LY::NCH2) *reinterpret_cast<f32x4*>(lds + LY::feat_plane(c >> 5, 0) + lr * 128 + u * 16) = vn;
            } else {
                unsigned h0, l0, h1, l1;
                split_f16_pair(vn[0], vn[1], h0, l0);
                split_f16_pair(vn[2], vn[3], h1, l1);
                const bool odd = hl & 1;                          // even lanes collect the hi halves of a lane pair, odd the lo
                const unsigned r0 = bfly ? __shfl_xor(odd ? h0 : l0, 1, 64) : (unsigned)__builtin_amdgcn_update_dpp(0, (int)(odd ? h0 : l0), 0xB1, 0xF, 0xF, true);
                const unsigned r1 = bfly ? __shfl_xor(odd ? h1 : l1, 1, 64) : (unsigned)__builtin_amdgcn_update_dpp(0, (int)(odd ? h1 : l1), 0xB1, 0xF, 0xF, true);
                const u32x4 d = odd ? u32x4{r0, r1, l0, l1} : u32x4{h0, h1, r0, r1};
                const int c = 128 * j + 4 * (hl & ~1);
                const int u = ((c & 31) >> 3) ^ ((qq >> 2) & 3);
                if (NJ != 2 || c < 32 * LY::NCH2) *reinterpret_cast<u32x4*>(lds + LY::feat_plane(c >> 5, odd ? 1 : 0) + lr * 64 + u * 16) = d;
            }
        }
    };
    auto code_part = [&](int g) {
        const int lr = lr0 + 2 * g + hw;
        const int qq = q[g];
        const bool valid = act[g] && qq < prm.P;
        const float4 wg = w[g];
        // ---- code: normalised; once as K-chunk operand stages (forward), once as a context row (backward)
        f32x2 r0 = f32x2{blend4(wg, ct[g].a[0][0], ct[g].a[1][0], ct[g].a[2][0], ct[g].a[3][0]),
                         blend4(wg, ct[g].a[0][1], ct[g].a[1][1], ct[g].a[2][1], ct[g].a[3][1])};
        float r1 = blend4(wg, ct[g].b[0], ct[g].b[1], ct[g].b[2], ct[g].b[3]);
        if (2 * hl >= prm.K) r0 = f32x2{0.f, 0.f};
        float r2 = blend4(wg, ct[g].c[0], ct[g].c[1], ct[g].c[2], ct[g].c[3]);
        if (64 + hl >= prm.K) r1 = 0.f;
        if (96 + hl >= prm.K) r2 = 0.f;
        float cs2 = __builtin_fmaf(r2, r2, __builtin_fmaf(r1, r1, __builtin_fmaf(r0[1], r0[1], r0[0] * r0[0])));
        cs2 = half_wave_sum(cs2, bfly);
        const float nr = valid ? sqrtf(cs2) : 0.f;
        const float cinv = valid ? __builtin_amdgcn_rcpf(fmaxf(nr, 1e-10f)) : 0.f;
        r0 = r0 * cinv;
        r1 = r1 * cinv;
        r2 = r2 * cinv;
        const int kall = prm.NKC * prm.kper;
        if constexpr (CH) {
            // staging: [stage][plane][ROWS][64 B]; a lane pair's two channels are one packed fp16 pair (4-byte stores, never 2)
            auto put = [&](int k, float x, float y) {
                const int sc = (k >= prm.kper) + (k >= 2 * prm.kper) + (k >= 3 * prm.kper), col = k - sc * prm.kper;
                unsigned h, l;
                split_f16_pair(x, y, h, l);
                unsigned char* d = lds_cf + ((sc * 2) * ROWS + lr) * 64 + (((col >> 3) ^ ((qq >> 2) & 3)) << 4) + (col & 7) * 2;
                *reinterpret_cast<unsigned*>(d) = h;
                *reinterpret_cast<unsigned*>(d + ROWS * 64) = l;
            };
            if (2 * hl < kall) put(2 * hl, r0[0], r0[1]);
            const float r1n = dpp_mov<0xB1>(r1), r2n = dpp_mov<0xB1>(r2);             // my neighbour's channel (quad_perm [1,0,3,2])
            if (!(hl & 1) && 64 + hl < kall) put(64 + hl, r1, r1n);
            if (!(hl & 1) && 96 + hl < kall) put(96 + hl, r2, r2n);
            // channels kper .. 31 of every stage: zeros (nothing masks them in the MFMAs; fp16 garbage may be NaN)
            const int pad_units = 4 - (prm.kper >> 3);
            for (int i = hl; i < prm.NKC * 2 * pad_units; i += 32) {
                const int pl = i / pad_units, u = (prm.kper >> 3) + (i - pl * pad_units);
                *reinterpret_cast<u32x4*>(lds_cf + (pl * ROWS + lr) * 64 + ((u ^ ((qq >> 2) & 3)) << 4)) = u32x4{0u, 0u, 0u, 0u};
            }
        } else {
        if (2 * hl < kall) {
            const int k = 2 * hl, sc = (k >= prm.kper) + (k >= 2 * prm.kper) + (k >= 3 * prm.kper), col = k - sc * prm.kper;
            *reinterpret_cast<f32x2*>(lds_cf + (sc * ROWS + lr) * 128 + (((col >> 2) ^ ((qq >> 1) & 7)) << 4) + (col & 3) * 4) = r0;
        }
        if (64 + hl < kall) {
            const int k = 64 + hl, sc = (k >= prm.kper) + (k >= 2 * prm.kper) + (k >= 3 * prm.kper), col = k - sc * prm.kper;
            *reinterpret_cast<float*>(lds_cf + (sc * ROWS + lr) * 128 + (((col >> 2) ^ ((qq >> 1) & 7)) << 4) + (col & 3) * 4) = r1;
        }
        }
        unsigned char* cx_row = lds_cx + lr * crow;
        if constexpr (LIGHT) cx_row = reinterpret_cast<unsigned char*>(prm.cs + ((size_t)ba[g] * TP + qq) * prm.LDK);     // straight to memory
        const bool cx_on = !LIGHT || act[g];
        if (cx_on && 2 * hl < prm.KQ) *reinterpret_cast<f32x2*>(cx_row + 8 * hl) = r0;
        if (cx_on && 64 + hl < prm.KQ) *reinterpret_cast<float*>(cx_row + 4 * (64 + hl)) = r1;
        if constexpr (!CH) {
        if (96 + hl < kall) {
            const int k = 96 + hl, sc = (k >= prm.kper) + (k >= 2 * prm.kper) + (k >= 3 * prm.kper), col = k - sc * prm.kper;
            *reinterpret_cast<float*>(lds_cf + (sc * ROWS + lr) * 128 + (((col >> 2) ^ ((qq >> 1) & 7)) << 4) + (col & 3) * 4) = r2;
        }
        }
        if (cx_on && 96 + hl < prm.KQ) *reinterpret_cast<float*>(cx_row + 4 * (96 + hl)) = r2;
        if (act[g] && hl == 0) prm.nrm[(size_t)ba[g] * TP + qq] = nr;
    };
#pragma unroll
    for (int g = 0; g < G; ++g) feat_part(g);
#pragma unroll
    for (int g = 0; g < G; ++g) code_part(g);
}

// Writes staged rows [row0, row0 + nrows) (global point index blk0 + row) out: the planes `first`, `first + step`, ...
// of the plane list {feature planes, code stages, context} by this wave, 16 bytes per lane, write-through.
template <int NJ, int PREC, bool LIGHT = false, bool CH = false>
__device__ __forceinline__ void p1_copy_out(const FusedParams& prm, int xa, int blk0, int row0, int nrows, int first,
                                            int step, int lane, const unsigned char* lds, __amdgpu_buffer_rsrc_t fs_rsrc,
                                            __amdgpu_buffer_rsrc_t csf_rsrc, __amdgpu_buffer_rsrc_t cs_rsrc)
{
    typedef P1Layout<NJ, PREC, LIGHT> LY;
    constexpr int NFP = LY::NCH2 * LY::PLANES, ROWS = LY::ROWS;
    const int crow = prm.LDK * 4;
    const unsigned char* lds_cf = lds + LY::CF;
    const unsigned char* lds_cx = lds + LY::CX;
    constexpr int CPL = CH ? 2 : 1, CRB = CH ? 64 : 128;             // planes and row bytes of a code stage (format H: hi + lo of 64-byte rows)
    const int ncode = prm.NKC * CPL;
    const int nplanes = NFP + ncode + 1;
    // all rows of a call belong to one anchor (the caller splits a pass at an anchor boundary)
    const int set = xa + 8 * ((blk0 + row0) >> 7), q0 = (blk0 + row0) & (TP - 1);
    int pl = first;
    for (; pl < NFP; pl += step) {                                   // feature planes: 16-byte units, a power of two per row
        constexpr int UPR = LY::RB / 16;
        const int s2 = pl / LY::PLANES, pp = pl - s2 * LY::PLANES;
        const unsigned char* src = lds + LY::feat_plane(s2, pp) + row0 * LY::RB;
        const unsigned base = (unsigned)(((size_t)set * LY::NCH2 + s2) * RS_SIDE + pp * 8192 + q0 * LY::RB);
        for (int u = lane; u < nrows * UPR; u += 64)                 // (staged rows are contiguous, and so is the run they leave as)
            __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(src + u * 16), fs_rsrc, base + u * 16, 0, 16);
    }
    for (; pl < NFP + ncode; pl += step) {                           // code operand stages
        const int cp = pl - NFP, sc = cp / CPL, pp = cp - sc * CPL;
        const unsigned char* src = lds_cf + (cp * ROWS + row0) * CRB;
        const unsigned base = (unsigned)(((size_t)set * prm.NKC + sc) * RS_SIDE + pp * 8192 + q0 * CRB);
        for (int u = lane; u < nrows * (CRB / 16); u += 64)
            __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(src + u * 16), csf_rsrc, base + u * 16, 0, 16);
    }
    if (!LIGHT && pl < nplanes) {                                    // context rows (LDK floats: whole 16-byte units)
        const unsigned char* src = lds_cx + row0 * crow;
        const unsigned base = (unsigned)(((size_t)set * TP + q0) * crow);
        for (int u = lane; u < (nrows * crow) >> 4; u += 64)
            __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(src + u * 16), cs_rsrc, base + u * 16, 0, 16);
    }
}

// Barrier of a team of waves inside the workgroup, through an LDS counter that only grows (s_barrier would take in the waves of
// the other team): everything this wave did in LDS before is visible to whoever sees the count (the LDS executes a wave's
// operations in order).  `target` = waves x (barriers so far).
__device__ __forceinline__ void team_arrive(unsigned* cnt, int lane)     // a guest: counted, does not wait
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void team_barrier(unsigned* cnt, unsigned target, int lane)
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}

// publish `n` finished points starting at flat index `first` of the list "anchors xa, xa + 8, ..." (one lane)
__device__ __forceinline__ void p1_publish(const FusedParams& prm, int xa, int first, int n)
{
    while (n > 0) {
        const int a = first >> 7;
        const int m = min(n, ((a + 1) << 7) - first);
        __hip_atomic_fetch_add(prm.anchor_cnt + (size_t)(xa + 8 * a) * ANCHOR_CNT_STRIDE, (unsigned)m, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        first += m;
        n -= m;
    }
}

// ------------------------------------------------------------------------------------------ ring: MFMA side
// LDS-DMA of one KiB from inline asm: the compiler neither counts it (no vmcnt(0) in front of the next ds_read or
// barrier) nor reorders it; the waits are the explicit counted s_waitcnt below.  M0 = LDS byte address of the piece.
__device__ __forceinline__ void dma_piece_sc1(const unsigned char* gsrc_lane, unsigned lds_addr)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc_lane), "s"(lds_addr) : "memory");
}

__device__ __forceinline__ unsigned lds_address(const void* p)
{
    return (unsigned)(uintptr_t)((__attribute__((address_space(3))) const unsigned char*)p);
}

// The ring's barrier.  It is also a SCHEDULING fence: the software pipeline of the gather team (commit the stage gathered
// two barriers ago, then re-issue its registers) only exists if the compiler keeps each stage's work between its two
// barriers.  Without the fence it hoisted the blend of the NEXT stage above the barrier, which needs that stage's gathers:
// an s_waitcnt vmcnt(0) per stage, i.e. no prefetch distance at all (1.2 -> 2.0 us per stage).
__device__ __forceinline__ void ring_barrier()
{
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
// one feature stage in format H: a.b ~= ah.bh + ah.bl + al.bh over 32 channels (2 k-steps of 16)
__device__ __forceinline__ void mma_stage_h(const unsigned char* __restrict__ As, const unsigned char* __restrict__ Bs,
                                            f32x16 (&acc)[2][2], int lane, int wr, int wc)
{
    const int r = lane & 31, half = lane >> 5;
    const int ra0 = 64 * wr + r, ra1 = ra0 + 32, rb0 = 64 * wc + r, rb1 = rb0 + 32;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int u = 2 * ks + half;
        const f16x8 ah0 = *reinterpret_cast<const f16x8*>(As + swz_h(ra0, u)), al0 = *reinterpret_cast<const f16x8*>(As + 8192 + swz_h(ra0, u));
        const f16x8 ah1 = *reinterpret_cast<const f16x8*>(As + swz_h(ra1, u)), al1 = *reinterpret_cast<const f16x8*>(As + 8192 + swz_h(ra1, u));
        const f16x8 bh0 = *reinterpret_cast<const f16x8*>(Bs + swz_h(rb0, u)), bl0 = *reinterpret_cast<const f16x8*>(Bs + 8192 + swz_h(rb0, u));
        const f16x8 bh1 = *reinterpret_cast<const f16x8*>(Bs + swz_h(rb1, u)), bl1 = *reinterpret_cast<const f16x8*>(Bs + 8192 + swz_h(rb1, u));
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

// one stage in format F (exact fp32) over k < kper channels
__device__ __forceinline__ void mma_stage_f(const unsigned char* __restrict__ As, const unsigned char* __restrict__ Bs, int kper,
                                            f32x16 (&acc)[2][2], int lane, int wr, int wc)
{
    const int r = lane & 31, half = lane >> 5;
    const int ra0 = 64 * wr + r, ra1 = ra0 + 32, rb0 = 64 * wc + r, rb1 = rb0 + 32;
    for (int kk = 0; kk < kper; kk += 8) {
        const int u = (kk >> 2) + half;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(As + swz_f(ra0, u));
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(As + swz_f(ra1, u));
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(Bs + swz_f(rb0, u));
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(Bs + swz_f(rb1, u));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[1][1], 0, 0, 0);
        }
    }
}

// one CODE stage for the split-fp16 matrix cores: operands stay in format F (fp32 in LDS: phase 1, the gather team's commit and the
// backward's context are unchanged), each wave splits its fragments into hi / lo halves in registers (8 channels of a row = two
// 16-byte units) - ~300 VALU instructions against 1.3 us of fp32 MFMAs (which run at the VALU rate on gfx950).  The B side is
// prescaled by the gather team (a power of two per point: exact), the A side is normalised.  Channels >= kper of a 16-wide
// k-step are masked: the pad of the anchor operand is whatever the workspace held.
__device__ __forceinline__ void split_f16x8(const f32x4 x0, const f32x4 x1, bool live, f16x8& hi, f16x8& lo)
{
    unsigned h[4], l[4];
    split_f16_pair(x0[0], x0[1], h[0], l[0]);
    split_f16_pair(x0[2], x0[3], h[1], l[1]);
    split_f16_pair(x1[0], x1[1], h[2], l[2]);
    split_f16_pair(x1[2], x1[3], h[3], l[3]);
    const u32x4 hv = live ? u32x4{h[0], h[1], h[2], h[3]} : u32x4{0u, 0u, 0u, 0u};
    const u32x4 lv = live ? u32x4{l[0], l[1], l[2], l[3]} : u32x4{0u, 0u, 0u, 0u};
    hi = __builtin_bit_cast(f16x8, hv);
    lo = __builtin_bit_cast(f16x8, lv);
}

__device__ __forceinline__ void mma_stage_fh(const unsigned char* __restrict__ As, const unsigned char* __restrict__ Bs, int kper,
                                             f32x16 (&acc)[2][2], int lane, int wr, int wc)
{
    const int r = lane & 31, half = lane >> 5;
    const int ra0 = 64 * wr + r, ra1 = ra0 + 32, rb0 = 64 * wc + r, rb1 = rb0 + 32;
    for (int ks = 0; 16 * ks < kper; ++ks) {
        const int u = 4 * ks + 2 * half;
        const bool live = 16 * ks + 8 * half < kper;
        f16x8 ah0, al0, ah1, al1, bh0, bl0, bh1, bl1;
        split_f16x8(*reinterpret_cast<const f32x4*>(As + swz_f(ra0, u)), *reinterpret_cast<const f32x4*>(As + swz_f(ra0, u + 1)), live, ah0, al0);
        split_f16x8(*reinterpret_cast<const f32x4*>(As + swz_f(ra1, u)), *reinterpret_cast<const f32x4*>(As + swz_f(ra1, u + 1)), live, ah1, al1);
        split_f16x8(*reinterpret_cast<const f32x4*>(Bs + swz_f(rb0, u)), *reinterpret_cast<const f32x4*>(Bs + swz_f(rb0, u + 1)), live, bh0, bl0);
        split_f16x8(*reinterpret_cast<const f32x4*>(Bs + swz_f(rb1, u)), *reinterpret_cast<const f32x4*>(Bs + swz_f(rb1, u + 1)), live, bh1, bl1);
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

// ------------------------------------------------------------------------------------------ ring: gather side
// The gather team is instruction-bound (measured: 1.1 us of blend / split / LDS work + 0.6 us of load issue per stage with
// four waves, against 0.7 us for the MFMA team), so it gets GW = 8 of the workgroup's 12 waves.
// Thread mapping (64 GW threads): g8 = gt & 7 owns 16 bytes (4 channels) of a point's 32-channel stage, prow = gt >> 3;
// item j (0..GI-1) is point q = GP j + prow.  A wave touches the points {GP j + 8 w + i}: it builds exactly those
// tap-table entries itself (no barrier before the first gather).
constexpr int GW = 8;                            // gather waves
constexpr int GP = 8 * GW;                       // points per item across the team
constexpr int GI = TP / GP;                      // items per lane and stage
constexpr int FUSED_WAVES = 4 + GW;
constexpr int FUSED_THREADS = 64 * FUSED_WAVES;
struct GSet {
    f32x4 tv[GI][4];         // [item][tap]
};

// ------------------------------------------------------------------------------------------ the last workgroup of the launch
// The three scalars and the saved means from the per-tile sums in image order (modules.py:331,393,395):
//   old_mean_p = sum_b sum(fd) / (B P^2);   mean(loss_p) = (sum lp - old_mean_p * sum clamp) / (B P^2);
// tiles whose rendezvous gave up (applied == 0) get their old_mean term now: loss = lp - old_mean * clamp(cd), the
// same fma the tile itself uses; and the hand-off words go back to zero for the next launch on this workspace.
__device__ __forceinline__ void last_workgroup_tail(const FusedParams& prm, float* Tfd, float* timed_out, int tid, int n_tiles,
                                                    unsigned long long* ts)
{
    if (ts && tid == 0) ts[12] = __builtin_amdgcn_s_memrealtime();        // (debug 256: the tail's own stamps, over the epilogue's)
    const int B = prm.B, P2 = prm.P * prm.P;
    const float cmin = prm.cmin, cmax = prm.cmax;
    unsigned long long* gst = prm.gran + n_tiles;                  // [n_tiles][3]: sum lp, sum clamp, old_mean applied
    float* sst = Tfd;                            // [n_tiles][4] staged copy of the sums (the ring is dead)
    float* som = sst + n_tiles * 4;              // [n_sets] old_mean per pair-set, [n_sets] sum of its loss
    float* sums3 = som + 2 * prm.n_sets;         // [n_sets][3]
    bool mine = false;                           // one of my flags says "rendezvous gave up"
    {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (int i = tid; i < n_tiles * 4; i += FUSED_THREADS) {
            const int t = i >> 2, k = i & 3;
            const unsigned long long* src = k == 0 ? prm.gran + t : gst + (size_t)t * 3 + (k - 1);
            unsigned long long x;
            for (;;) {
                x = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((x >> 32) == 1ull) break;
                // A granule that does not arrive within the bound (every workgroup has taken its ticket, i.e. parked its tile: what is left of
                // it is two sweeps) means the launch is broken, not slow: the scalars become NaN - loudly wrong - instead of folding in a stale value.
                if ((long long)(__builtin_amdgcn_s_memrealtime() - t0) > prm.timeout_ticks) { timed_out[0] = 1.f; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            const float v = __builtin_bit_cast(float, (unsigned)x);
            sst[i] = v;
            mine |= k == 3 && t >= 2 * B && v == 0.f;
        }
    }
    // one barrier: the staged sums are complete AND the vote on the repair path (a serial scan of the flags cost 160 dependent LDS
    // reads = 6.7 us at the very end of every launch)
    const bool repair = __syncthreads_or(mine) && prm.pointwise;
    if (ts && tid == 0) ts[13] = __builtin_amdgcn_s_memrealtime();
    // the per-pair-set sums (sum fd, sum lp, sum clamp over the images, wave_tree_sum's order): (pair-set, quantity) idx by wave idx % 12
    for (int idx = tid >> 6; idx < 3 * prm.n_sets; idx += FUSED_WAVES) {
        const int ps = idx / 3, k = idx - 3 * ps, lane = tid & 63;
        const float* st = sst + (size_t)ps * B * 4 + k;
        float acc = 0.f;
        for (int i0 = 0; i0 < B; i0 += 64) acc += wave_tree_sum(i0 + lane < B ? st[(i0 + lane) * 4] : 0.f);
        if (lane == 0) sums3[idx] = acc;
    }
    if (tid >= 64) {
        // every hand-off word has been copied: waves 1.. write them back to zero
        constexpr int NZ = FUSED_THREADS - 64;
        for (int i = tid - 64; i < B; i += NZ) {
            __hip_atomic_store(prm.anchor_cnt + (size_t)i * ANCHOR_CNT_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        for (int i = tid - 64; i < n_tiles * 4; i += NZ)
            __hip_atomic_store(prm.gran + i, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();                             // sums3 complete
    if (tid < 64) {
        const float inv_cnt = 1.f / ((float)B * (float)P2);
        for (int ps = tid; ps < prm.n_sets; ps += 64) {
            const float fsum = sums3[3 * ps], lsum = sums3[3 * ps + 1], csum = sums3[3 * ps + 2];
            const float omp = prm.pointwise ? fsum * inv_cnt : 0.f;
            som[ps] = fsum * inv_cnt;
            som[prm.n_sets + ps] = lsum - omp * csum;               // sum of this pair-set's loss
            const float poison = timed_out[0] != 0.f ? __builtin_nanf("") : 0.f;       // (written before the first barrier above)
            if (prm.saved_mean) prm.saved_mean[ps] = omp + poison;
            if (ps < 2) prm.loss_means[ps] = (lsum - omp * csum) * inv_cnt + poison;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (tid == 0) {                          // torch.cat(negative losses).mean() (:390, train_segmentation.py:176), pair-set order
            float nsum = 0.f;
            for (int pp = 2; pp < prm.n_sets; ++pp) nsum += som[prm.n_sets + pp];
            prm.loss_means[2] = (prm.n_neg > 0 ? nsum * inv_cnt / (float)prm.n_neg : 0.f) + (timed_out[0] != 0.f ? __builtin_nanf("") : 0.f);
        }
        if (ts && tid == 0) ts[14] = __builtin_amdgcn_s_memrealtime();
    }
    if (repair) {                                // (workgroup-uniform; never taken in normal operation)
        __syncthreads();                         // old means of wave 0
        for (int t = 2 * B; t < n_tiles; ++t) {
            if (sst[t * 4 + 3] != 0.f) continue;                   // (workgroup-uniform)
            const float omp = som[t / B];
            float* lossr = prm.neg_loss + (size_t)(t - 2 * B) * P2;
            const float* cdr = prm.neg_cd + (size_t)(t - 2 * B) * P2;
            for (int e = tid; e < P2; e += FUSED_THREADS) {
                const float cdv = __hip_atomic_load(cdr + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float lpv = __hip_atomic_load(lossr + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float cl = fminf(fmaxf(cdv, cmin), cmax);
                lossr[e] = __builtin_fmaf(-omp, cl, lpv);
            }
        }
    }
    if (tid == 64) __hip_atomic_store(prm.done_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ts && tid == 0) ts[15] = __builtin_amdgcn_s_memrealtime();
}

// ------------------------------------------------------------------------------------------ the kernel
template <int PREC, int NJ, int NKCT, bool ODDK>
__global__ void __launch_bounds__(FUSED_THREADS) corr_fused_kernel(const FusedParams prm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* rowmean = reinterpret_cast<float*>(smem + RD_ROWMEAN);
    float* red = reinterpret_cast<float*>(smem + RD_RED);
    float* csc = reinterpret_cast<float*>(smem + RD_CSC);
    float* cscc = reinterpret_cast<float*>(smem + RD_CSCC);
    int4* tapof = reinterpret_cast<int4*>(smem + RD_TAPOF);
    int4* tapoc = reinterpret_cast<int4*>(smem + RD_TAPOC);
    float4* tapw = reinterpret_cast<float4*>(smem + RD_TAPW);
    unsigned char* ring = smem + RD_RING;
    float* Tfd = reinterpret_cast<float*>(smem + RD_RING);   // epilogue alias of the ring
    float* Tcd = Tfd + TP * LDT;
    typedef P1Layout<NJ, PREC> LY;
    constexpr int NCH2 = LY::NCH2;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool mfma_team = wave8 < 4;            // wave-uniform role
    const int wave = wave8 & 3;
    const int gt = mfma_team ? tid : tid - NTHREADS;   // index inside the team
    const int wr = wave >> 1, wc = wave & 1;
    const int B = prm.B, P = prm.P;
    constexpr int NKC = NKCT;                    // code K-chunks (compile time: the head of the stream is fully static)
    const int kper = prm.kper;
    constexpr int NT = NKC + NCH2;               // stages of the stream: the code K-chunks, then the feature stages
    const int me = blockIdx.x;
    const int n_tiles = prm.n_sets * B;

    unsigned long long* ts = reinterpret_cast<unsigned long long*>(prm.stats + (size_t)n_tiles * 4 + 256) + (size_t)me * 16;
    const bool helper = me >= n_tiles;           // no tile: phase 1 only (the launch covers every CU)
    const bool stamp_on = (prm.debug & 256) && tid == 0 && !helper;
    if (stamp_on) ts[0] = __builtin_amdgcn_s_memrealtime();
#ifdef STEGO_FUSED_TIMELINE
    // (tools/timeline_fused.py builds with -DSTEGO_FUSED_TIMELINE) per-stage timeline of wave 0 (MFMA team) and wave 4
    // (gather team), 4 stamps per stage each; enabled at run time by debug bit 512
    unsigned long long* tl = reinterpret_cast<unsigned long long*>(prm.stats + (size_t)n_tiles * 4 + 256) + (size_t)n_tiles * 16 + (size_t)me * 128 + (mfma_team ? 0 : 64);
    const bool tl_on = (prm.debug & 512) && (tid == 0 || tid == 256) && !helper;
#define TL(n, k) do { if (tl_on && (n) < 16) tl[(n) * 4 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TL(n, k) do { } while (0)
#endif

    const __amdgpu_buffer_rsrc_t fs_rsrc = __builtin_amdgcn_make_buffer_rsrc(prm.fs, 0, prm.fs_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t csf_rsrc = __builtin_amdgcn_make_buffer_rsrc(prm.csf, 0, prm.csf_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t cs_rsrc = __builtin_amdgcn_make_buffer_rsrc(prm.cs, 0, prm.cs_bytes, 0x00020000);

    // ---- hand-off words inside the workgroup (LDS comes up with whatever the previous workgroup left)
    int* tile_slot = reinterpret_cast<int*>(red + 56);
    unsigned* team_cnt = reinterpret_cast<unsigned*>(red + 57);
    float* fin = red + 48;                       // [0] 1 = I am the last workgroup
    if (tid == 0) { tile_slot[0] = -1; team_cnt[0] = 0u; fin[1] = 0.f; }       // fin[1]: the tail saw a hand-off word time out
    __syncthreads();

    // the gather waves that wait for the tile (5 .. 11) pull coords2 into this CU's L1 meanwhile: the tap tables are then built from
    // L1 hits instead of a cold round trip (the gather stream - the critical path - starts that much earlier: -1.0 us same box, r04d)
    if (!helper && wave8 >= 5) {
        const int idx = 32 * ((wave8 - 5) * 64 + lane);
        if (idx < B * P * 2) { const float x = prm.coords2[idx]; asm volatile("" :: "v"(x)); }
    }
    // ---- phase 1 (the MFMA team, 4 waves): my share of the anchor sets of my XCD, 2 G points per wave and pass.  The team
    // syncs through an LDS counter, not s_barrier: the gather team is not part of it - it works out the tile, builds its tap
    // table and fills the B sides of the first four ring slots meanwhile (none of which needs an anchor).
    const bool p1_here = me < prm.n_owner && !(prm.debug & 64);       // (debug 64: nobody samples, every tile takes the give-up path)
    // The share is 16 rows when every CU has a workgroup; one workgroup per tile (STEGO_SHARED_DEVICE) makes it 18-19, and a second
    // pass for two or three rows would repeat the whole latency chain (+3 us): gather waves 8 and 9 - idle until the tile is known -
    // take two overflow rows each in the first pass, stage them like everybody else and ARRIVE at the team's first barrier (they
    // do not wait: copying out and publishing stay with the MFMA team).
    const int p1x = me & 7, p1r = me >> 3;
    const int p1nb = p1x < B ? (B - p1x + 7) >> 3 : 0;
    const int p1nslot = (prm.n_owner - p1x + 7) >> 3;
    const long long p1L = (long long)p1nb * TP;
    int p1beg = (int)(p1L * p1r / p1nslot), p1end = (int)(p1L * (p1r + 1) / p1nslot);
    // Round 4: who carries phase 1.  The workgroups whose gather team streams a B side finish their share 4 (median) to 9 us (slowest)
    // later than the ones without a gather stream (stamps by slot class, profiles/r04f): their 16 points' taps and write-through
    // stores queue behind the gather head in the CU's own memory pipeline, and every anchor waits for its slowest contributor.  With
    // prm.p1_light (one round, B a multiple of 8, a workgroup on every CU) the LIGHT workgroups of an XCD - the self-correlation
    // tiles, which the placement puts on its slots 0 .. B/8 - 1 (they come first in tile order and prefer the XCD of their own image),
    // and the tile-less helpers behind the tiles - sample with all twelve waves, 24 G rows per workgroup in one pass; what they do
    // not cover is spread over the others (5-6 rows at B = 32, nothing at B = 16).
    typedef P1Layout<NJ, PREC, true> LYL;
    bool p1_is_light = false;
    if (prm.p1_light) {
        const int tile_slots = (n_tiles - p1x + 7) >> 3;                 // slots of this XCD that hold a tile
        const int n_light = p1nb + (p1nslot - tile_slots), n_heavy = p1nslot - n_light;
        const int rowsL = (prm.debug & 2048) ? 12 * 2 * LYL::G : LYL::ROWS;       // (debug 2048: no second chunk - the gathered tiles keep a few rows)
        const int R = (int)p1L, capL = n_light * rowsL;
        p1_is_light = p1r < p1nb || p1r >= tile_slots;
        const int lr = p1r < p1nb ? p1r : p1nb + (p1r - tile_slots);     // rank among the light / the other workgroups of the XCD
        const int hr = p1r - p1nb;
        if (R <= capL) {
            p1beg = p1_is_light ? R * lr / n_light : 0;
            p1end = p1_is_light ? R * (lr + 1) / n_light : 0;
        } else if (p1_is_light) {
            p1beg = rowsL * lr;
            p1end = p1beg + rowsL;
        } else {
            p1beg = capL + (int)((long long)(R - capL) * hr / n_heavy);
            p1end = capL + (int)((long long)(R - capL) * (hr + 1) / n_heavy);
        }
    }
    if (p1_here && p1_is_light) {
        // ---- a light workgroup: every wave samples 2 G rows (the gather waves 2 more), all twelve copy out, one lane publishes
        const int lr0 = mfma_team ? 2 * LYL::G * wave : LYL::MROWS + 2 * LYL::G * (wave8 - 4);
        __builtin_amdgcn_s_setprio(3);
        if (p1beg + lr0 < p1end)
            p1_sample_rows<NJ, PREC, NKCT, LYL::G, true, ODDK>(prm, p1x, p1beg, p1end, lr0, lane, ring, (stamp_on && mfma_team) ? ts + 8 : nullptr);
        if constexpr (LYL::XB > 0) {
            const int lr1 = LYL::MROWS + 8 * 2 * LYL::G + 2 * (wave8 - 4);
            if (!mfma_team && p1beg + lr1 < p1end) p1_sample_rows<NJ, PREC, NKCT, 1, true, ODDK>(prm, p1x, p1beg, p1end, lr1, lane, ring, nullptr);
        }
        const int nrows = p1end - p1beg;
        team_barrier(team_cnt, FUSED_WAVES, lane);
        if (nrows > 0) {
            const int to_edge = (((p1beg >> 7) + 1) << 7) - p1beg;
            const int n0 = min(nrows, to_edge);
            p1_copy_out<NJ, PREC, true>(prm, p1x, p1beg, 0, n0, wave8, FUSED_WAVES, lane, ring, fs_rsrc, csf_rsrc, cs_rsrc);
            if (n0 < nrows) p1_copy_out<NJ, PREC, true>(prm, p1x, p1beg, n0, nrows - n0, wave8, FUSED_WAVES, lane, ring, fs_rsrc, csf_rsrc, cs_rsrc);
        }
        if (stamp_on) ts[10] = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this wave's write-through stores have landed
        team_barrier(team_cnt, 2 * FUSED_WAVES, lane);
        if (tid == 0 && nrows > 0) p1_publish(prm, p1x, p1beg, nrows);
        __builtin_amdgcn_s_setprio(0);
    } else if (mfma_team && p1_here) {
        const int x = p1x, beg = p1beg, end = p1end;
        unsigned epoch = 0;
        // the team shares its SIMDs with two gather waves each, and everybody's anchors wait for it: it goes first
        __builtin_amdgcn_s_setprio(3);
        bool first = true;
        for (int blk0 = beg; blk0 < end; first = false) {
            const int cap = first ? LY::ROWS : LY::MROWS;
            const int nrows = min(cap, end - blk0);
            const int n_extra = nrows > LY::MROWS ? (nrows - LY::MROWS + 1) >> 1 : 0;          // gather waves arriving at this barrier
            if (2 * LY::G * wave < min(nrows, LY::MROWS))       // (a wave without rows in this pass goes straight to the barrier)
                p1_sample_rows<NJ, PREC, NKCT, LY::G, false, ODDK>(prm, x, blk0, min(end, blk0 + LY::MROWS), 2 * LY::G * wave, lane, ring, stamp_on ? ts + 8 : nullptr);
            if (stamp_on && (prm.debug & 1024)) ts[12] = __builtin_amdgcn_s_memrealtime();
            epoch += 4 + n_extra;
            team_barrier(team_cnt, epoch, lane);
            if (stamp_on && (prm.debug & 1024)) ts[13] = __builtin_amdgcn_s_memrealtime();
            // a run must stay inside one anchor: split the pass at an anchor boundary
            const int to_edge = (((blk0 >> 7) + 1) << 7) - blk0;
            const int n0 = min(nrows, to_edge);
            p1_copy_out<NJ, PREC>(prm, x, blk0, 0, n0, wave, 4, lane, ring, fs_rsrc, csf_rsrc, cs_rsrc);
            if (n0 < nrows) p1_copy_out<NJ, PREC>(prm, x, blk0, n0, nrows - n0, wave, 4, lane, ring, fs_rsrc, csf_rsrc, cs_rsrc);
            if (stamp_on) ts[10] = __builtin_amdgcn_s_memrealtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's write-through stores have landed
            if (stamp_on && (prm.debug & 1024)) ts[14] = __builtin_amdgcn_s_memrealtime();
            epoch += 4;
            team_barrier(team_cnt, epoch, lane);
            if (tid == 0) p1_publish(prm, x, blk0, nrows);
            blk0 += cap;
        }
        __builtin_amdgcn_s_setprio(0);
    } else if (p1_here && LY::XW > 0 && wave8 >= 8 && wave8 < 8 + LY::XW) {
        const int lr0 = LY::MROWS + 2 * (wave8 - 8);
        if (p1beg + lr0 < p1end) {
            p1_sample_rows<NJ, PREC, NKCT, 1, false, ODDK>(prm, p1x, p1beg, p1end, lr0, lane, ring, nullptr);
            team_arrive(team_cnt, lane);
        }
    }
    if (stamp_on) ts[1] = __builtin_amdgcn_s_memrealtime();
    if (helper) {
        // a workgroup without a tile (the CUs the tiles leave free): phase 1 was all; it still takes a ticket, because the
        // LAST ticket is what says that nobody will touch the hand-off words of this launch any more
        if (tid == 0) {
            const unsigned t = __hip_atomic_fetch_add(prm.done_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            fin[0] = t == gridDim.x - 1 ? 1.f : 0.f;
        }
        __syncthreads();
        if (fin[0] != 0.f) last_workgroup_tail(prm, Tfd, fin + 1, tid, n_tiles, nullptr);
        return;
    }

    // ---- which tile am I: the first gather wave works it out (ballots over perms, ~1.3 us), everybody else picks it up from LDS
    int tile;
    // Round 6: the first 2 nb slots of an XCD hold, by construction of the placement (tile order, preferences b % 8), the self-correlation
    // and the inter tiles of its nb anchors: those workgroups know their tile from their index - no cold look at perms, no ballots.  It
    // matters for the INTER tiles: the only readers of their image, every line a cold miss, the last tiles to finish in the skeleton
    // (profiles/r06_skeleton_B32_C384.txt) - their gather stream starts ~3 us earlier.
    const int k_slot = me >> 3;
    const int k_nb = (me & 7) < B ? (B - (me & 7) + 7) >> 3 : 0;
    const bool known = prm.ps_round * B >= n_tiles && k_slot < 2 * k_nb && !(prm.debug & 16);      // (debug 16: everybody looks it up)
    if (known) {
        tile = (k_slot < k_nb ? 0 : B) + (me & 7) + 8 * (k_slot < k_nb ? k_slot : k_slot - k_nb);
        if ((prm.debug & 256) && tid == 256) ts[6] = __builtin_amdgcn_s_memrealtime();
    } else if (wave8 == 4) {
        // More tiles than compute units (B = 64 with 5 negatives: 448): ROUNDS of whole pair-sets - the old_mean rendezvous of a
        // pair-set is between workgroups that run at the same time - of ps_round * B tiles each; the workgroups of round r are
        // the r-th window of the grid and start as the compute units of earlier workgroups become free (every workgroup needs a whole
        // one: 136 KB of LDS).  Should the hardware start a later window first, its tiles' bounded spins fall back on the repair
        // path: slower, never wrong.  A later round finds its anchors ready (phase 1 belongs to the first n_owner workgroups).
        const int per_round = prm.ps_round * B;
        int pref0[ASSIGN_NB];
        if (per_round >= n_tiles) {              // one round (workgroup-uniform)
            assign_prefetch<false>(prm, lane, 0, n_tiles, pref0);
            tile = assign_tile<false>(prm, me, lane, 0, n_tiles, pref0);
        } else {
            const int w0 = (me / per_round) * per_round;
            const int ntr = n_tiles - w0 < per_round ? n_tiles - w0 : per_round;
            assign_prefetch<true>(prm, lane, w0, ntr, pref0);
            tile = assign_tile<true>(prm, me, lane, w0, ntr, pref0);
        }
        if (lane == 0) __hip_atomic_store(tile_slot, tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((prm.debug & 256) && lane == 0) ts[6] = __builtin_amdgcn_s_memrealtime();
    } else {
        for (;;) {
            tile = __hip_atomic_load(tile_slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (tile >= 0) break;
            __builtin_amdgcn_s_sleep(2);
        }
        tile = __builtin_amdgcn_readfirstlane(tile);
    }
    const int b = tile % B, p = tile / B;
    const bool sameAB = p == 0;
    const int sA = b;
    const int sB = p == 0 ? b : p * B + b;
    const bool usePos = p == 1;
    int src = b;
    if (p >= 2) src = (int)prm.perms[(size_t)(p - 2) * B + b];
    src = __builtin_amdgcn_readfirstlane(src);
    const unsigned char* fsA = prm.fs + (size_t)sA * NCH2 * RS_SIDE;
    const unsigned char* csfA = prm.csf + (size_t)sA * NKC * RS_SIDE;
    const MapV mfB = usePos ? prm.feats_pos : prm.feats;
    const MapV mcB = usePos ? prm.code_pos : prm.code;
    const float* imgB = mfB.p + (long long)src * mfB.sn;
    const float* cimgB = mcB.p + (long long)src * mcB.sn;
    const float* coordsB = prm.coords2 + (size_t)b * P * 2;
    const bool gatherB = !sameAB;

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
    f32x16 accf[2][2], accc[2][2];
    if (mfma_team) {
        // ================================================================= MFMA team
        // A side of stage n = code K-chunk n (n < NKC) or feature stage n - NKC of my anchor: 16 pieces of 1 KiB
        auto stage_src = [&](int n) { return n < NKC ? csfA + (size_t)n * RS_SIDE : fsA + (size_t)(n - NKC) * RS_SIDE; };
        const unsigned ring_addr = __builtin_amdgcn_readfirstlane(lds_address(ring));
        // Wait for my anchor set (published by its phase-1 owners).  ONE wave per workgroup polls (a relaxed load, a long
        // sleep): 900 waves polling 32 words every 0.25 us collapsed the channels holding them and delayed the very
        // stores and counter updates they were waiting for (phase 1 took 17-33 us instead of 9).  Wave 0 then issues the
        // first three stages alone; the other waves only need the anchor after the barrier wave 0 arrives at.
        if (wave == 0) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            bool ready = false;
            for (;;) {
                const unsigned c = __hip_atomic_load(prm.anchor_cnt + (size_t)sA * ANCHOR_CNT_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__builtin_amdgcn_readfirstlane(c) >= (unsigned)TP) { ready = true; break; }
                if ((long long)(__builtin_amdgcn_s_memrealtime() - t0) > ((prm.debug & 64) ? 100 : prm.timeout_ticks)) break;
                __builtin_amdgcn_s_sleep(10);
            }
            if (!ready) {
                // (counted: a host reads the two event words through stego_corr_event_counters - a launch that keeps taking this path
                // on a shared or partitioned device is ~30 us slower and says so nowhere else)
                if (lane == 0) __hip_atomic_fetch_add(prm.done_cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // The owners did not show up in time (not co-resident: a shared or over-subscribed device).  Sample the
                // whole anchor here, this wave alone (staging in the A sides, which nothing touches before my own LDS-DMA below):
                // identical inputs give identical bytes, so racing with a late owner is benign; nothing else in the
                // launch depends on the counter being exact.
                for (int q0 = 0; q0 < TP; q0 += 2 * LY::G) {
                    p1_sample_rows<NJ, PREC, NKCT, LY::G, false, ODDK>(prm, sA, q0, TP, 0, lane, ring, nullptr);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    p1_copy_out<NJ, PREC>(prm, sA, q0, 0, 2 * LY::G, 0, 1, lane, ring, fs_rsrc, csf_rsrc, cs_rsrc);
                }
            }
            if (stamp_on) ts[2] = __builtin_amdgcn_s_memrealtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // nothing of mine in flight: the counted waits below start from zero
#pragma unroll
            for (int n = 0; n < 3; ++n)
                for (int pc = 0; pc < 16; ++pc)
                    dma_piece_sc1(stage_src(n) + pc * 1024 + lane * 16, ring_addr + n * RS_STAGE + pc * 1024);
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // head of stage n: my pieces of stage n have landed (in-order completion: only what I issued after them may still
        // be outstanding), barrier B(n), then the A side of stage n + 3 goes into the slot stage n - 1 just left
        auto stage_head = [&](int n) {
            TL(n, 0);
            if (wave == 0 && n == 0) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            else if (wave == 0 && n == 1) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
            else if (n + 2 < NT) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (n + 1 < NT) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            TL(n, 1);
            ring_barrier();                          // B(n): stage n complete in LDS; everyone is done with stage n - 1
            TL(n, 2);
            if (n + 3 < NT) {
                const unsigned char* s3 = stage_src(n + 3);
                const unsigned dst = ring_addr + ((n + 3) & (RS_NS - 1)) * RS_STAGE;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int pc = wave + 4 * i;
                    dma_piece_sc1(s3 + pc * 1024 + lane * 16, dst + pc * 1024);
                }
            }
        };
        // the MFMA wave goes ahead of the two gather waves on its SIMD: it also issues the A side's LDS-DMA, and a stage it finishes late is
        // a late barrier for everybody (measured same-box: -0.4 us per step; the reverse, the gather team first, +1.2 us)
        __builtin_amdgcn_s_setprio(2);
        // the code K-chunks first (exact fp32 MFMAs, VALU rate: they run while the ring fills with feature stages) ...
        zero_acc(accc);
#pragma unroll
        for (int n = 0; n < NKC; ++n) {
            stage_head(n);
            const unsigned char* As = ring + (n & (RS_NS - 1)) * RS_STAGE;
            if constexpr (PREC == PREC_F32) mma_stage_f(As, sameAB ? As : As + RS_SIDE, kper, accc, lane, wr, wc);
            else mma_stage_fh(As, sameAB ? As : As + RS_SIDE, kper, accc, lane, wr, wc);
            TL(n, 3);
        }
        // ... then the feature stages
        zero_acc(accf);
#pragma unroll 1
        for (int n = NKC; n < NT; ++n) {
            stage_head(n);
            const unsigned char* As = ring + (n & (RS_NS - 1)) * RS_STAGE;
            const unsigned char* Bs = sameAB ? As : As + RS_SIDE;
            if constexpr (PREC == PREC_F32) mma_stage_f(As, Bs, KC2, accf, lane, wr, wc);
            else mma_stage_h(As, Bs, accf, lane, wr, wc);
            TL(n, 3);
        }
        __builtin_amdgcn_s_setprio(0);
    } else if (!gatherB) {
        // ================================================================= gather team of a self-correlation tile (B = A)
        for (int n = 0; n < NT; ++n) ring_barrier();
        if (gt < TP) { csc[gt] = 1.f; cscc[gt] = 1.f; }
    } else {
        // ================================================================= gather team
        const int g8 = gt & 7, prow = gt >> 3;
        const int gwave = wave8 - 4;
        float ss[GI], bsc[GI], ssc[GI], bscc[GI];
#pragma unroll
        for (int j = 0; j < GI; ++j) { ss[j] = 0.f; bsc[j] = 0.f; ssc[j] = 0.f; bscc[j] = 0.f; }
        if (lane < 8 * GI) {
            // tap table of the B points: every wave computes the 8 GI entries it reads itself (no barrier needed)
            const int q = GP * (lane >> 3) + 8 * gwave + (lane & 7);
            const f32x2 cxy = *reinterpret_cast<const f32x2*>(coordsB + coord_index(prm, q));
            int4 yx;
            float4 w;
            point_taps(prm, cxy, q, yx, w);
            tapof[q] = taps_to_offsets(yx, mfB.sh, mfB.sw);
            tapoc[q] = yx;                                          // (y << 16 | x) of the four taps: code offsets are rebuilt from it
            tapw[q] = w;
            prm.tapyx[(size_t)sB * TP + q] = yx;                    // saved context of the backward
            prm.tapw[(size_t)sB * TP + q] = w;
        }
        // The taps of my GI points stay in REGISTERS for the whole stream (32-bit byte offsets from a wave-uniform base,
        // and the bilinear weights): the stream itself then contains no LDS read at all, only gathers, VALU and LDS
        // stores.  (Re-reading the table every stage cost issue slots, and the build that did so - LDS read results and
        // gather results sharing registers - produced rare wrong samples in its last lanes; see DESIGN.md.)
        unsigned fo[GI][4], pk[GI];
        float4 tw[GI];
#pragma unroll
        for (int j = 0; j < GI; ++j) {
            const int4 o = tapof[GP * j + prow];
            const int4 c = tapoc[GP * j + prow];
            tw[j] = tapw[GP * j + prow];
            fo[j][0] = (unsigned)(o.x + 4 * g8) * 4u; fo[j][1] = (unsigned)(o.y + 4 * g8) * 4u;
            fo[j][2] = (unsigned)(o.z + 4 * g8) * 4u; fo[j][3] = (unsigned)(o.w + 4 * g8) * 4u;
            // y0 | x0 << 8 | y1 << 16 | x1 << 24 (maps up to 256 x 256: fused_supported): the code stages are few, their
            // offsets are rebuilt per use instead of holding 16 more registers
            pk[j] = (unsigned)(c.x >> 16) | ((unsigned)(c.x & 0xff) << 8) | ((unsigned)(c.w >> 16) << 16) | ((unsigned)(c.w & 0xff) << 24);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // Issue the gathers of a stage into a register set: branch-free, a fixed number of loads, so that the compiler's
        // counted vmcnt waits stay exact (a conditional issue made every wait a near-drain).
        // Code K-chunk m: this lane's 4 channels as two 8-byte halves (a pixel is only 8-byte aligned when K % 4 == 2, 4-byte
        // aligned when K is odd); channels beyond K re-read channel 0 and are zeroed.
        auto issue_code = [&](GSet& g, int m) {
            const int k = m * kper + 4 * g8;
            const bool in = 4 * g8 < kper;
            const char* cb = reinterpret_cast<const char*>(cimgB);                  // wave-uniform base + 32-bit lane offsets
            if constexpr (!ODDK) {
                const bool v0 = in && k + 1 < prm.K, v1 = in && k + 3 < prm.K;
                const unsigned k0 = v0 ? 4u * k : 0u, k1 = v1 ? 4u * (k + 2) : 0u;  // byte offsets inside a pixel
#pragma unroll
                for (int j = 0; j < GI; ++j) {
                    const unsigned y0 = (pk[j] & 0xff) * (unsigned)mcB.sh, x0 = ((pk[j] >> 8) & 0xff) * (unsigned)mcB.sw;
                    const unsigned y1 = ((pk[j] >> 16) & 0xff) * (unsigned)mcB.sh, x1 = (pk[j] >> 24) * (unsigned)mcB.sw;
                    const unsigned co[4] = {(y0 + x0) * 4u, (y0 + x1) * 4u, (y1 + x0) * 4u, (y1 + x1) * 4u};
#pragma unroll
                    for (int tq = 0; tq < 4; ++tq) {
                        const f32x2 lo = *reinterpret_cast<const f32x2*>(cb + (co[tq] + k0));
                        const f32x2 hi = *reinterpret_cast<const f32x2*>(cb + (co[tq] + k1));
                        g.tv[j][tq] = f32x4{v0 ? lo[0] : 0.f, v0 ? lo[1] : 0.f, v1 ? hi[0] : 0.f, v1 ? hi[1] : 0.f};
                    }
                }
                return;
            }
            // odd K (its own instantiations: the masks below cost the even-K kernels registers they do not have): element validity; the pair that straddles the end of a pixel is fetched one channel lower and shifted (see phase 1)
            const bool e0 = in && k < prm.K, e1 = in && k + 1 < prm.K, e2 = in && k + 2 < prm.K, e3 = in && k + 3 < prm.K;
            const bool s0 = e0 && !e1, s1 = e2 && !e3;
            const unsigned k0 = e0 ? 4u * (unsigned)(k - (s0 ? 1 : 0)) : 0u, k1 = e2 ? 4u * (unsigned)(k + 2 - (s1 ? 1 : 0)) : 0u;
#pragma unroll
            for (int j = 0; j < GI; ++j) {
                const unsigned y0 = (pk[j] & 0xff) * (unsigned)mcB.sh, x0 = ((pk[j] >> 8) & 0xff) * (unsigned)mcB.sw;
                const unsigned y1 = ((pk[j] >> 16) & 0xff) * (unsigned)mcB.sh, x1 = (pk[j] >> 24) * (unsigned)mcB.sw;
                const unsigned co[4] = {(y0 + x0) * 4u, (y0 + x1) * 4u, (y1 + x0) * 4u, (y1 + x1) * 4u};
#pragma unroll
                for (int tq = 0; tq < 4; ++tq) {
                    const f32x2 lo = *reinterpret_cast<const f32x2_u*>(cb + (co[tq] + k0));
                    const f32x2 hi = *reinterpret_cast<const f32x2_u*>(cb + (co[tq] + k1));
                    g.tv[j][tq] = f32x4{e0 ? (s0 ? lo[1] : lo[0]) : 0.f, e1 ? lo[1] : 0.f, e2 ? (s1 ? hi[1] : hi[0]) : 0.f, e3 ? hi[1] : 0.f};
                }
            }
        };
        // Buffer loads (round 5): resource in scalar registers, the stage as the scalar offset, ONE 32-bit lane offset per tap.  With plain
        // pointers the compiler is free to turn the eight taps into eight 64-bit lane pointers - 16 registers - and did so as soon as the
        // stream was rearranged, spilling them around the unrolled ends of the stream (a scratch reload there is a vmcnt(0) drain of the
        // whole gather stream: 51 -> 67 us, profiles/r05d_experiments.txt); as the kernel stands: 24 -> 20 spilled registers, same time.
        const __amdgpu_buffer_rsrc_t imgB_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(imgB), 0, 0x7fffffff, 0x00020000);
        auto issue_feat = [&](GSet& g, int f) {
            const int so = __builtin_amdgcn_readfirstlane(f * KC2 * 4);
#pragma unroll
            for (int j = 0; j < GI; ++j)
#pragma unroll
                for (int tq = 0; tq < 4; ++tq)
                    g.tv[j][tq] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(imgB_rsrc, (int)fo[j][tq], so, 0));
        };
        auto blend = [&](const GSet& g, int j, float (&v)[4]) {
            const float4 w = tw[j];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = w.x * g.tv[j][0][e] + w.y * g.tv[j][1][e] + w.z * g.tv[j][2][e] + w.w * g.tv[j][3][e];
        };
        // Commit FEATURE stage m (m >= NKC) to the B side of its ring slot (raw samples; 1/||b|| becomes a column scale).
        // decide: the point's power-of-two prescale is chosen from the first feature stage in which it is non-zero
        // among the first two (|x| * bsc in [0.5, 1)); later stages are branch-free (F.normalize is scale invariant, the
        // column scale divides the prescale out again; fp16 leaves 2^15 of headroom for channels beyond those stages).
        auto commit_feat = [&](const GSet& g, int m, bool decide) {
            unsigned char* dst = ring + (m & (RS_NS - 1)) * RS_STAGE + RS_SIDE;
#pragma unroll
            for (int j = 0; j < GI; ++j) {
                const int q = GP * j + prow;
                float v[4];
                blend(g, j, v);
                ss[j] += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
                if constexpr (PREC == PREC_F32) {
                    *reinterpret_cast<f32x4*>(dst + swz_f(q, g8)) = f32x4{v[0], v[1], v[2], v[3]};
                } else {
                    if (decide && bsc[j] == 0.f) {
                        float mx = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
#pragma unroll
                        for (int mm = 4; mm >= 1; mm >>= 1) mx = fmaxf(mx, __shfl_xor(mx, mm, 64));
                        if (mx > 0.f) bsc[j] = __builtin_ldexpf(1.f, -__builtin_amdgcn_frexp_expf(mx));
                    }
                    const float sc = bsc[j] == 0.f ? 1.f : bsc[j];
                    unsigned h0, l0, h1, l1;
                    split_f16_pair(v[0] * sc, v[1] * sc, h0, l0);
                    split_f16_pair(v[2] * sc, v[3] * sc, h1, l1);
                    // lane pairs (g8 even / odd) own one 16-byte unit: the even lane stores its hi halves + its neighbour's (hi plane),
                    // the odd lane the two lo halves (lo plane): one ds_write_b128 per lane instead of two ds_write_b64
                    const bool odd = g8 & 1;
                    const unsigned s0 = odd ? h0 : l0, s1 = odd ? h1 : l1;          // what my neighbour needs from me
                    const unsigned r0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s0, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
                    const unsigned r1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s1, 0xB1, 0xF, 0xF, true);
                    const u32x4 d = odd ? u32x4{r0, r1, l0, l1} : u32x4{h0, h1, r0, r1};
                    *reinterpret_cast<u32x4*>(dst + (odd ? 8192 : 0) + swz_h(q, g8 >> 1)) = d;
                }
            }
        };
        // commit a CODE stage m: raw fp32 samples to the B side of its ring slot
        // F16X3: the MFMA team splits these operands into fp16 halves, so the raw samples get a power-of-two prescale per point,
        // chosen from the first code stage in which the point is non-zero among the first two (as for the features)
        auto commit_code = [&](const GSet& g, int m) {
            unsigned char* dst = ring + (m & (RS_NS - 1)) * RS_STAGE + RS_SIDE;
#pragma unroll
            for (int j = 0; j < GI; ++j) {
                const int q = GP * j + prow;
                float v[4];
                blend(g, j, v);
                ssc[j] += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
                float sc = 1.f;
                if constexpr (PREC == PREC_F16X3) {
                    if (m < 2 && bscc[j] == 0.f) {
                        float mx = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
#pragma unroll
                        for (int mm = 4; mm >= 1; mm >>= 1) mx = fmaxf(mx, __shfl_xor(mx, mm, 64));
                        if (mx > 0.f) bscc[j] = __builtin_ldexpf(1.f, -__builtin_amdgcn_frexp_expf(mx));
                    }
                    sc = bscc[j] == 0.f ? 1.f : bscc[j];
                }
                *reinterpret_cast<f32x4*>(dst + swz_f(q, g8)) = f32x4{v[0] * sc, v[1] * sc, v[2] * sc, v[3] * sc};
            }
        };
        // after the last code stage: norms -> column scale of cd; normalised rows + norms -> the backward's context.
        // The raw samples are read back from the ring (this lane wrote them; their slots are not reused before B(2)).
        auto finish_codes = [&]() {
#pragma unroll
            for (int j = 0; j < GI; ++j) {
                const int q = GP * j + prow;
                float sq = ssc[j];
#pragma unroll
                for (int mm = 4; mm >= 1; mm >>= 1) sq += __shfl_xor(sq, mm, 64);
                const bool valid = q < P;
                const float nr = valid ? sqrtf(sq) : 0.f;
                const float inv = valid ? __builtin_amdgcn_rcpf(fmaxf(nr, 1e-10f)) : 0.f;
                const float unsc = (PREC == PREC_F16X3 && bscc[j] != 0.f) ? 1.f / bscc[j] : 1.f;     // (a power of two: exact)
                if (g8 == 0) { cscc[q] = inv * unsc; prm.nrm[(size_t)sB * TP + q] = nr; }
                float* crow = prm.cs + ((size_t)sB * TP + q) * prm.LDK;
#pragma unroll
                for (int m = 0; m < NKC; ++m) {
                    const int k = m * kper + 4 * g8;
                    if (4 * g8 < kper && k < prm.KQ)
                        *reinterpret_cast<f32x4*>(crow + k) = *reinterpret_cast<const f32x4*>(ring + m * RS_STAGE + RS_SIDE + swz_f(q, g8)) * (inv * unsc);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        GSet ga, gb;
        // stage m of the stream (compile-time m after unrolling): a code K-chunk or a feature stage
        auto issue = [&](GSet& g, int m) {
            if (m < NKC) issue_code(g, m);
            else issue_feat(g, m - NKC);
        };
        auto commit = [&](GSet& g, int m) {
            if (m < NKC) { commit_code(g, m); if (m == NKC - 1) finish_codes(); }
            else commit_feat(g, m, m - NKC < 2);
        };
        // Head (static): the B sides of stages 0..3 fill the four (still free) ring slots before the first barrier, out
        // of ONE round trip: three stages in flight at once (a third register set that only lives here), the fourth
        // behind the first commit.  None of this depends on the anchor, so it runs while the MFMA team still waits for it.
        static_assert(NT >= 8, "the static head covers the code chunks and the first feature stages");
        const bool gstamp = (prm.debug & 256) && tid == NTHREADS;
        if (gstamp) ts[15] = __builtin_amdgcn_s_memrealtime();
        {
            GSet gc;
            issue(ga, 0);
            issue(gb, 1);
            issue(gc, 2);
            commit(ga, 0);
            issue(ga, 3);
            commit(gb, 1);
            commit(gc, 2);
            commit(ga, 3);
        }
        if (gstamp) ts[11] = __builtin_amdgcn_s_memrealtime();
        // after B(n): commit stage n + 2 (gathered two barriers ago; the head did it for n < 2), then re-issue its registers
        // for stage n + 4 (interleaving the two item by item was measured slower).  Stages alternate between the sets.
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            TL(n, 0);
            ring_barrier();                          // B(n)
            TL(n, 1);
            if (n & 1) { if (n >= 2) commit(gb, n + 2); TL(n, 2); issue(gb, n + 4); }
            else { if (n >= 2) commit(ga, n + 2); TL(n, 2); issue(ga, n + 4); }
            TL(n, 3);
        }
        constexpr int IT = (NT - 8) / 2;             // iterations in which both halves commit and issue
        constexpr int TAIL0 = 4 + 2 * IT;
#pragma unroll 1
        for (int n = 4; n < TAIL0; n += 2) {
            TL(n, 0);
            ring_barrier();                          // B(n)
            TL(n, 1);
            commit_feat(ga, n + 2, false);
            TL(n, 2);
            __builtin_amdgcn_sched_barrier(0);
            issue_feat(ga, n + 4 - NKC);
            TL(n, 3);
            TL(n + 1, 0);
            ring_barrier();                          // B(n + 1)
            TL(n + 1, 1);
            commit_feat(gb, n + 3, false);
            TL(n + 1, 2);
            __builtin_amdgcn_sched_barrier(0);
            issue_feat(gb, n + 5 - NKC);
            TL(n + 1, 3);
        }
#pragma unroll
        for (int n = TAIL0; n < NT; ++n) {          // the last stages: nothing left to issue, then nothing left to commit
            ring_barrier();                          // B(n)
            if (n + 2 < NT) { if (n & 1) commit_feat(gb, n + 2, false); else commit_feat(ga, n + 2, false); }
            if (n + 4 < NT) { if (n & 1) issue_feat(gb, n + 4 - NKC); else issue_feat(ga, n + 4 - NKC); }
        }
        // ---- 1 / ||b_j|| of the gathered feature side (F.normalize eps, modules.py:276); the anchor side is pre-normalised
#pragma unroll
        for (int j = 0; j < GI; ++j) {
            float sq = ss[j];
#pragma unroll
            for (int mm = 4; mm >= 1; mm >>= 1) sq += __shfl_xor(sq, mm, 64);
            if (g8 == 0) {
                const int q = GP * j + prow;
                csc[q] = ((PREC == PREC_F16X3 && bsc[j] != 0.f) ? 1.f / bsc[j] : 1.f) / fmaxf(sqrtf(sq), 1e-10f);
            }
        }
    }
    if (stamp_on) ts[7] = __builtin_amdgcn_s_memrealtime();

    __syncthreads();                             // E0: the ring is dead, csc / cscc complete
    if (stamp_on) ts[3] = __builtin_amdgcn_s_memrealtime();
    if (mfma_team) {
        // park fd in the flat output layout.  (sum(fd) of the tile - what the old_mean rendezvous needs - used to be taken here from the
        // accumulators: 0.9 us of masked adds and a 64-lane reduction on every tile's way out; it is now the sum of the row sums the gather
        // waves take from the parked tile anyway, round 5.)
        if (stamp_on && !(prm.debug & 1024)) ts[12] = __builtin_amdgcn_s_memrealtime();
        if (stamp_on && !(prm.debug & 1024)) ts[13] = __builtin_amdgcn_s_memrealtime();
        park_flat(accf, Tfd + a, P, csc, lane, wr, wc);
        if (stamp_on && !(prm.debug & 1024)) ts[14] = __builtin_amdgcn_s_memrealtime();
    }
    __syncthreads();                             // E1: Tfd is complete
    float* omv = red + 8;                        // [0] old_mean, [1] applied
    if (mfma_team) {
        park_flat(accc, Tcd + a, P, cscc, lane, wr, wc);
    } else {
        // row means of fd (fd.mean([3,4]), modules.py:332) from the parked tile, four lanes per row over the 8 gather waves, while
        // the MFMA team parks cd.  A fixed trip count with predicated, independent loads: with a data-dependent loop every
        // ds_read waited for the one before (60 serial round trips, 2.5 us on every tile's way out).
        const int row = gt >> 2, t = gt & 3;
        const float* srcr = Tfd + a + (row < P ? row : 0) * P;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int i = 0; i < TP / 4; i += 2) {
            const int c0 = 4 * i + t, c1 = c0 + 4;
            const float v0 = srcr[c0 < P ? c0 : 0], v1 = srcr[c1 < P ? c1 : 0];
            s0 += c0 < P ? v0 : 0.f;
            s1 += c1 < P ? v1 : 0.f;
        }
        float sfull = s0 + s1;
        sfull += __shfl_xor(sfull, 1, 64);
        sfull += __shfl_xor(sfull, 2, 64);
        if (t == 0 && row < TP) rowmean[row] = (prm.pointwise && row < P) ? sfull / (float)P : 0.f;
        // sum(fd) of the tile = the sum of its row sums, in a fixed order: the 16 rows of a wave (one lane per row counts), then the waves
        float ws = (t == 0 && row < P) ? sfull : 0.f;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) ws += __shfl_xor(ws, m, 64);
        if (lane == 0) red[40 + (wave8 - 4)] = ws;
    }
    __syncthreads();                             // E2: Tcd, rowmean, the eight partial sums of fd
    // The closing ticket is taken HERE, a sweep's length in front of the workgroup's end (round 6): whoever draws the last one finishes the
    // launch (last_workgroup_tail), and that workgroup POLLS every tile's granules by tag - so the ticket need not say "everything is
    // published", only "every workgroup is past this point"; the atomic's round trip (~1 us, formerly behind the last store of the
    // launch's slowest tile) and the first round of the tail's polls now run under the sweeps.  The hand-off words are zeroed by the tail
    // only after every tile's LAST granules (published behind its last read of any of them) have arrived.  (STEGO_DEBUG bit 4: at the end.)
    unsigned tick = 0u;
    const bool early_ticket = !(prm.debug & 4);
    if (tid == 0) {
        const float sfd = ((red[40] + red[41]) + (red[42] + red[43])) + ((red[44] + red[45]) + (red[46] + red[47]));
        // one aligned 8-byte write-through store: {tag, value} (read by the tiles of my pair-set and by the last workgroup)
        __hip_atomic_store(prm.gran + tile, (1ull << 32) | __builtin_bit_cast(unsigned, sfd), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        if (early_ticket) tick = __hip_atomic_fetch_add(prm.done_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (stamp_on) ts[4] = __builtin_amdgcn_s_memrealtime();

    // ---- output sweep, in two parts: cd and the backward's w (+ the two sums) do not need old_mean and leave first, so that
    // the rendezvous of the pair-set - which waits for its slowest tile - is hidden behind two thirds of the stores; the
    // negative loss follows from a second pass over the parked tiles
    const float cmin = prm.cmin, cmax = prm.cmax;
    const float invP = 1.f / (float)P;
    float loss_part = 0.f, clamp_part = 0.f;
    const int nvec = (a + P2 + 3) >> 2;
    // element k of vector f0: w = fd_centred - shift (pass bit in the lsb), cl = clamp(cd), lp = loss without the old_mean term
    auto element = [&](int f0, int k, float fd, float cd, float& w, float& cl, float& lp, bool& ok) {
        const int e = f0 + k - a;
        ok = e >= 0 && e < P2;
        const int r = min(max((int)(((float)e + 0.5f) * invP), 0), P - 1);
        const float wv = fd - (rowmean[r] + shift);
        cl = fminf(fmaxf(cd, cmin), cmax);
        lp = -cl * wv;
        const unsigned pass = (cd >= cmin && cd <= cmax) ? 1u : 0u;
        w = __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, wv) & ~1u) | pass);
    };
    for (int v = tid; v < nvec; v += FUSED_THREADS) {
        const int f0 = 4 * v;
        const f32x4 fd4 = *reinterpret_cast<const f32x4*>(Tfd + f0);
        const f32x4 cd4 = *reinterpret_cast<const f32x4*>(Tcd + f0);
        f32x4 w4;
        bool ok[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float w, cl, lp;
            element(f0, k, fd4[k], cd4[k], w, cl, lp, ok[k]);
            w4[k] = w;
            if (ok[k]) { loss_part += lp; clamp_part += cl; }
        }
        const int e0 = f0 - a;
        if (vec_ok && ok[0] && ok[3]) {
            // streaming stores: nobody in this launch reads the outputs again, and lines that never become dirty in the
            // L2s do not have to be written back when the kernel ends
            __builtin_nontemporal_store(cd4, reinterpret_cast<f32x4*>(cd_out + e0));
            if (w_out) __builtin_nontemporal_store(w4, reinterpret_cast<f32x4*>(w_out + e0));
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (ok[k]) {
                    cd_out[e0 + k] = cd4[k];
                    if (w_out) w_out[e0 + k] = w4[k];
                }
        }
    }
    bool gave_up = false;
    if (loss_out) {                              // (workgroup-uniform)
        if (wave8 == 0) {
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
                    acc += wave_tree_sum(v);
                }
                if (ok_all) om = acc * (1.f / ((float)B * (float)P2));     // same expression as the scalar kernel
                else applied = 0.f;
            } else if (prm.pointwise) {
                applied = 0.f;                   // debug & 32: leave it to the last workgroup
            }
            if (lane == 0) { omv[0] = om; omv[1] = applied; }
        }
        __syncthreads();                         // E3: old_mean
        const float om = omv[0];
        gave_up = omv[1] == 0.f;
        for (int v = tid; v < nvec; v += FUSED_THREADS) {
            const int f0 = 4 * v;
            const f32x4 fd4 = *reinterpret_cast<const f32x4*>(Tfd + f0);
            const f32x4 cd4 = *reinterpret_cast<const f32x4*>(Tcd + f0);
            f32x4 lo4;
            bool ok[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float w, cl, lp;
                element(f0, k, fd4[k], cd4[k], w, cl, lp, ok[k]);
                lo4[k] = __builtin_fmaf(-om, cl, lp);                          // loss = -clamp(cd) * (fd_centred + old_mean - shift)
            }
            const int e0 = f0 - a;
            if (vec_ok && ok[0] && ok[3]) {
                __builtin_nontemporal_store(lo4, reinterpret_cast<f32x4*>(loss_out + e0));
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (ok[k]) loss_out[e0 + k] = lo4[k];
            }
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        loss_part += __shfl_xor(loss_part, m, 64);
        clamp_part += __shfl_xor(clamp_part, m, 64);
    }
    if (lane == 0) { red[16 + wave8 * 2] = loss_part; red[16 + wave8 * 2 + 1] = clamp_part; }
    if (gave_up) __threadfence();                // (rare) whoever repairs this tile must see its cd / loss: release my stores
    __syncthreads();
    // ---- my sums as {tag, value} granules, then one ticket, WITHOUT waiting for the stores in between (a store-ack round
    // trip per workgroup at the very end of the launch): the LAST workgroup of the launch finishes the job below and polls
    // the granules it needs - by then they have been in flight for at least an atomic's round trip.
    unsigned long long* gst = prm.gran + n_tiles;                  // [n_tiles][3]: sum lp, sum clamp, old_mean applied
    if (tid == 0) {
        float s1 = 0.f, s2 = 0.f;
        for (int w = 0; w < FUSED_WAVES; ++w) { s1 += red[16 + w * 2]; s2 += red[16 + w * 2 + 1]; }
        if (gave_up) __hip_atomic_fetch_add(prm.done_cnt + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (event word: see above)
        unsigned long long* g3 = gst + (size_t)tile * 3;
        __hip_atomic_store(g3 + 0, (1ull << 32) | __builtin_bit_cast(unsigned, s1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(g3 + 1, (1ull << 32) | __builtin_bit_cast(unsigned, s2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(g3 + 2, (1ull << 32) | __builtin_bit_cast(unsigned, gave_up ? 0.f : 1.f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!early_ticket) tick = __hip_atomic_fetch_add(prm.done_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fin[0] = tick == gridDim.x - 1 ? 1.f : 0.f;
    }
    if (stamp_on) ts[5] = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
    if (fin[0] == 0.f) return;

    last_workgroup_tail(prm, Tfd, fin + 1, tid, n_tiles, (prm.debug & 256) ? ts : nullptr);
}

// ------------------------------------------------------------------------------------------------------ launch
#if STEGO_FUSED_PART == 0
bool fused_supported(const FusedParams& prm, int precision)
{
    auto cl4 = [&](const MapV& m) {
        return m.sc == 1 && (m.sn % 4) == 0 && (m.sh % 4) == 0 && (m.sw % 4) == 0 &&
               (reinterpret_cast<uintptr_t>(m.p) % 16) == 0 &&
               ((long long)(prm.H - 1) * m.sh + (long long)(prm.W - 1) * m.sw + prm.C) * 4 < (1ll << 31);
    };
    auto cl2 = [&](const MapV& m) {          // channels-last code map; 4-byte aligned pixels are enough (odd K), K >= 3 then (see code_loads)
        return m.sc == 1 && (reinterpret_cast<uintptr_t>(m.p) % 4) == 0;
    };
    if (!(prm.C == 192 || prm.C == 384 || prm.C == 768)) return false;         // NJ instantiations below (ViT-T / ViT-S / ViT-B)
    // One workgroup per compute unit at a time (136 KB of LDS): the in-launch hand-offs (anchors, old_mean) are between workgroups
    // that run at the same time.  More tiles than CUs run as rounds of whole pair-sets (see the kernel), so a pair-set must fit:
    if (prm.B > (device_cu_count() & ~7)) return false;                       // (one pair-set = B tiles per round at least)
    if (!cl4(prm.feats) || !cl4(prm.feats_pos) || !cl2(prm.code) || !cl2(prm.code_pos)) return false;
    if (prm.K > 128 || (prm.K % 2 != 0 && prm.K < 3)) return false;             // four code K-chunks of <= 32 channels
    if (prm.H > 256 || prm.W > 256) return false;                             // packed tap coordinates (8 bits each)
    if ((long long)(prm.H - 1) * prm.code.sh + (long long)(prm.W - 1) * prm.code.sw + prm.K >= (1ll << 29)) return false;
    if ((long long)(prm.H - 1) * prm.code_pos.sh + (long long)(prm.W - 1) * prm.code_pos.sw + prm.K >= (1ll << 29)) return false;
    (void)precision;
    return true;
}

// Zeroes the hand-off words of a workspace (once per workspace; every launch leaves them zero again).
hipError_t prepare_corr_fused(const FusedParams& prm, size_t sync_bytes, hipStream_t stream)
{
    return hipMemsetAsync(prm.anchor_cnt, 0, sync_bytes, stream);
}

hipError_t launch_fused_odd(const FusedParams& prm, int precision, dim3 grid, dim3 block, int lds, hipStream_t stream);      // corr_fused_odd.hip
hipError_t launch_fused_half(const FusedParams& prm, int precision, hipStream_t stream);                                     // corr_fused_half.hip

// The column-half launch (corr_fused_half.hip): small batches - every column half of every tile gets a compute unit of its own (16 B <= CUs
// with five negatives: the reference's batch size of 16), the device is ours alone, the BASELINE widths with an even code dimension, more
// than 64 sample points (below that the second half would be padding).  Any STEGO_DEBUG bit but the stamps (256) keeps the full-tile
// kernel - its ablation / forced-path bits mean that kernel; bit 16384 has no other meaning: same-process A/B of the two launches.
static bool half_launch_covers(const FusedParams& prm, bool shared, int all, int* n_anchor_wg)
{
    if (shared || (prm.debug & ~(256 | 7)) != 0 || !prm.rowg) return false;       // (1: timing ablation of the half kernel - no MFMA; 2: its twelve-wave form; 4: the closing ticket at the end)
    if (!(prm.C == 384 || prm.C == 768) || (prm.K & 1) || prm.P <= 64) return false;
    const int n_items = 2 * prm.n_sets * prm.B;
    if (n_items + 8 > all) return false;
    *n_anchor_wg = (all - n_items) & ~7;
    return true;
}
hipError_t launch_fused_c192(const FusedParams& prm, int precision, dim3 grid, dim3 block, int lds, hipStream_t stream);     // corr_fused_c192.hip

// one instantiation: dynamic LDS attribute, launch
#define STEGO_FUSED_ONE(PR, N, NK, ODD)                                                                \
    do {                                                                                               \
        hipError_t e_ = ensure_dynamic_lds(reinterpret_cast<const void*>(&corr_fused_kernel<PR, N, NK, ODD>), lds); \
        if (e_ != hipSuccess) return e_;                                                               \
        hipLaunchKernelGGL((corr_fused_kernel<PR, N, NK, ODD>), grid, block, lds, stream, prm);        \
        return hipSuccess;                                                                             \
    } while (0)
#define STEGO_FUSED_NK(PR, N, ODD)                                                                     \
    do {                                                                                               \
        if (prm.NKC == 1) STEGO_FUSED_ONE(PR, N, 1, ODD);                                              \
        else if (prm.NKC == 2) STEGO_FUSED_ONE(PR, N, 2, ODD);                                         \
        else if (prm.NKC == 3) STEGO_FUSED_ONE(PR, N, 3, ODD);                                         \
        else STEGO_FUSED_ONE(PR, N, 4, ODD);                                                           \
    } while (0)
static hipError_t launch_fused_even(const FusedParams& prm, int precision, dim3 grid, dim3 block, int lds, hipStream_t stream)
{
    if (precision == PREC_F32) { if (prm.C == 384) STEGO_FUSED_NK(PREC_F32, 3, false); else STEGO_FUSED_NK(PREC_F32, 6, false); }
    else { if (prm.C == 384) STEGO_FUSED_NK(PREC_F16X3, 3, false); else STEGO_FUSED_NK(PREC_F16X3, 6, false); }
}
#undef STEGO_FUSED_NK
#undef STEGO_FUSED_ONE

hipError_t launch_corr_fused(const FusedParams& prm_in, int precision, size_t sync_bytes, bool prepared, bool shared_device,
                             hipStream_t stream, hipEvent_t* ev /* null or [4]: before the memset, before / after the kernel, end */)
{
    FusedParams prm = prm_in;
    const int n_tiles = prm.n_sets * prm.B;
    const int cus = device_cu_count();
    // Every CU gets a workgroup: those beyond the tiles only help with phase 1.  That needs the WHOLE device at once; when other
    // kernels run beside the loss (STEGO_SHARED_DEVICE: the gradient all-reduce of step t overlaps the forward of step t + 1), a
    // helper that cannot be placed would hold up its anchors - and their 7 tiles each - for as long as the other kernel runs.
    // Then the tiles' own workgroups share phase 1 (a third of them take a second pass) and the CUs beyond the tiles stay free.
    const int all = (cus & ~7) < 8 ? 8 : (cus & ~7);
    // rounds of whole pair-sets: as many as fit the compute units at once (all 2 + n_neg of them when n_tiles <= CUs)
    prm.ps_round = all / prm.B < 1 ? 1 : (all / prm.B > prm.n_sets ? prm.n_sets : all / prm.B);
    const int sd = shared_device ? 1 : knob(KNOB_SHARED_DEVICE);       // (the knob: tools only - a per-call setting is a descriptor flag)
    prm.n_owner = sd == 0 ? all : (n_tiles < all ? n_tiles : all);
    if (sd > 8) {                          // (tools: an explicit number of phase-1 owners between the two, a multiple of 8)
        const int want = sd & ~7;
        prm.n_owner = want < prm.n_owner ? prm.n_owner : (want > all ? all : want);
    }
    // phase 1 carried by the light workgroups (see the kernel): one round, whole images per XCD, a workgroup on every CU, and what the
    // light ones cannot take fits one pass (16 rows at C = 384, 8 at 768) of the others
    {
        const int G = prm.C <= 384 ? 2 : 1, rowsL = 24 * G + ((prm.C <= 384 && !(prm.debug & 2048)) ? 16 : 0), rowsH = 8 * G;
        const int per_x = prm.B / 8, slots = all / 8, tile_slots = n_tiles / 8;
        const int n_light = per_x + (slots - tile_slots), n_heavy = slots - n_light;
        const int R = per_x * TP, rest = R - n_light * rowsL;
        prm.p1_light = !(prm.debug & 128) && prm.B % 8 == 0 && n_tiles % 8 == 0 && n_tiles <= all && prm.n_owner == all && prm.ps_round >= prm.n_sets &&
                       slots >= tile_slots && n_light > 0 && (rest <= 0 || (n_heavy > 0 && (rest + n_heavy - 1) / n_heavy <= rowsH)) ? 1 : 0;
    }
    prm.timeout_ticks = 20000;                                    // 200 us of the 100 MHz clock (debug 64: the anchor wait gives up after 1 us)
    const int lds = RING_LDS_BYTES;
    hipError_t e = hipSuccess;
    if (ev) (void)hipEventRecord(ev[0], stream);
    if (!prepared && (e = prepare_corr_fused(prm, sync_bytes, stream)) != hipSuccess) return e;
    if (ev) (void)hipEventRecord(ev[1], stream);
    if (half_launch_covers(prm, sd != 0, all, &prm.n_anchor_wg)) {
        if ((e = launch_fused_half(prm, precision, stream)) != hipSuccess) return e;
        if ((e = hipGetLastError()) != hipSuccess) return e;
        if (ev) { (void)hipEventRecord(ev[2], stream); (void)hipEventRecord(ev[3], stream); }
        return hipGetLastError();
    }
    const dim3 grid(n_tiles > prm.n_owner ? n_tiles : prm.n_owner), block(FUSED_THREADS);
    e = prm.C == 192 ? launch_fused_c192(prm, precision, grid, block, lds, stream)
        : (prm.K & 1) ? launch_fused_odd(prm, precision, grid, block, lds, stream) : launch_fused_even(prm, precision, grid, block, lds, stream);
    if (e != hipSuccess) return e;
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (ev) { (void)hipEventRecord(ev[2], stream); (void)hipEventRecord(ev[3], stream); }
    return hipGetLastError();
}


#elif STEGO_FUSED_PART == 1
// one instantiation: dynamic LDS attribute, launch
#define STEGO_FUSED_ONE(PR, N, NK, ODD)                                                                \
    do {                                                                                               \
        hipError_t e_ = ensure_dynamic_lds(reinterpret_cast<const void*>(&corr_fused_kernel<PR, N, NK, ODD>), lds); \
        if (e_ != hipSuccess) return e_;                                                               \
        hipLaunchKernelGGL((corr_fused_kernel<PR, N, NK, ODD>), grid, block, lds, stream, prm);        \
        return hipSuccess;                                                                             \
    } while (0)
#define STEGO_FUSED_NK(PR, N, ODD)                                                                     \
    do {                                                                                               \
        if (prm.NKC == 1) STEGO_FUSED_ONE(PR, N, 1, ODD);                                              \
        else if (prm.NKC == 2) STEGO_FUSED_ONE(PR, N, 2, ODD);                                         \
        else if (prm.NKC == 3) STEGO_FUSED_ONE(PR, N, 3, ODD);                                         \
        else STEGO_FUSED_ONE(PR, N, 4, ODD);                                                           \
    } while (0)
hipError_t launch_fused_odd(const FusedParams& prm, int precision, dim3 grid, dim3 block, int lds, hipStream_t stream)
{
    if (precision == PREC_F32) { if (prm.C == 384) STEGO_FUSED_NK(PREC_F32, 3, true); else STEGO_FUSED_NK(PREC_F32, 6, true); }
    else { if (prm.C == 384) STEGO_FUSED_NK(PREC_F16X3, 3, true); else STEGO_FUSED_NK(PREC_F16X3, 6, true); }
}
#undef STEGO_FUSED_NK
#undef STEGO_FUSED_ONE
#elif STEGO_FUSED_PART == 2
// one instantiation: dynamic LDS attribute, launch
#define STEGO_FUSED_ONE(PR, N, NK, ODD)                                                                \
    do {                                                                                               \
        hipError_t e_ = ensure_dynamic_lds(reinterpret_cast<const void*>(&corr_fused_kernel<PR, N, NK, ODD>), lds); \
        if (e_ != hipSuccess) return e_;                                                               \
        hipLaunchKernelGGL((corr_fused_kernel<PR, N, NK, ODD>), grid, block, lds, stream, prm);        \
        return hipSuccess;                                                                             \
    } while (0)
#define STEGO_FUSED_NK(PR, N, ODD)        /* C = 192: six feature stages, at least two code chunks (c_api.hip: geometry) */ \
    do {                                                                                               \
        if (prm.NKC == 2) STEGO_FUSED_ONE(PR, N, 2, ODD);                                              \
        else if (prm.NKC == 3) STEGO_FUSED_ONE(PR, N, 3, ODD);                                         \
        else if (prm.NKC == 4) STEGO_FUSED_ONE(PR, N, 4, ODD);                                         \
        else return hipErrorInvalidValue;                                                              \
    } while (0)
hipError_t launch_fused_c192(const FusedParams& prm, int precision, dim3 grid, dim3 block, int lds, hipStream_t stream)
{
    if (precision == PREC_F32) { if (prm.K & 1) STEGO_FUSED_NK(PREC_F32, 2, true); else STEGO_FUSED_NK(PREC_F32, 2, false); }
    else { if (prm.K & 1) STEGO_FUSED_NK(PREC_F16X3, 2, true); else STEGO_FUSED_NK(PREC_F16X3, 2, false); }
}
#undef STEGO_FUSED_NK
#undef STEGO_FUSED_ONE
#endif

}  // namespace stego
