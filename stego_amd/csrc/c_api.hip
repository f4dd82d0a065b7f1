// extern "C" boundary of libstego_corr.so (declared in include/stego_corr.h).
// Host-side validation + parameter packing + launch sequencing; all arithmetic is in the HIP kernels.
#include <cstdlib>
#include <limits>

#include "../../include/stego_corr.h"
#include "corr_common.h"
#include "host_util.h"
#include "../../include/stego_head.h"
#include "corr_wide.h"

namespace stego {
hipError_t launch_corr_sample(const SampleParams& prm, int precision, hipStream_t stream);
hipError_t launch_corr_tile(const CorrParams& prm, int precision, hipStream_t stream);
hipError_t launch_corr_finalize(const CorrParams& prm, hipStream_t stream);
bool fused_supported(const FusedParams& prm, int precision);
hipError_t launch_corr_fused(const FusedParams& prm, int precision, size_t sync_bytes, bool prepared, bool shared_device, hipStream_t stream, hipEvent_t* ev);
hipError_t prepare_corr_fused(const FusedParams& prm, size_t sync_bytes, hipStream_t stream);
hipError_t launch_fast_draws(const long long* seed, long long n_coord, int n_neg, int B, float* c1, float* c2, long long* perms,
                             hipStream_t stream);
hipError_t launch_finish_draws(const float* u1, const float* u2, long long n_coord, const long long* const* raw, int n_neg,
                               int B, float* c1, float* c2, long long* perms, hipStream_t stream);
hipError_t launch_ref_draws(unsigned long long seed, unsigned long long offset, const long long* seed_ptr, const long long* offset_ptr,
                            int variant, long long n_coord, int n_neg, int B,
                            int cus, int threads_per_cu, float* c1, float* c2, long long* perms, hipStream_t stream);
unsigned long long ref_draws_advance(long long n_coord, int n_neg, int B, int variant, int cus, int threads_per_cu);
unsigned long long ref_masks_advance(long long numel, int n_masks, int variant, int cus, int threads_per_cu);
hipError_t launch_ref_masks(unsigned long long seed, unsigned long long offset, const long long* seed_ptr, const long long* offset_ptr,
                            int variant, int n_masks, long long numel, float keep_prob, int cus, int threads_per_cu, float* out,
                            hipStream_t stream);
size_t dense_workspace_bytes(int B, int C, int M, int N);
hipError_t launch_dense_corr_panels(const void* imgA, const float* rsA, int imagesA, const void* imgB, const float* rsB, int B, int C, int M, int N,
                                    float* out, float* rowsum, hipStream_t stream);
size_t dense_panel_image_bytes(int C, int P);
hipError_t launch_dense_corr(const MapV& a, const MapV& b, int B, int C, int H1, int W1, int H2, int W2, int normalize,
                             float* out, void* ws, hipStream_t stream);
size_t knn_workspace_bytes(long long N, int D, int k, long long q_count);
hipError_t launch_knn(const float* X, long long N, int D, long long ldx, int k, int normalize, long long q_begin,
                      long long q_count, long long* out_idx, float* out_val, void* ws, hipStream_t stream);
hipError_t launch_corr_bwd(const BwdParams& prm, hipStream_t stream);
}  // namespace stego

using namespace stego;

namespace {

bool aligned4(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 3) == 0; }
size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// largest element offset inside one image must fit in int32 (the kernels use 32-bit tap offsets)
int to_mapv(const StegoMap* m, int channels, int H, int W, MapV* out)
{
    if (!m || !m->data) return STEGO_ERR_NULL;
    if (!aligned4(m->data)) return STEGO_ERR_ALIGN;
    const int64_t lim = std::numeric_limits<int32_t>::max();
    if (m->stride_c < 0 || m->stride_h < 0 || m->stride_w < 0) return STEGO_ERR_UNSUPPORTED;
    const int64_t span = m->stride_c * (channels - 1) + m->stride_h * (H - 1) + m->stride_w * (W - 1);
    if (span >= lim || m->stride_c >= lim || m->stride_h >= lim || m->stride_w >= lim) return STEGO_ERR_UNSUPPORTED;
    out->p = m->data;
    out->sn = m->stride_n;
    out->sc = (int)m->stride_c;
    out->sh = (int)m->stride_h;
    out->sw = (int)m->stride_w;
    return STEGO_OK;
}

int check_desc(const StegoCorrDesc* d, bool helper)
{
    if (!d) return STEGO_ERR_NULL;
    if (d->B <= 0 || d->C <= 0 || d->K <= 0 || d->H <= 0 || d->W <= 0) return STEGO_ERR_SHAPE;
    if (d->H > 32767 || d->W > 32767) return STEGO_ERR_UNSUPPORTED;
    if (d->K > (helper ? 72 : 128)) return STEGO_ERR_UNSUPPORTED;    /* 72 < K <= 128: fused forward only (plan_fwd) */
    if (helper) {
        if ((int64_t)d->H * d->W > TP) return STEGO_ERR_UNSUPPORTED;
    } else {
        if (d->S <= 0 || d->n_neg < 0) return STEGO_ERR_SHAPE;
        if (d->S > 16) return STEGO_ERR_UNSUPPORTED;
        if (d->n_neg + 2 > 256) return STEGO_ERR_UNSUPPORTED;
        if (d->S * d->S > TP && !wide_supported(d->B, d->C, d->K, d->S, d->n_neg)) return STEGO_ERR_UNSUPPORTED;   /* 129 .. 256 points: corr_wide.hip */
    }
    if (d->precision != STEGO_PREC_F32 && d->precision != STEGO_PREC_F16X3) return STEGO_ERR_UNSUPPORTED;
    if (d->flags & ~STEGO_FLAG_SHARED_DEVICE) return STEGO_ERR_UNSUPPORTED;
    return STEGO_OK;
}

// ---- buffer geometry (all derived from the descriptor)
struct Geometry {
    int n_roles, nset, NCH, KQ, LDK;
    int NCH2, NKC, kper;            // fused forward: C / 32 feature stages, code K-chunks of kper channels
    size_t rowg_off;
    size_t stats_bytes, sync_bytes, fs_bytes, csf_bytes, cs_bytes, nrm_bytes, tap_bytes, ctx_bytes, ws_bytes, bwd_ws_bytes;
    size_t bwd_dt_bytes, bwd_slots_bytes, bwd_pool_bytes;    // lists-first backward: DT rows, list slots, overflow pool (0: not covered)
};
constexpr size_t BWD_STAMP_BYTES = 65536;    // debug stamps behind the DT rows

Geometry geometry(const StegoCorrDesc* d, bool helper)
{
    Geometry g;
    g.n_roles = helper ? 2 : 2 + d->n_neg;
    g.nset = g.n_roles * d->B;
    g.NCH = (d->C + KC - 1) / KC;
    g.KQ = (d->K + 7) & ~7;
    g.LDK = g.KQ + 4;
    const size_t n_tiles = (size_t)(helper ? 1 : 2 + d->n_neg) * d->B;
    g.stats_bytes = round_up(n_tiles * 4 * sizeof(float) + 1024 + n_tiles * (128 + 1024) + 40960, 256);     // tail: debug stamps (16 + 128 per tile; the column-half launch: 16 per workgroup of a whole-device grid)
    const size_t fside = d->precision == STEGO_PREC_F16X3 ? (size_t)2 * TP * LDH * 2 : (size_t)TP * LDA * 4;
    g.fs_bytes = round_up((size_t)d->B * g.NCH * fside + 1024, 256);            // anchor sets only (three-launch layout)
    // fused forward: anchor operands as ring stages of 16 KB (32 channels: fp16 hi + lo, or fp32), codes as K-chunks
    g.NCH2 = (d->C + 31) / 32;
    g.NKC = (g.KQ + 31) / 32;
    if (d->C == 192 && g.NKC < 2) g.NKC = 2;      // six feature stages: the forward's static stream head needs eight stages in all
    g.kper = (((g.KQ + g.NKC - 1) / g.NKC) + 7) & ~7;
    const size_t fs_ring = round_up((size_t)d->B * g.NCH2 * 16384 + 1024, 256);
    if (fs_ring > g.fs_bytes) g.fs_bytes = fs_ring;
    g.csf_bytes = round_up((size_t)d->B * g.NKC * 16384 + 1024, 256);
    g.cs_bytes = round_up((size_t)g.nset * TP * g.LDK * sizeof(float) + 1024, 256);
    g.nrm_bytes = round_up((size_t)g.nset * TP * sizeof(float), 256);
    g.tap_bytes = round_up((size_t)g.nset * TP * 16, 256);        // each of tapyx / tapw
    g.ctx_bytes = g.cs_bytes + g.nrm_bytes + 2 * g.tap_bytes;
    // in-launch hand-off words of the fused forward: one counter per anchor + one 8-byte granule per tile
    g.rowg_off = round_up((size_t)d->B * 256 + 2 * n_tiles * 8 * 4, 256);       // column-half launch: 2 x tiles items, 4 granules each, then 128 row-sum granules per item
    g.sync_bytes = g.rowg_off + round_up(2 * n_tiles * (size_t)TP * 8, 256) + 256;   // + 3 more granules per tile: sum lp, sum clamp, applied;   // counters 256 B apart (ANCHOR_CNT_STRIDE); + the done counter
    g.ws_bytes = g.stats_bytes + g.sync_bytes + g.fs_bytes + g.csf_bytes + g.ctx_bytes;
    g.bwd_dt_bytes = round_up(n_tiles * 2 * TP * g.LDK * sizeof(float), 256);
    g.bwd_slots_bytes = g.bwd_pool_bytes = 0;
    if (!helper && d->H <= 64 && d->W <= 64) {
        // corr_unsample_list_kernel: a 1 KB slot per (image, pixel row, 16-pixel bin) unit of orig_code, a 256-byte slot per unit of
        // orig_code_pos, and room for every entry (16 bytes, at most 4 per (item, point)) in the overflow pool - an item is one
        // (tile, side) whose gradient rows land in a destination image
        const size_t units = (size_t)d->B * d->H * ((d->W + 15) / 16);
        g.bwd_slots_bytes = round_up(units * (1024 + 256), 256);
        g.bwd_pool_bytes = round_up((size_t)4 * d->S * d->S * ((size_t)(2 + d->n_neg) * d->B + (size_t)d->n_neg * d->B + d->B) * 16, 256);
    }
    g.bwd_ws_bytes = g.bwd_dt_bytes + BWD_STAMP_BYTES + g.bwd_slots_bytes + g.bwd_pool_bytes;
    return g;
}

int hip_rc(hipError_t e) { return e == hipSuccess ? STEGO_OK : STEGO_ERR_HIP + (int)e; }

// feature_samples 12 .. 16: more points than one tile holds - the multi-launch path of corr_wide.hip behind the same entry points
bool is_wide(const StegoCorrDesc* d) { return d->S * d->S > TP; }
WideGeom wide_geom(const StegoCorrDesc* d) { return wide_geometry(d->B, d->C, d->K, d->H, d->W, d->S, d->n_neg); }

int wide_fwd(const StegoCorrDesc* d, const StegoMap* feats, const StegoMap* feats_pos, const StegoMap* code, const StegoMap* code_pos,
             const float* coords1, const float* coords2, const int64_t* perms, float* loss_means, float* pos_intra_cd, float* pos_inter_cd,
             float* neg_inter_loss, float* neg_inter_cd, float* saved_w, float* saved_mean, void* saved_ctx, void* workspace,
             size_t workspace_bytes, hipStream_t stream)
{
    (void)hipGetLastError();
    if (!workspace || !coords1 || !coords2 || !loss_means || !pos_intra_cd || !pos_inter_cd) return STEGO_ERR_NULL;
    if (d->n_neg > 0 && (!perms || !neg_inter_loss || !neg_inter_cd)) return STEGO_ERR_NULL;
    if ((saved_w == nullptr) != (saved_mean == nullptr)) return STEGO_ERR_NULL;
    if (workspace_bytes < wide_geom(d).ws_bytes) return STEGO_ERR_WORKSPACE;
    const StegoMap* maps[4] = {feats, feats_pos, code, code_pos};
    for (const StegoMap* m : maps) {
        if (!m || !m->data) return STEGO_ERR_NULL;
        if (!aligned4(m->data)) return STEGO_ERR_ALIGN;
    }
    WideFwdArgs a{};
    a.feats = feats; a.feats_pos = feats_pos; a.code = code; a.code_pos = code_pos;
    a.coords1 = coords1; a.coords2 = coords2; a.perms = reinterpret_cast<const long long*>(perms);
    a.loss_means = loss_means; a.intra_cd = pos_intra_cd; a.inter_cd = pos_inter_cd; a.neg_loss = neg_inter_loss; a.neg_cd = neg_inter_cd;
    a.saved_w = saved_w; a.saved_mean = saved_mean; a.saved_ctx = saved_ctx; a.workspace = workspace;
    a.B = d->B; a.C = d->C; a.K = d->K; a.H = d->H; a.W = d->W; a.S = d->S; a.n_neg = d->n_neg; a.pointwise = d->pointwise ? 1 : 0;
    a.cmin = d->zero_clamp ? 0.0f : -9999.0f;
    a.cmax = d->stabalize ? 0.8f : std::numeric_limits<float>::infinity();
    a.shift[0] = d->pos_intra_shift; a.shift[1] = d->pos_inter_shift; a.shift[2] = d->neg_inter_shift;
    return hip_rc(launch_wide_fwd(a, stream));
}

// Measurement knobs: host_util.h (read from the environment once at load; tools flip them with stego_debug_set).

int fill_bwd_ctx(const StegoCorrDesc* d, bool helper, const void* saved_ctx, void* workspace, size_t workspace_bytes,
                 BwdParams* prm)
{
    const Geometry g = geometry(d, helper);
    if (!saved_ctx || !workspace) return STEGO_ERR_NULL;
    if (workspace_bytes < g.bwd_ws_bytes) return STEGO_ERR_WORKSPACE;
    if ((size_t)d->W * d->K * 4 > 96 * 1024) return STEGO_ERR_UNSUPPORTED;      // one image row must fit the LDS band
    if (!helper && (size_t)d->n_neg * d->B + d->n_neg + 2 > 1024) return STEGO_ERR_UNSUPPORTED; // unsample contribution table
    const unsigned char* ctx = static_cast<const unsigned char*>(saved_ctx);
    prm->cs = reinterpret_cast<const float*>(ctx);
    prm->nrm = reinterpret_cast<const float*>(ctx + g.cs_bytes);
    prm->tapyx = reinterpret_cast<const int4*>(ctx + g.cs_bytes + g.nrm_bytes);
    prm->tapw = reinterpret_cast<const float4*>(ctx + g.cs_bytes + g.nrm_bytes + g.tap_bytes);
    prm->dt = static_cast<float*>(workspace);
    if (!helper && g.bwd_pool_bytes && g.bwd_dt_bytes < ((size_t)1 << 32) && g.bwd_pool_bytes < ((size_t)1 << 32) &&
        g.bwd_slots_bytes < ((size_t)1 << 32)) {
        unsigned char* ws = static_cast<unsigned char*>(workspace);
        prm->uslots = reinterpret_cast<unsigned*>(ws + g.bwd_dt_bytes + BWD_STAMP_BYTES);
        prm->upool = reinterpret_cast<unsigned*>(ws + g.bwd_dt_bytes + BWD_STAMP_BYTES + g.bwd_slots_bytes);
        prm->dt_bytes = (unsigned)g.bwd_dt_bytes;
        prm->uslots_bytes = (unsigned)g.bwd_slots_bytes;
        prm->upool_bytes = (unsigned)g.bwd_pool_bytes;
    }
    prm->KQ = g.KQ;
    prm->LDK = g.LDK;
    return STEGO_OK;
}

struct FwdPlan {
    CorrParams tile;
    SampleParams samp;
    FusedParams fused;
    size_t sync_bytes;
    bool use_fused;
    bool shared_device;
    int precision;
};

int plan_fwd(const StegoCorrDesc* d, bool helper, const StegoMap* feats, const StegoMap* feats_pos, const StegoMap* code,
             const StegoMap* code_pos, const float* coords1, const float* coords2, const int64_t* perms,
             float* loss_means, float* pos_intra_cd, float* pos_inter_cd, float* neg_inter_loss, float* neg_inter_cd,
             float* saved_w, float* saved_mean, void* saved_ctx, void* workspace, size_t workspace_bytes, FwdPlan* out)
{
    (void)hipGetLastError();          // drop any stale sticky error of an earlier (failed) call
    int rc = check_desc(d, helper);
    if (rc) return rc;
    if (!workspace) return STEGO_ERR_NULL;
    if (!helper) {
        if (!coords1 || !coords2 || !loss_means || !pos_intra_cd || !pos_inter_cd) return STEGO_ERR_NULL;
        if (d->n_neg > 0 && (!perms || !neg_inter_loss || !neg_inter_cd)) return STEGO_ERR_NULL;
    } else if (!neg_inter_loss || !neg_inter_cd) {
        return STEGO_ERR_NULL;
    }
    if ((saved_w == nullptr) != (saved_mean == nullptr)) return STEGO_ERR_NULL;
    const Geometry g = geometry(d, helper);
    if (workspace_bytes < g.ws_bytes) return STEGO_ERR_WORKSPACE;

    CorrParams prm{};
    if ((rc = to_mapv(feats, d->C, d->H, d->W, &prm.feats))) return rc;
    if ((rc = to_mapv(feats_pos, d->C, d->H, d->W, &prm.feats_pos))) return rc;
    if ((rc = to_mapv(code, d->K, d->H, d->W, &prm.code))) return rc;
    if ((rc = to_mapv(code_pos, d->K, d->H, d->W, &prm.code_pos))) return rc;
    prm.coords1 = coords1; prm.coords2 = coords2;
    prm.perms = reinterpret_cast<const long long*>(perms);  /* int64_t == long long on LP64 */
    prm.intra_cd = pos_intra_cd; prm.inter_cd = pos_inter_cd;
    prm.neg_loss = neg_inter_loss; prm.neg_cd = neg_inter_cd;
    prm.saved_w = saved_w; prm.saved_mean = saved_mean; prm.loss_means = loss_means;
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    prm.stats = reinterpret_cast<float*>(ws);
    unsigned char* sync = ws + g.stats_bytes;
    unsigned char* fs = sync + g.sync_bytes;
    unsigned char* csf = fs + g.fs_bytes;
    unsigned char* ctx = saved_ctx ? static_cast<unsigned char*>(saved_ctx) : csf + g.csf_bytes;
    prm.fs = fs;
    prm.cs = reinterpret_cast<const float*>(ctx);
    prm.NCH = g.NCH; prm.KQ = g.KQ; prm.LDK = g.LDK;
    prm.B = d->B; prm.C = d->C; prm.K = d->K; prm.H = d->H; prm.W = d->W;
    prm.S = helper ? d->W : d->S;
    prm.P = helper ? d->H * d->W : d->S * d->S;
    prm.n_neg = helper ? 0 : d->n_neg;
    prm.n_sets = helper ? 1 : 2 + d->n_neg;
    prm.mode = helper ? 1 : 0;
    prm.pointwise = d->pointwise ? 1 : 0;
    prm.cmin = d->zero_clamp ? 0.0f : -9999.0f;                                   // modules.py:337-340
    prm.cmax = d->stabalize ? 0.8f : std::numeric_limits<float>::infinity();      // modules.py:342-345
    prm.shift[0] = d->pos_intra_shift;
    prm.shift[1] = helper ? d->pos_intra_shift : d->pos_inter_shift;
    prm.shift[2] = helper ? d->pos_intra_shift : d->neg_inter_shift;
    prm.debug = knob(KNOB_DEBUG);

    SampleParams sp{};
    sp.feats = prm.feats; sp.feats_pos = prm.feats_pos; sp.code = prm.code; sp.code_pos = prm.code_pos;
    sp.coords1 = coords1; sp.coords2 = coords2; sp.perms = prm.perms;
    sp.fs = fs;
    sp.cs = reinterpret_cast<float*>(ctx);
    sp.nrm = reinterpret_cast<float*>(ctx + g.cs_bytes);
    sp.tapyx = reinterpret_cast<int4*>(ctx + g.cs_bytes + g.nrm_bytes);
    sp.tapw = reinterpret_cast<float4*>(ctx + g.cs_bytes + g.nrm_bytes + g.tap_bytes);
    sp.B = d->B; sp.C = d->C; sp.K = d->K; sp.H = d->H; sp.W = d->W; sp.S = prm.S; sp.P = prm.P;
    sp.n_roles = g.n_roles; sp.feat_roles = 1; sp.NCH = g.NCH; sp.KQ = g.KQ; sp.LDK = g.LDK; sp.mode = prm.mode;
    sp.debug = knob(KNOB_DEBUG_SAMPLE);

    prm.tapyx = sp.tapyx; prm.tapw = sp.tapw;
    out->tile = prm;
    out->samp = sp;
    out->precision = d->precision;
    out->shared_device = (d->flags & STEGO_FLAG_SHARED_DEVICE) != 0;

    // the fused single-launch forward (corr_fused.hip) covers forward() on channels-last maps of the ViT widths;
    // everything else (helper mode, generic strides, other widths) takes the three-launch path
    FusedParams fp{};
    fp.feats = prm.feats; fp.feats_pos = prm.feats_pos; fp.code = prm.code; fp.code_pos = prm.code_pos;
    fp.coords1 = coords1; fp.coords2 = coords2; fp.perms = prm.perms;
    fp.intra_cd = prm.intra_cd; fp.inter_cd = prm.inter_cd; fp.neg_loss = prm.neg_loss; fp.neg_cd = prm.neg_cd;
    fp.saved_w = saved_w; fp.saved_mean = saved_mean; fp.loss_means = loss_means;
    fp.stats = prm.stats;
    fp.anchor_cnt = reinterpret_cast<unsigned*>(sync);
    fp.gran = reinterpret_cast<unsigned long long*>(sync + (size_t)d->B * 256);
    fp.done_cnt = reinterpret_cast<unsigned*>(sync + g.sync_bytes - 256);
    fp.rowg = reinterpret_cast<unsigned long long*>(sync + g.rowg_off);
    fp.fs = fs;
    fp.csf = csf;
    fp.cs = sp.cs; fp.nrm = sp.nrm; fp.tapyx = sp.tapyx; fp.tapw = sp.tapw;
    fp.fs_bytes = (unsigned)g.fs_bytes; fp.csf_bytes = (unsigned)g.csf_bytes; fp.cs_bytes = (unsigned)g.cs_bytes;
    fp.NCH2 = g.NCH2; fp.NKC = g.NKC; fp.kper = g.kper; fp.KQ = g.KQ; fp.LDK = g.LDK;
    fp.B = prm.B; fp.C = prm.C; fp.K = prm.K; fp.H = prm.H; fp.W = prm.W; fp.S = prm.S; fp.P = prm.P;
    fp.n_neg = prm.n_neg; fp.n_sets = prm.n_sets;
    fp.pointwise = prm.pointwise;
    fp.debug = prm.debug;
    fp.cmin = prm.cmin; fp.cmax = prm.cmax;
    fp.shift[0] = prm.shift[0]; fp.shift[1] = prm.shift[1]; fp.shift[2] = prm.shift[2];
    out->fused = fp;
    out->sync_bytes = g.sync_bytes;
    const int variant = knob(KNOB_FWD_VARIANT);              // -1 (default) automatic, 1 three launches, 2 fused
    out->use_fused = !helper && variant != 1 && g.fs_bytes < ((size_t)1 << 31) && g.cs_bytes < ((size_t)1 << 31) &&
                     fused_supported(fp, d->precision);
    // code dimensions above 72 exist on the fused path only (channels-last maps of the ViT widths): the
    // three-launch kernels hold both code operands in one LDS stage
    if (d->K > 72 && !out->use_fused) return STEGO_ERR_UNSUPPORTED;
    return STEGO_OK;
}

hipError_t run_fwd(const FwdPlan& pl, hipStream_t s, hipEvent_t* ev /* null or [4] */, bool prepared = false)
{
    hipError_t e;
    if (pl.use_fused) return launch_corr_fused(pl.fused, pl.precision, pl.sync_bytes, prepared, pl.shared_device, s, ev);
    if (ev) (void)hipEventRecord(ev[0], s);
    if ((e = launch_corr_sample(pl.samp, pl.precision, s)) != hipSuccess) return e;
    if (ev) (void)hipEventRecord(ev[1], s);
    if ((e = launch_corr_tile(pl.tile, pl.precision, s)) != hipSuccess) return e;
    if (ev) (void)hipEventRecord(ev[2], s);
    if ((e = launch_corr_finalize(pl.tile, s)) != hipSuccess) return e;
    if (ev) (void)hipEventRecord(ev[3], s);
    return hipSuccess;
}

}  // namespace

extern "C" {

int stego_abi_version(void) { return STEGO_ABI_VERSION; }

int stego_debug_set(int32_t which, int32_t value)
{
    if (which < 0 || which >= KNOB_COUNT) return STEGO_ERR_SHAPE;
    set_knob(which, value);
    return STEGO_OK;
}

int stego_debug_occupy(int32_t n_workgroups, int32_t lds_bytes, int32_t microseconds, stego_stream_t stream)
{
    (void)hipGetLastError();
    if (n_workgroups <= 0 || lds_bytes < 1024 || lds_bytes > 160 * 1024 || microseconds < 0 || microseconds > 100000) return STEGO_ERR_SHAPE;
    return hip_rc(launch_occupy(n_workgroups, lds_bytes, microseconds, static_cast<hipStream_t>(stream)));
}

const char* stego_error_string(int code)
{
    switch (code) {
        case STEGO_OK: return "ok";
        case STEGO_ERR_NULL: return "required pointer is NULL";
        case STEGO_ERR_SHAPE: return "bad or inconsistent dimension";
        case STEGO_ERR_UNSUPPORTED: return "unsupported configuration (limits: S<=16, K<=128 (K>72: channels-last maps with C = 192 / 384 / 768, B <= compute units), n_neg<=254, fp32 maps, <2^31 elements per image, no unknown flags)";
        case STEGO_ERR_WORKSPACE: return "workspace too small";
        case STEGO_ERR_ALIGN: return "pointer not 4-byte aligned";
        default: return code >= STEGO_ERR_HIP ? "HIP runtime error (code - 1000 = hipError_t)" : "unknown error";
    }
}

size_t stego_corr_workspace_bytes(const StegoCorrDesc* desc)
{
    if (check_desc(desc, false) != STEGO_OK) return 0;
    if (is_wide(desc)) return wide_geom(desc).ws_bytes;
    return geometry(desc, false).ws_bytes;
}

const uint32_t* stego_corr_event_counters(const StegoCorrDesc* desc, const void* workspace, size_t workspace_bytes)
{
    if (check_desc(desc, false) != STEGO_OK || !workspace || is_wide(desc)) return nullptr;      // (no hand-off words in the multi-launch path)
    const Geometry g = geometry(desc, false);
    if (workspace_bytes < g.ws_bytes) return nullptr;
    // the block of the done counter (plan_fwd: sync + sync_bytes - 256): word 0 is the counter, words 1-2 the events
    return reinterpret_cast<const uint32_t*>(static_cast<const unsigned char*>(workspace) + g.stats_bytes + g.sync_bytes - 256) + 1;
}

size_t stego_corr_saved_ctx_bytes(const StegoCorrDesc* desc)
{
    if (check_desc(desc, false) != STEGO_OK) return 0;
    if (is_wide(desc)) return wide_geom(desc).ctx_bytes;
    return geometry(desc, false).ctx_bytes;
}

size_t stego_corr_bwd_workspace_bytes(const StegoCorrDesc* desc)
{
    if (check_desc(desc, false) != STEGO_OK) return 0;
    if (is_wide(desc)) return wide_geom(desc).bwd_ws_bytes;
    return geometry(desc, false).bwd_ws_bytes;
}

size_t stego_corr_helper_bwd_workspace_bytes(const StegoCorrDesc* desc)
{
    if (check_desc(desc, true) != STEGO_OK) return 0;
    return geometry(desc, true).bwd_ws_bytes;
}

size_t stego_corr_helper_workspace_bytes(const StegoCorrDesc* desc)
{
    if (check_desc(desc, true) != STEGO_OK) return 0;
    return geometry(desc, true).ws_bytes;
}

size_t stego_corr_helper_saved_ctx_bytes(const StegoCorrDesc* desc)
{
    if (check_desc(desc, true) != STEGO_OK) return 0;
    return geometry(desc, true).ctx_bytes;
}

int stego_corr_fwd(const StegoCorrDesc* d, const StegoMap* feats, const StegoMap* feats_pos, const StegoMap* code,
                   const StegoMap* code_pos, const float* coords1, const float* coords2, const int64_t* perms,
                   float* loss_means, float* pos_intra_cd, float* pos_inter_cd, float* neg_inter_loss,
                   float* neg_inter_cd, float* saved_w, float* saved_mean, void* saved_ctx, void* workspace,
                   size_t workspace_bytes, stego_stream_t stream)
{
    FwdPlan pl;
    int rc = check_desc(d, false);
    if (rc) return rc;
    if (is_wide(d)) return wide_fwd(d, feats, feats_pos, code, code_pos, coords1, coords2, perms, loss_means, pos_intra_cd, pos_inter_cd, neg_inter_loss,
                                    neg_inter_cd, saved_w, saved_mean, saved_ctx, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
    rc = plan_fwd(d, false, feats, feats_pos, code, code_pos, coords1, coords2, perms, loss_means, pos_intra_cd,
                  pos_inter_cd, neg_inter_loss, neg_inter_cd, saved_w, saved_mean, saved_ctx, workspace,
                  workspace_bytes, &pl);
    if (rc) return rc;
    return hip_rc(run_fwd(pl, static_cast<hipStream_t>(stream), nullptr));
}

int stego_finish_draws(const float* u1, const float* u2, int64_t n_coord, const int64_t* const* raw_perms, int32_t n_neg,
                       int32_t B, float* coords1, float* coords2, int64_t* perms, stego_stream_t stream)
{
    (void)hipGetLastError();
    if (n_coord < 0 || n_neg < 0 || n_neg > 16 || B < 1) return STEGO_ERR_SHAPE;
    if (n_coord > 0 && (!u1 || !u2 || !coords1 || !coords2)) return STEGO_ERR_NULL;
    if (n_neg > 0 && (!raw_perms || !perms)) return STEGO_ERR_NULL;
    for (int i = 0; i < n_neg; ++i)
        if (!raw_perms[i]) return STEGO_ERR_NULL;
    return hip_rc(launch_finish_draws(u1, u2, n_coord, reinterpret_cast<const long long* const*>(raw_perms), n_neg, B, coords1,
                                      coords2, reinterpret_cast<long long*>(perms), static_cast<hipStream_t>(stream)));
}

static int device_threads_per_cu()
{
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 2048;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMaxThreadsPerMultiProcessor, dev) != hipSuccess || n <= 0) n = 2048;
    return n;
}

int stego_ref_draws(uint64_t seed, uint64_t offset, int32_t variant, int64_t n_coord, int32_t n_neg, int32_t B, float* coords1,
                    float* coords2, int64_t* perms, stego_stream_t stream)
{
    (void)hipGetLastError();
    if (n_coord < 0 || n_neg < 0 || B < 1 || B > 2048 || variant < 0 || variant > 7 || (offset & 3)) return STEGO_ERR_SHAPE;
    if ((n_coord > 0 && (!coords1 || !coords2)) || (n_neg > 0 && !perms)) return STEGO_ERR_NULL;
    return hip_rc(launch_ref_draws(seed, offset, nullptr, nullptr, variant, n_coord, n_neg, B, device_cu_count(), device_threads_per_cu(),
                                   coords1, coords2, reinterpret_cast<long long*>(perms), static_cast<hipStream_t>(stream)));
}

int stego_ref_draws_indirect(const int64_t* seed_ptr, const int64_t* offset_ptr, uint64_t offset_intragraph, int32_t variant,
                             int64_t n_coord, int32_t n_neg, int32_t B, float* coords1, float* coords2, int64_t* perms,
                             stego_stream_t stream)
{
    (void)hipGetLastError();
    if (n_coord < 0 || n_neg < 0 || B < 1 || B > 2048 || variant < 0 || variant > 7 || (offset_intragraph & 3)) return STEGO_ERR_SHAPE;
    if (!seed_ptr || !offset_ptr || (n_coord > 0 && (!coords1 || !coords2)) || (n_neg > 0 && !perms)) return STEGO_ERR_NULL;
    return hip_rc(launch_ref_draws(0, offset_intragraph, reinterpret_cast<const long long*>(seed_ptr),
                                   reinterpret_cast<const long long*>(offset_ptr), variant, n_coord, n_neg, B, device_cu_count(),
                                   device_threads_per_cu(), coords1, coords2, reinterpret_cast<long long*>(perms),
                                   static_cast<hipStream_t>(stream)));
}

int stego_ref_dropout_masks(uint64_t seed, uint64_t offset, const int64_t* seed_ptr, const int64_t* offset_ptr, int32_t variant,
                            int32_t n_masks, int64_t numel, float keep_prob, float* masks, stego_stream_t stream)
{
    (void)hipGetLastError();
    if (n_masks < 0 || n_masks > 16 || numel < 0 || variant < 0 || variant > 3 || (offset & 3) || !(keep_prob > 0.f) || keep_prob > 1.f)
        return STEGO_ERR_SHAPE;
    if ((n_masks > 0 && numel > 0 && !masks) || ((seed_ptr == nullptr) != (offset_ptr == nullptr))) return STEGO_ERR_NULL;
    return hip_rc(launch_ref_masks(seed, offset, reinterpret_cast<const long long*>(seed_ptr), reinterpret_cast<const long long*>(offset_ptr),
                                   variant, n_masks, numel, keep_prob, device_cu_count(), device_threads_per_cu(), masks,
                                   static_cast<hipStream_t>(stream)));
}

uint64_t stego_ref_dropout_masks_advance(int64_t numel, int32_t n_masks, int32_t variant)
{
    if (numel < 0 || n_masks < 0) return 0;
    return ref_masks_advance(numel, n_masks, variant, device_cu_count(), device_threads_per_cu());
}

uint64_t stego_ref_draws_advance(int64_t n_coord, int32_t n_neg, int32_t B, int32_t variant)
{
    if (n_coord < 0 || n_neg < 0 || B < 1) return 0;
    return ref_draws_advance(n_coord, n_neg, B, variant, device_cu_count(), device_threads_per_cu());
}

int stego_fast_draws(const int64_t* seed, int64_t n_coord, int32_t n_neg, int32_t B, float* coords1, float* coords2,
                     int64_t* perms, stego_stream_t stream)
{
    (void)hipGetLastError();
    if (n_coord < 0 || n_neg < 0 || B < 1 || B > 6000) return STEGO_ERR_SHAPE;          // (keys of a permutation live in LDS)
    if (!seed || (n_coord > 0 && (!coords1 || !coords2)) || (n_neg > 0 && !perms)) return STEGO_ERR_NULL;
    return hip_rc(launch_fast_draws(reinterpret_cast<const long long*>(seed), n_coord, n_neg, B, coords1, coords2,
                                    reinterpret_cast<long long*>(perms), static_cast<hipStream_t>(stream)));
}

int stego_corr_fwd_launches(const StegoCorrDesc* d, const StegoMap* feats, const StegoMap* feats_pos, const StegoMap* code,
                            const StegoMap* code_pos)
{
    int rc = check_desc(d, false);
    if (rc) return -rc;
    if (!feats || !feats_pos || !code || !code_pos) return -STEGO_ERR_NULL;
    if (is_wide(d)) return 5 + (d->n_neg > 0 ? 3 : 2);            // corr_wide.hip: 2 or 3 sampler launches, two correlations, three elementwise
    const Geometry g = geometry(d, false);
    FusedParams fp{};
    if ((rc = to_mapv(feats, d->C, d->H, d->W, &fp.feats)) || (rc = to_mapv(feats_pos, d->C, d->H, d->W, &fp.feats_pos)) ||
        (rc = to_mapv(code, d->K, d->H, d->W, &fp.code)) || (rc = to_mapv(code_pos, d->K, d->H, d->W, &fp.code_pos)))
        return -rc;
    fp.B = d->B; fp.C = d->C; fp.K = d->K; fp.H = d->H; fp.W = d->W; fp.S = d->S; fp.P = d->S * d->S; fp.n_neg = d->n_neg;
    const bool fused = knob(KNOB_FWD_VARIANT) != 1 && g.fs_bytes < ((size_t)1 << 31) && g.cs_bytes < ((size_t)1 << 31) &&
                       fused_supported(fp, d->precision);
    return fused ? 1 : 3;
}

int stego_corr_workspace_prepare(const StegoCorrDesc* d, void* workspace, size_t workspace_bytes, stego_stream_t stream)
{
    (void)hipGetLastError();
    int rc = check_desc(d, false);
    if (rc) return rc;
    if (!workspace) return STEGO_ERR_NULL;
    if (is_wide(d)) return workspace_bytes < wide_geom(d).ws_bytes ? STEGO_ERR_WORKSPACE : STEGO_OK;       // nothing to prepare
    const Geometry g = geometry(d, false);
    if (workspace_bytes < g.ws_bytes) return STEGO_ERR_WORKSPACE;
    // the hand-off words sit right behind the per-tile sums (plan_fwd carves the same way)
    return hip_rc(hipMemsetAsync(static_cast<unsigned char*>(workspace) + g.stats_bytes, 0, g.sync_bytes,
                                 static_cast<hipStream_t>(stream)));
}

int stego_corr_workspace_prepare_now(const StegoCorrDesc* d, void* workspace, size_t workspace_bytes)
{
    (void)hipGetLastError();
    hipStream_t side = nullptr;
    hipError_t e = side_begin(&side);
    if (e != hipSuccess) return hip_rc(e);
    if (!d && !workspace) return STEGO_OK;                 // first call of a host: creates the side stream outside any capture
    const int rc = stego_corr_workspace_prepare(d, workspace, workspace_bytes, static_cast<stego_stream_t>(side));
    if (rc) return rc;
    return hip_rc(side_finish());
}

int stego_corr_fwd_prepared(const StegoCorrDesc* d, const StegoMap* feats, const StegoMap* feats_pos, const StegoMap* code,
                            const StegoMap* code_pos, const float* coords1, const float* coords2, const int64_t* perms,
                            float* loss_means, float* pos_intra_cd, float* pos_inter_cd, float* neg_inter_loss,
                            float* neg_inter_cd, float* saved_w, float* saved_mean, void* saved_ctx, void* workspace,
                            size_t workspace_bytes, stego_stream_t stream)
{
    FwdPlan pl;
    int rc = check_desc(d, false);
    if (rc) return rc;
    if (is_wide(d)) return wide_fwd(d, feats, feats_pos, code, code_pos, coords1, coords2, perms, loss_means, pos_intra_cd, pos_inter_cd, neg_inter_loss,
                                    neg_inter_cd, saved_w, saved_mean, saved_ctx, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
    rc = plan_fwd(d, false, feats, feats_pos, code, code_pos, coords1, coords2, perms, loss_means, pos_intra_cd,
                  pos_inter_cd, neg_inter_loss, neg_inter_cd, saved_w, saved_mean, saved_ctx, workspace,
                  workspace_bytes, &pl);
    if (rc) return rc;
    return hip_rc(run_fwd(pl, static_cast<hipStream_t>(stream), nullptr, true));
}

int stego_corr_fwd_profile(const StegoCorrDesc* d, const StegoMap* feats, const StegoMap* feats_pos,
                           const StegoMap* code, const StegoMap* code_pos, const float* coords1, const float* coords2,
                           const int64_t* perms, float* loss_means, float* pos_intra_cd, float* pos_inter_cd,
                           float* neg_inter_loss, float* neg_inter_cd, float* saved_w, float* saved_mean,
                           void* saved_ctx, void* workspace, size_t workspace_bytes, stego_stream_t stream,
                           int32_t iters, float* ms_kernels /* [3] */)
{
    if (!ms_kernels || iters <= 0) return STEGO_ERR_NULL;
    if (d && check_desc(d, false) == STEGO_OK && is_wide(d)) {
        // the multi-launch path: the whole forward between two events, reported in the slot of the tile kernel
        hipStream_t s = static_cast<hipStream_t>(stream);
        hipEvent_t ev[2];
        hipError_t e;
        for (int i = 0; i < 2; ++i)
            if ((e = hipEventCreate(&ev[i])) != hipSuccess) return hip_rc(e);
        double acc = 0.0;
        int rc = STEGO_OK;
        for (int i = 0; i < iters && rc == STEGO_OK; ++i) {
            (void)hipEventRecord(ev[0], s);
            rc = wide_fwd(d, feats, feats_pos, code, code_pos, coords1, coords2, perms, loss_means, pos_intra_cd, pos_inter_cd, neg_inter_loss,
                          neg_inter_cd, saved_w, saved_mean, saved_ctx, workspace, workspace_bytes, s);
            (void)hipEventRecord(ev[1], s);
            if (rc == STEGO_OK) rc = hip_rc(hipEventSynchronize(ev[1]));
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, ev[0], ev[1]);
            acc += ms;
        }
        for (int i = 0; i < 2; ++i) (void)hipEventDestroy(ev[i]);
        ms_kernels[0] = 0.f; ms_kernels[1] = (float)(acc / iters); ms_kernels[2] = 0.f;
        return rc;
    }
    FwdPlan pl;
    int rc = plan_fwd(d, false, feats, feats_pos, code, code_pos, coords1, coords2, perms, loss_means, pos_intra_cd,
                      pos_inter_cd, neg_inter_loss, neg_inter_cd, saved_w, saved_mean, saved_ctx, workspace,
                      workspace_bytes, &pl);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipEvent_t ev[4];
    hipError_t e = hipSuccess;
    for (int i = 0; i < 4; ++i)
        if ((e = hipEventCreate(&ev[i])) != hipSuccess) return hip_rc(e);
    double acc[3] = {0.0, 0.0, 0.0};
    if (pl.use_fused) e = prepare_corr_fused(pl.fused, pl.sync_bytes, s);
    for (int i = 0; i < iters && e == hipSuccess; ++i) {
        e = run_fwd(pl, s, ev, true);
        if (e == hipSuccess) e = hipEventSynchronize(ev[3]);
        for (int k = 0; k < 3 && e == hipSuccess; ++k) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, ev[k], ev[k + 1]);
            acc[k] += ms;
        }
    }
    for (int i = 0; i < 4; ++i) (void)hipEventDestroy(ev[i]);
    for (int k = 0; k < 3; ++k) ms_kernels[k] = (float)(acc[k] / iters);
    return hip_rc(e);
}

int stego_corr_bwd(const StegoCorrDesc* d, const int64_t* perms, const float* saved_w, const float* saved_mean,
                   const void* saved_ctx, const float* pos_intra_cd, const float* pos_inter_cd,
                   const float* neg_inter_cd, const float* g_intra, const float* g_inter, const float* g_neg_loss,
                   int32_t g_neg_loss_stride, const float* g_intra_cd, const float* g_inter_cd, const float* g_neg_cd,
                   float* d_code, float* d_code_pos, void* workspace, size_t workspace_bytes, stego_stream_t stream)
{
    (void)hipGetLastError();
    int rc = check_desc(d, false);
    if (rc) return rc;
    if (!saved_w || !saved_mean || !pos_intra_cd || !pos_inter_cd || !d_code || !d_code_pos) return STEGO_ERR_NULL;
    if (d->n_neg > 0 && (!perms || !neg_inter_cd)) return STEGO_ERR_NULL;
    if (g_neg_loss_stride < -1 || g_neg_loss_stride > 1) return STEGO_ERR_SHAPE;
    if (is_wide(d)) {
        if (!saved_ctx || !workspace) return STEGO_ERR_NULL;
        if (workspace_bytes < wide_geom(d).bwd_ws_bytes) return STEGO_ERR_WORKSPACE;
        WideBwdArgs a{};
        a.perms = reinterpret_cast<const long long*>(perms);
        a.saved_w = saved_w; a.saved_mean = saved_mean; a.saved_ctx = saved_ctx;
        a.g_intra = g_intra; a.g_inter = g_inter; a.g_neg = g_neg_loss; a.g_neg_stride = g_neg_loss_stride;
        a.g_intra_cd = g_intra_cd; a.g_inter_cd = g_inter_cd; a.g_neg_cd = g_neg_cd;
        a.d_code = d_code; a.d_code_pos = d_code_pos; a.workspace = workspace;
        a.B = d->B; a.C = d->C; a.K = d->K; a.H = d->H; a.W = d->W; a.S = d->S; a.n_neg = d->n_neg;
        return hip_rc(launch_wide_bwd(a, static_cast<hipStream_t>(stream)));
    }
    BwdParams prm{};
    if ((rc = fill_bwd_ctx(d, false, saved_ctx, workspace, workspace_bytes, &prm))) return rc;
    prm.perms = reinterpret_cast<const long long*>(perms);
    prm.saved_w = saved_w; prm.saved_mean = saved_mean;
    prm.intra_cd = pos_intra_cd; prm.inter_cd = pos_inter_cd; prm.neg_cd = neg_inter_cd;
    prm.g_intra = g_intra; prm.g_inter = g_inter; prm.g_neg_loss = g_neg_loss;
    prm.g_neg_loss_stride = g_neg_loss_stride;
    prm.g_intra_cd = g_intra_cd; prm.g_inter_cd = g_inter_cd; prm.g_neg_cd = g_neg_cd;
    prm.d_code = d_code; prm.d_code_pos = d_code_pos;
    prm.B = d->B; prm.K = d->K; prm.H = d->H; prm.W = d->W; prm.S = d->S; prm.P = d->S * d->S;
    prm.n_neg = d->n_neg; prm.n_sets = 2 + d->n_neg; prm.mode = 0;
    prm.precision = d->precision;
    prm.debug = knob(KNOB_DEBUG_BWD);
    prm.cmin = d->zero_clamp ? 0.0f : -9999.0f;
    prm.cmax = d->stabalize ? 0.8f : std::numeric_limits<float>::infinity();
    return hip_rc(launch_corr_bwd(prm, static_cast<hipStream_t>(stream)));
}

int stego_corr_helper_fwd(const StegoCorrDesc* d, const StegoMap* f1, const StegoMap* f2, const StegoMap* c1,
                          const StegoMap* c2, float* loss, float* cd, float* saved_w, float* saved_mean,
                          void* saved_ctx, void* workspace, size_t workspace_bytes, stego_stream_t stream)
{
    FwdPlan pl;
    int rc = plan_fwd(d, true, f1, f2, c1, c2, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, loss, cd, saved_w,
                      saved_mean, saved_ctx, workspace, workspace_bytes, &pl);
    if (rc) return rc;
    return hip_rc(run_fwd(pl, static_cast<hipStream_t>(stream), nullptr));
}

int stego_corr_helper_bwd(const StegoCorrDesc* d, const float* saved_w, const float* saved_mean,
                          const void* saved_ctx, const float* cd, const float* g_loss, const float* g_cd, float* d_c1,
                          float* d_c2, void* workspace, size_t workspace_bytes, stego_stream_t stream)
{
    (void)hipGetLastError();
    int rc = check_desc(d, true);
    if (rc) return rc;
    if (!saved_w || !saved_mean || !cd || !d_c1 || !d_c2) return STEGO_ERR_NULL;
    BwdParams prm{};
    if ((rc = fill_bwd_ctx(d, true, saved_ctx, workspace, workspace_bytes, &prm))) return rc;
    prm.saved_w = saved_w; prm.saved_mean = saved_mean; prm.neg_cd = cd;
    prm.g_neg_loss = g_loss; prm.g_neg_loss_stride = 1; prm.g_neg_cd = g_cd;
    prm.d_code = d_c1; prm.d_code_pos = d_c2;
    prm.B = d->B; prm.K = d->K; prm.H = d->H; prm.W = d->W; prm.S = d->W; prm.P = d->H * d->W;
    prm.n_neg = 0; prm.n_sets = 1; prm.mode = 1;
    prm.debug = knob(KNOB_DEBUG_BWD);
    prm.cmin = d->zero_clamp ? 0.0f : -9999.0f;
    prm.cmax = d->stabalize ? 0.8f : std::numeric_limits<float>::infinity();
    return hip_rc(launch_corr_bwd(prm, static_cast<hipStream_t>(stream)));
}

static int knn_check(int64_t N, int32_t D, int32_t k, int64_t q_begin, int64_t q_count)
{
    if (N <= 0 || D <= 0 || q_count <= 0 || q_begin < 0) return STEGO_ERR_SHAPE;
    if (k < 1 || k > 32 || k > N || N >= ((int64_t)1 << 31)) return STEGO_ERR_UNSUPPORTED;
    if (q_begin % 128 != 0 || q_begin + q_count > N) return STEGO_ERR_SHAPE;
    return STEGO_OK;
}

size_t stego_knn_workspace_bytes(int64_t N, int32_t D, int32_t k, int64_t q_count)
{
    if (knn_check(N, D, k, 0, q_count) != STEGO_OK) return 0;
    return knn_workspace_bytes(N, D, k, q_count);
}

int stego_knn_topk(const float* X, int64_t N, int32_t D, int64_t ldx, int32_t k, int32_t normalize,
                   int64_t q_begin, int64_t q_count, int64_t* out_idx, float* out_sims,
                   void* workspace, size_t workspace_bytes, stego_stream_t stream)
{
    (void)hipGetLastError();
    int rc = knn_check(N, D, k, q_begin, q_count);
    if (rc) return rc;
    if (!X || !out_idx || !workspace) return STEGO_ERR_NULL;
    if (ldx < D) return STEGO_ERR_SHAPE;
    if ((reinterpret_cast<uintptr_t>(X) & 3) || (reinterpret_cast<uintptr_t>(out_idx) & 7)) return STEGO_ERR_ALIGN;
    if (workspace_bytes < knn_workspace_bytes(N, D, k, q_count)) return STEGO_ERR_WORKSPACE;
    unsigned char* w = static_cast<unsigned char*>(workspace);
    w += (256 - (reinterpret_cast<uintptr_t>(w) & 255)) & 255;                 // (the size includes this slack)
    return hip_rc(launch_knn(X, N, D, ldx, k, normalize ? 1 : 0, q_begin, q_count, reinterpret_cast<long long*>(out_idx),
                             out_sims, w, static_cast<hipStream_t>(stream)));
}

static int dense_check(int32_t B, int32_t C, int32_t H1, int32_t W1, int32_t H2, int32_t W2)
{
    if (B <= 0 || C <= 0 || H1 <= 0 || W1 <= 0 || H2 <= 0 || W2 <= 0) return STEGO_ERR_SHAPE;
    if ((int64_t)H1 * W1 > (1 << 24) || (int64_t)H2 * W2 > (1 << 24) || B > 65535) return STEGO_ERR_UNSUPPORTED;
    return STEGO_OK;
}

size_t stego_dense_corr_workspace_bytes(int32_t B, int32_t C, int32_t H1, int32_t W1, int32_t H2, int32_t W2)
{
    if (dense_check(B, C, H1, W1, H2, W2) != STEGO_OK) return 0;
    return dense_workspace_bytes(B, C, H1 * W1, H2 * W2);
}

int stego_dense_corr(const StegoMap* a, const StegoMap* b, int32_t B, int32_t C, int32_t H1, int32_t W1, int32_t H2,
                     int32_t W2, int32_t normalize, float* out, void* workspace, size_t workspace_bytes,
                     stego_stream_t stream)
{
    (void)hipGetLastError();
    int rc = dense_check(B, C, H1, W1, H2, W2);
    if (rc) return rc;
    if (!a || !b || !out || !workspace) return STEGO_ERR_NULL;
    if (workspace_bytes < dense_workspace_bytes(B, C, H1 * W1, H2 * W2)) return STEGO_ERR_WORKSPACE;
    MapV ma, mb;
    if ((rc = to_mapv(a, C, H1, W1, &ma))) return rc;
    if ((rc = to_mapv(b, C, H2, W2, &mb))) return rc;
    return hip_rc(launch_dense_corr(ma, mb, B, C, H1, W1, H2, W2, normalize ? 1 : 0, out, workspace,
                                    static_cast<hipStream_t>(stream)));
}

size_t stego_panel_image_bytes(int32_t C, int32_t P)
{
    if (C <= 0 || P <= 0 || P > (1 << 24)) return 0;
    return dense_panel_image_bytes(C, P);
}

int stego_dense_corr_panels(const void* panels_a, const float* row_scale_a, int32_t images_a, const void* panels_b, const float* row_scale_b,
                            int32_t N, int32_t C, int32_t M, int32_t Ncols, float* out, float* rowsum, stego_stream_t stream)
{
    (void)hipGetLastError();
    if (N <= 0 || C <= 0 || M <= 0 || Ncols <= 0 || images_a <= 0) return STEGO_ERR_SHAPE;
    if (M > (1 << 24) || Ncols > (1 << 24) || N > 65535) return STEGO_ERR_UNSUPPORTED;
    if (!panels_a || !row_scale_a || !panels_b || !row_scale_b || !out) return STEGO_ERR_NULL;
    if (reinterpret_cast<uintptr_t>(panels_a) % 16 != 0 || reinterpret_cast<uintptr_t>(panels_b) % 16 != 0) return STEGO_ERR_ALIGN;
    return hip_rc(launch_dense_corr_panels(panels_a, row_scale_a, images_a, panels_b, row_scale_b, N, C, M, Ncols, out, rowsum, static_cast<hipStream_t>(stream)));
}

}  // extern "C"
