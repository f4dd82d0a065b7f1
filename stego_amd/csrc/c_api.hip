// extern "C" boundary of libstego_corr.so (declared in include/stego_corr.h).
// Host-side validation + parameter packing only; all arithmetic is in the HIP kernels.
#include <cstdlib>
#include <limits>

#include "../../include/stego_corr.h"
#include "corr_common.h"

namespace stego {
hipError_t launch_corr_fwd(const CorrParams& prm, int precision, int variant, hipStream_t stream);
hipError_t launch_corr_fwd_main(const CorrParams& prm, int precision, int variant, hipStream_t stream);
hipError_t launch_corr_finalize(const CorrParams& prm, hipStream_t stream);
hipError_t launch_corr_bwd(const BwdParams& prm, hipStream_t stream);
}  // namespace stego

using namespace stego;

namespace {

bool aligned4(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 3) == 0; }

// largest element offset inside one image must fit in int32 (the kernels use 32-bit tap offsets)
int to_mapv(const StegoMap* m, int channels, int H, int W, MapV* out)
{
    if (!m || !m->data) return STEGO_ERR_NULL;
    if (!aligned4(m->data)) return STEGO_ERR_ALIGN;
    const int64_t lim = std::numeric_limits<int32_t>::max();
    auto absv = [](int64_t v) { return v < 0 ? -v : v; };
    if (m->stride_c < 0 || m->stride_h < 0 || m->stride_w < 0) return STEGO_ERR_UNSUPPORTED;
    const int64_t span = absv(m->stride_c) * (channels - 1) + absv(m->stride_h) * (H - 1) + absv(m->stride_w) * (W - 1);
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
    if (helper) {
        if ((int64_t)d->H * d->W > TP) return STEGO_ERR_UNSUPPORTED;
    } else {
        if (d->S <= 0 || d->n_neg < 0) return STEGO_ERR_SHAPE;
        if (d->S * d->S > TP) return STEGO_ERR_UNSUPPORTED;
        if (d->n_neg + 2 > 256) return STEGO_ERR_UNSUPPORTED;
    }
    if (d->precision != STEGO_PREC_F32 && d->precision != STEGO_PREC_BF16X3) return STEGO_ERR_UNSUPPORTED;
    return STEGO_OK;
}

size_t ws_bytes(const StegoCorrDesc* d) { return (size_t)(2 + (d->n_neg > 0 ? d->n_neg : 0)) * d->B * 4 * sizeof(float); }

int hip_rc(hipError_t e) { return e == hipSuccess ? STEGO_OK : STEGO_ERR_HIP + (int)e; }

// Measurement knobs (read per call so a bench can flip them): STEGO_DEBUG = ablation bit mask
// (see CorrParams::debug / BwdParams::debug), STEGO_FWD_VARIANT = 0 simple kernel, 1 warp-specialised.
int env_int(const char* name, int dflt)
{
    const char* v = std::getenv(name);
    return v && *v ? std::atoi(v) : dflt;
}

}  // namespace

extern "C" {

int stego_abi_version(void) { return STEGO_ABI_VERSION; }

const char* stego_error_string(int code)
{
    switch (code) {
        case STEGO_OK: return "ok";
        case STEGO_ERR_NULL: return "required pointer is NULL";
        case STEGO_ERR_SHAPE: return "bad or inconsistent dimension";
        case STEGO_ERR_UNSUPPORTED: return "unsupported configuration (limits: S*S<=128, K<=80 in backward, fp32 maps, <2^31 elements per image)";
        case STEGO_ERR_WORKSPACE: return "workspace too small";
        case STEGO_ERR_ALIGN: return "pointer not 4-byte aligned";
        default: return code >= STEGO_ERR_HIP ? "HIP runtime error (code - 1000 = hipError_t)" : "unknown error";
    }
}

size_t stego_corr_workspace_bytes(const StegoCorrDesc* desc)
{
    if (!desc || desc->B <= 0) return 0;
    return ws_bytes(desc);
}

static int pack_fwd(const StegoCorrDesc* d, const StegoMap* feats, const StegoMap* feats_pos, const StegoMap* code,
                    const StegoMap* code_pos, const float* coords1, const float* coords2, const int64_t* perms,
                    float* loss_means, float* pos_intra_cd, float* pos_inter_cd, float* neg_inter_loss,
                    float* neg_inter_cd, float* saved_w, float* saved_mean, void* workspace, size_t workspace_bytes,
                    CorrParams* out)
{
    int rc = check_desc(d, false);
    if (rc) return rc;
    if (!coords1 || !coords2 || !loss_means || !pos_intra_cd || !pos_inter_cd || !workspace) return STEGO_ERR_NULL;
    if (d->n_neg > 0 && (!perms || !neg_inter_loss || !neg_inter_cd)) return STEGO_ERR_NULL;
    if ((saved_w == nullptr) != (saved_mean == nullptr)) return STEGO_ERR_NULL;
    if (workspace_bytes < ws_bytes(d)) return STEGO_ERR_WORKSPACE;
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
    prm.stats = static_cast<float*>(workspace);
    prm.B = d->B; prm.C = d->C; prm.K = d->K; prm.H = d->H; prm.W = d->W; prm.S = d->S; prm.P = d->S * d->S;
    prm.n_neg = d->n_neg; prm.n_sets = 2 + d->n_neg;
    prm.mode = 0; prm.pointwise = d->pointwise ? 1 : 0;
    prm.cmin = d->zero_clamp ? 0.0f : -9999.0f;                                   // modules.py:337-340
    prm.cmax = d->stabalize ? 0.8f : std::numeric_limits<float>::infinity();      // modules.py:342-345
    prm.shift[0] = d->pos_intra_shift; prm.shift[1] = d->pos_inter_shift; prm.shift[2] = d->neg_inter_shift;
    prm.debug = env_int("STEGO_DEBUG", 0);
    *out = prm;
    return STEGO_OK;
}

int stego_corr_fwd(const StegoCorrDesc* d, const StegoMap* feats, const StegoMap* feats_pos, const StegoMap* code,
                   const StegoMap* code_pos, const float* coords1, const float* coords2, const int64_t* perms,
                   float* loss_means, float* pos_intra_cd, float* pos_inter_cd, float* neg_inter_loss,
                   float* neg_inter_cd, float* saved_w, float* saved_mean, void* workspace, size_t workspace_bytes,
                   stego_stream_t stream)
{
    CorrParams prm{};
    int rc = pack_fwd(d, feats, feats_pos, code, code_pos, coords1, coords2, perms, loss_means, pos_intra_cd,
                      pos_inter_cd, neg_inter_loss, neg_inter_cd, saved_w, saved_mean, workspace, workspace_bytes, &prm);
    if (rc) return rc;
    return hip_rc(launch_corr_fwd(prm, d->precision, env_int("STEGO_FWD_VARIANT", 1), static_cast<hipStream_t>(stream)));
}

int stego_corr_fwd_profile(const StegoCorrDesc* d, const StegoMap* feats, const StegoMap* feats_pos,
                           const StegoMap* code, const StegoMap* code_pos, const float* coords1, const float* coords2,
                           const int64_t* perms, float* loss_means, float* pos_intra_cd, float* pos_inter_cd,
                           float* neg_inter_loss, float* neg_inter_cd, float* saved_w, float* saved_mean,
                           void* workspace, size_t workspace_bytes, stego_stream_t stream, int32_t iters,
                           float* ms_main, float* ms_finalize)
{
    if (!ms_main || !ms_finalize || iters <= 0) return STEGO_ERR_NULL;
    CorrParams prm{};
    int rc = pack_fwd(d, feats, feats_pos, code, code_pos, coords1, coords2, perms, loss_means, pos_intra_cd,
                      pos_inter_cd, neg_inter_loss, neg_inter_cd, saved_w, saved_mean, workspace, workspace_bytes, &prm);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipEvent_t e0, e1, e2;
    hipError_t e;
    if ((e = hipEventCreate(&e0)) != hipSuccess) return hip_rc(e);
    if ((e = hipEventCreate(&e1)) != hipSuccess) return hip_rc(e);
    if ((e = hipEventCreate(&e2)) != hipSuccess) return hip_rc(e);
    double tm = 0.0, tf = 0.0;
    for (int i = 0; i < iters && e == hipSuccess; ++i) {
        hipEventRecord(e0, s);
        e = launch_corr_fwd_main(prm, d->precision, env_int("STEGO_FWD_VARIANT", 1), s);
        hipEventRecord(e1, s);
        if (e == hipSuccess) e = launch_corr_finalize(prm, s);
        hipEventRecord(e2, s);
        if (e == hipSuccess) e = hipEventSynchronize(e2);
        float a = 0.f, b = 0.f;
        hipEventElapsedTime(&a, e0, e1);
        hipEventElapsedTime(&b, e1, e2);
        tm += a; tf += b;
    }
    hipEventDestroy(e0); hipEventDestroy(e1); hipEventDestroy(e2);
    *ms_main = (float)(tm / iters);
    *ms_finalize = (float)(tf / iters);
    return hip_rc(e);
}

int stego_corr_bwd(const StegoCorrDesc* d, const StegoMap* code, const StegoMap* code_pos, const float* coords1,
                   const float* coords2, const int64_t* perms, const float* saved_w, const float* saved_mean,
                   const float* pos_intra_cd, const float* pos_inter_cd, const float* neg_inter_cd,
                   const float* g_intra, const float* g_inter, const float* g_neg_loss, int32_t g_neg_loss_stride,
                   const float* g_intra_cd, const float* g_inter_cd, const float* g_neg_cd, float* d_code,
                   float* d_code_pos, void* workspace, size_t workspace_bytes, stego_stream_t stream)
{
    (void)workspace; (void)workspace_bytes;
    int rc = check_desc(d, false);
    if (rc) return rc;
    if (d->K > 80) return STEGO_ERR_UNSUPPORTED;
    if (!coords1 || !coords2 || !saved_w || !saved_mean || !pos_intra_cd || !pos_inter_cd || !d_code || !d_code_pos)
        return STEGO_ERR_NULL;
    if (d->n_neg > 0 && (!perms || !neg_inter_cd)) return STEGO_ERR_NULL;
    if (g_neg_loss_stride != 0 && g_neg_loss_stride != 1) return STEGO_ERR_SHAPE;
    BwdParams prm{};
    if ((rc = to_mapv(code, d->K, d->H, d->W, &prm.code))) return rc;
    if ((rc = to_mapv(code_pos, d->K, d->H, d->W, &prm.code_pos))) return rc;
    prm.coords1 = coords1; prm.coords2 = coords2; prm.perms = reinterpret_cast<const long long*>(perms);
    prm.saved_w = saved_w; prm.saved_mean = saved_mean;
    prm.intra_cd = pos_intra_cd; prm.inter_cd = pos_inter_cd; prm.neg_cd = neg_inter_cd;
    prm.g_intra = g_intra; prm.g_inter = g_inter; prm.g_neg_loss = g_neg_loss;
    prm.g_neg_loss_stride = g_neg_loss_stride;
    prm.g_intra_cd = g_intra_cd; prm.g_inter_cd = g_inter_cd; prm.g_neg_cd = g_neg_cd;
    prm.d_code = d_code; prm.d_code_pos = d_code_pos;
    prm.B = d->B; prm.K = d->K; prm.H = d->H; prm.W = d->W; prm.S = d->S; prm.P = d->S * d->S;
    prm.n_neg = d->n_neg; prm.n_sets = 2 + d->n_neg; prm.mode = 0;
    prm.debug = env_int("STEGO_DEBUG_BWD", 0);
    prm.cmin = d->zero_clamp ? 0.0f : -9999.0f;
    prm.cmax = d->stabalize ? 0.8f : std::numeric_limits<float>::infinity();
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t bytes = (size_t)d->B * d->H * d->W * d->K * sizeof(float);
    hipError_t e = hipMemsetAsync(d_code, 0, bytes, s);
    if (e != hipSuccess) return hip_rc(e);
    e = hipMemsetAsync(d_code_pos, 0, bytes, s);
    if (e != hipSuccess) return hip_rc(e);
    return hip_rc(launch_corr_bwd(prm, s));
}

int stego_corr_helper_fwd(const StegoCorrDesc* d, const StegoMap* f1, const StegoMap* f2, const StegoMap* c1,
                          const StegoMap* c2, float* loss, float* cd, float* saved_w, float* saved_mean,
                          void* workspace, size_t workspace_bytes, stego_stream_t stream)
{
    int rc = check_desc(d, true);
    if (rc) return rc;
    if (!loss || !cd || !workspace) return STEGO_ERR_NULL;
    if ((saved_w == nullptr) != (saved_mean == nullptr)) return STEGO_ERR_NULL;
    if (workspace_bytes < (size_t)d->B * 4 * sizeof(float)) return STEGO_ERR_WORKSPACE;
    CorrParams prm{};
    if ((rc = to_mapv(f1, d->C, d->H, d->W, &prm.feats))) return rc;
    if ((rc = to_mapv(f2, d->C, d->H, d->W, &prm.feats_pos))) return rc;
    if ((rc = to_mapv(c1, d->K, d->H, d->W, &prm.code))) return rc;
    if ((rc = to_mapv(c2, d->K, d->H, d->W, &prm.code_pos))) return rc;
    prm.neg_loss = loss; prm.neg_cd = cd; prm.saved_w = saved_w; prm.saved_mean = saved_mean;
    prm.stats = static_cast<float*>(workspace);
    prm.B = d->B; prm.C = d->C; prm.K = d->K; prm.H = d->H; prm.W = d->W; prm.S = d->W; prm.P = d->H * d->W;
    prm.n_neg = 0; prm.n_sets = 1; prm.mode = 1; prm.pointwise = d->pointwise ? 1 : 0;
    prm.cmin = d->zero_clamp ? 0.0f : -9999.0f;
    prm.cmax = d->stabalize ? 0.8f : std::numeric_limits<float>::infinity();
    prm.shift[0] = prm.shift[1] = prm.shift[2] = d->pos_intra_shift;
    prm.debug = env_int("STEGO_DEBUG", 0);
    return hip_rc(launch_corr_fwd(prm, d->precision, env_int("STEGO_FWD_VARIANT", 1), static_cast<hipStream_t>(stream)));
}

int stego_corr_helper_bwd(const StegoCorrDesc* d, const StegoMap* c1, const StegoMap* c2, const float* saved_w,
                          const float* saved_mean, const float* cd, const float* g_loss, const float* g_cd,
                          float* d_c1, float* d_c2, void* workspace, size_t workspace_bytes, stego_stream_t stream)
{
    (void)workspace; (void)workspace_bytes;
    int rc = check_desc(d, true);
    if (rc) return rc;
    if (d->K > 80) return STEGO_ERR_UNSUPPORTED;
    if (!saved_w || !saved_mean || !cd || !d_c1 || !d_c2) return STEGO_ERR_NULL;
    BwdParams prm{};
    if ((rc = to_mapv(c1, d->K, d->H, d->W, &prm.code))) return rc;
    if ((rc = to_mapv(c2, d->K, d->H, d->W, &prm.code_pos))) return rc;
    prm.saved_w = saved_w; prm.saved_mean = saved_mean; prm.neg_cd = cd;
    prm.g_neg_loss = g_loss; prm.g_neg_loss_stride = 1; prm.g_neg_cd = g_cd;
    prm.d_code = d_c1; prm.d_code_pos = d_c2;
    prm.B = d->B; prm.K = d->K; prm.H = d->H; prm.W = d->W; prm.S = d->W; prm.P = d->H * d->W;
    prm.n_neg = 0; prm.n_sets = 1; prm.mode = 1;
    prm.cmin = d->zero_clamp ? 0.0f : -9999.0f;
    prm.cmax = d->stabalize ? 0.8f : std::numeric_limits<float>::infinity();
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t bytes = (size_t)d->B * d->H * d->W * d->K * sizeof(float);
    hipError_t e = hipMemsetAsync(d_c1, 0, bytes, s);
    if (e != hipSuccess) return hip_rc(e);
    e = hipMemsetAsync(d_c2, 0, bytes, s);
    if (e != hipSuccess) return hip_rc(e);
    return hip_rc(launch_corr_bwd(prm, s));
}

}  // extern "C"
