// Host-side interface of corr_wide.hip (the loss for point sets of 129 .. 256 points per image) towards c_api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include "../../include/stego_corr.h"

namespace stego {

struct WideGeom {
    int n_sets, n_img, P, nb, Kr, Kc, nwin;
    size_t fimg, cimg;                                   // bytes of one operand image (features / codes)
    size_t o_fpan, o_frs, o_cpan, o_crs, o_rowsum, o_lrowsum, o_fd, o_mean, o_ctx, ws_bytes;        // forward workspace offsets
    size_t c_cn, c_inv, c_co1, c_co2, ctx_bytes;        // saved context offsets
    size_t b_rows, b_anchor, b_tiles, b_v, b_ent, b_off, bwd_ws_bytes;      // backward workspace offsets
    int tile_bytes;                                      // one transposed code tile (hi | lo), a multiple of 1 KB
};

struct WideFwdArgs {
    const StegoMap *feats, *feats_pos, *code, *code_pos;
    const float *coords1, *coords2;
    const long long* perms;
    float *loss_means, *intra_cd, *inter_cd, *neg_loss, *neg_cd, *saved_w, *saved_mean;
    void* saved_ctx;
    void* workspace;
    int B, C, K, H, W, S, n_neg, pointwise;
    float cmin, cmax, shift[3];
};

struct WideBwdArgs {
    const long long* perms;
    const float *saved_w, *saved_mean;
    const void* saved_ctx;
    const float *g_intra, *g_inter, *g_neg;
    int g_neg_stride;
    const float *g_intra_cd, *g_inter_cd, *g_neg_cd;
    float *d_code, *d_code_pos;                    // channels-last dense [B][H][W][K]
    void* workspace;
    int B, C, K, H, W, S, n_neg;
};

bool wide_supported(int B, int C, int K, int S, int n_neg);
WideGeom wide_geometry(int B, int C, int K, int H, int W, int S, int n_neg);
hipError_t launch_wide_fwd(const WideFwdArgs& a, hipStream_t stream);
hipError_t launch_wide_bwd(const WideBwdArgs& a, hipStream_t stream);

}  // namespace stego
