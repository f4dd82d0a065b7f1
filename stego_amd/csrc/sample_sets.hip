// sample() with an image index - the bilinear point sampler of the loss for shapes the fused kernels do not take (gfx950).
//
//   reference: sample(t, coords) src/modules.py:287-288 = F.grid_sample(t, coords.permute(0, 2, 1, 3), padding_mode='border',
//              align_corners=True): out[n, :, h, w] is t[n] blended at coords[n, w, h, :] ([..., 0] = x, [..., 1] = y).
//              ContrastiveCorrelationLoss.forward calls it on orig_feats[perm] / orig_code[perm] (:384-385): a full copy of the
//              permuted maps per negative set (190 MB + 35 MB at BASELINE config 2 - SURVEY.md 8 a3: 44 % of the reference's time)
//              before 121..256 points of each are read.
//
// stego_sample gathers straight from map[index[n]] (index may be null: n itself) with the coordinates of row n % n_coords, and
// writes channels-last rows [n][point][C]; stego_sample_bwd is its adjoint (atomic fp32 adds into a zeroed map: the order of the
// adds - hence the last bits - is not fixed, as torch's grid_sampler backward / index_put(accumulate) are not).  One wave per
// (n, point): 16-byte loads along the channels of a channels-last map, scalar loads for any other layout.  Used by
// stego_amd.modules.ContrastiveCorrelationLoss.generic_forward (cfg.feature_samples > 11, cfg.dim > 128).
#include "corr_common.h"
#include "host_util.h"
#include "../../include/stego_corr.h"

namespace stego {

struct MapL {                     // (64-bit strides: any view torch can make)
    const float* p;
    long long sn, sc, sh, sw;
};

struct GatherParams {
    MapL map;
    const long long* index;     // [N] image of map per output row, or null
    const float* coords;        // [n_coords][S][S][2]
    float* io;                  // forward: out [N][P][C]; backward: g_out [N][P][C] (read)
    float* dmap_p;              // backward: the map's gradient (same strides as map)
    const float* rows_n;        // backward, optional: the normalised rows [N][P][C] and 1 / max(|row|, eps) [N][P]: io is the gradient of the
    const float* inv;           //   NORMALISED rows, the backward of norm() is applied before the scatter
    const float* extra;         // backward, optional: n_extra more gradients of the same rows, extra_stride floats apart, summed in (fixed order)
    long long extra_stride;
    int n_extra;
    int N, C, H, W, S, P, n_coords;
};

__device__ __forceinline__ void point_of(const GatherParams& p, int gw, int& n, int& q, long long& img, int4& yx, float4& w)
{
    n = gw / p.P;
    q = gw - n * p.P;
    img = p.index ? p.index[n] : (long long)n;
    const int hh = q / p.S, ww = q - hh * p.S;          // point (h, w) reads coords[w][h] (the permute of modules.py:288)
    const int nc = p.n_coords < 0 ? -p.n_coords : p.n_coords;
    const float* c = p.coords + ((size_t)(n % nc) * p.P + (size_t)ww * p.S + hh) * 2;
    make_taps(c[0], c[1], p.H, p.W, yx, w);
}

template <bool VEC>
__global__ void __launch_bounds__(256) sample_gather_kernel(const GatherParams p)
{
    const int lane = threadIdx.x & 63;
    const int gw = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));       // wave-uniform: index, coordinates and taps are scalar work
    if (gw >= p.N * p.P) return;
    int n, q;
    long long img;
    int4 yx;
    float4 w;
    point_of(p, gw, n, q, img, yx, w);
    const float* base = p.map.p + img * p.map.sn;
    const long long o0 = (long long)(yx.x >> 16) * p.map.sh + (long long)(yx.x & 0xffff) * p.map.sw;
    const long long o1 = (long long)(yx.y >> 16) * p.map.sh + (long long)(yx.y & 0xffff) * p.map.sw;
    const long long o2 = (long long)(yx.z >> 16) * p.map.sh + (long long)(yx.z & 0xffff) * p.map.sw;
    const long long o3 = (long long)(yx.w >> 16) * p.map.sh + (long long)(yx.w & 0xffff) * p.map.sw;
    float* out = p.io + (size_t)gw * p.C;
    if constexpr (VEC) {
        for (int c = 4 * lane; c < p.C; c += 256) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(base + o0 + c), b = *reinterpret_cast<const f32x4*>(base + o1 + c);
            const f32x4 cc = *reinterpret_cast<const f32x4*>(base + o2 + c), d = *reinterpret_cast<const f32x4*>(base + o3 + c);
            f32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = __builtin_fmaf(w.w, d[e], __builtin_fmaf(w.z, cc[e], __builtin_fmaf(w.y, b[e], w.x * a[e])));
            *reinterpret_cast<f32x4*>(out + c) = r;
        }
    } else {
        for (int c = lane; c < p.C; c += 64) {
            const long long oc = (long long)c * p.map.sc;
            out[c] = __builtin_fmaf(w.w, base[o3 + oc], __builtin_fmaf(w.z, base[o2 + oc], __builtin_fmaf(w.y, base[o1 + oc], w.x * base[o0 + oc])));
        }
    }
}

__global__ void __launch_bounds__(256) sample_scatter_kernel(const GatherParams p)
{
    const int lane = threadIdx.x & 63;
    const int gw = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));       // wave-uniform: index, coordinates and taps are scalar work
    if (gw >= p.N * p.P) return;
    int n, q;
    long long img;
    int4 yx;
    float4 w;
    point_of(p, gw, n, q, img, yx, w);
    float* base = p.dmap_p + img * p.map.sn;
    const long long o0 = (long long)(yx.x >> 16) * p.map.sh + (long long)(yx.x & 0xffff) * p.map.sw;
    const long long o1 = (long long)(yx.y >> 16) * p.map.sh + (long long)(yx.y & 0xffff) * p.map.sw;
    const long long o2 = (long long)(yx.z >> 16) * p.map.sh + (long long)(yx.z & 0xffff) * p.map.sw;
    const long long o3 = (long long)(yx.w >> 16) * p.map.sh + (long long)(yx.w & 0xffff) * p.map.sw;
    const float* g = p.io + (size_t)gw * p.C;
    const float* gx = p.n_extra > 0 ? p.extra + (size_t)gw * p.C : nullptr;
    auto grad = [&](int c) {
        float v = g[c];
        for (int t = 0; t < p.n_extra; ++t) v += gx[(size_t)t * p.extra_stride + c];
        return v;
    };
    // norm() backward (F.normalize, modules.py:275-276): y = x / max(|x|, eps) -> dx = (g - y <y, g>) / |x|, or g / eps below eps
    const float* yn = p.rows_n ? p.rows_n + (size_t)gw * p.C : nullptr;
    float invn = 1.f, proj = 0.f;
    if (yn) {
        invn = p.inv[gw];
        for (int c = lane; c < p.C; c += 64) proj = __builtin_fmaf(yn[c], grad(c), proj);
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) proj += __shfl_xor(proj, m, 64);
        if (invn > 0.99e10f) proj = 0.f;
    }
    for (int c = lane; c < p.C; c += 64) {
        const long long oc = (long long)c * p.map.sc;
        const float gc = grad(c);
        const float v = yn ? invn * (gc - yn[c] * proj) : gc;
        if (p.n_coords < 0) {          // (tools: plain stores instead of atomics - wrong sums, the cost of the atomics by difference)
            base[o0 + oc] = w.x * v; base[o1 + oc] = w.y * v; base[o2 + oc] = w.z * v; base[o3 + oc] = w.w * v;
            continue;
        }
        unsafeAtomicAdd(base + o0 + oc, w.x * v);
        unsafeAtomicAdd(base + o1 + oc, w.y * v);
        unsafeAtomicAdd(base + o2 + oc, w.z * v);
        unsafeAtomicAdd(base + o3 + oc, w.w * v);
    }
}

// ---- round 5: the sampled points as OPERANDS of the dense-correspondence kernels.  One wave per (n, point): blend the four taps, L2-normalise
// the row (norm(), modules.py:275-276), give it the power-of-two scale that puts its largest magnitude into [0.5, 1) and write it as split-fp16
// (hi | lo) into the operand image dense_prep_kernel would have made of the fp32 rows - [n][128-point block][64-channel chunk][hi | lo][128][72]
// (csrc/dense_corr.hip) - with 1 / scale per row beside it.  For tensors that carry a gradient (the codes) the normalised fp32 rows and
// 1 / max(|row|, eps) are written too: what the backward of norm() and of the correlation needs.  The fp32 rows of the FEATURES never exist.
struct PanelSide {
    MapL map;
    int C, NCH;
    half_t* panels;
    float* row_scale;           // [N][nb * 128]
    float* rows_out;            // optional [N][P][C]
    float* inv_out;             // optional [N][P]
};

struct PanelParams {
    GatherParams g;             // (map / io / dmap_p unused: the maps are in `side`)
    PanelSide side[2];          // the loss samples two maps at the same points - features and codes: one launch, the second map's taps in the
    int nb, normalize;          // same round trip (side[1] unused by the one-map instantiations)
};

// One row of one map.  VW = channels per lane and load: 4 (16-byte loads; C % 4 == 0, C <= 1024), 2 (8-byte loads; C % 2 == 0, C <= 512: the
// reference's dim = 70), 0 = any layout, scalar loads, two passes over the row.
template <int VW>
struct RowSampler {
    static constexpr int MAXJ = 4, VWN = VW > 0 ? VW : 1;
    typedef float vec_t __attribute__((ext_vector_type(VWN)));
    vec_t v[MAXJ];
    float ss, mx;
    const float* base;
    long long o0, o1, o2, o3;

    __device__ __forceinline__ float blend1(const float4& w, long long oc) const
    {
        return __builtin_fmaf(w.w, base[o3 + oc], __builtin_fmaf(w.z, base[o2 + oc], __builtin_fmaf(w.y, base[o1 + oc], w.x * base[o0 + oc])));
    }
    __device__ __forceinline__ void load(const PanelSide& sd, long long img, const int4& yx, const float4& w, int lane)
    {
        base = sd.map.p + img * sd.map.sn;
        o0 = (long long)(yx.x >> 16) * sd.map.sh + (long long)(yx.x & 0xffff) * sd.map.sw;
        o1 = (long long)(yx.y >> 16) * sd.map.sh + (long long)(yx.y & 0xffff) * sd.map.sw;
        o2 = (long long)(yx.z >> 16) * sd.map.sh + (long long)(yx.z & 0xffff) * sd.map.sw;
        o3 = (long long)(yx.w >> 16) * sd.map.sh + (long long)(yx.w & 0xffff) * sd.map.sw;
        ss = 0.f;
        mx = 0.f;
        if constexpr (VW > 0) {
#pragma unroll
            for (int j = 0; j < MAXJ; ++j) {
                const int c = VW * lane + 64 * VW * j;
#pragma unroll
                for (int e = 0; e < VW; ++e) v[j][e] = 0.f;
                if (c < sd.C) {
                    const vec_t a = *reinterpret_cast<const vec_t*>(base + o0 + c), b = *reinterpret_cast<const vec_t*>(base + o1 + c);
                    const vec_t cc = *reinterpret_cast<const vec_t*>(base + o2 + c), d = *reinterpret_cast<const vec_t*>(base + o3 + c);
#pragma unroll
                    for (int e = 0; e < VW; ++e) {
                        v[j][e] = __builtin_fmaf(w.w, d[e], __builtin_fmaf(w.z, cc[e], __builtin_fmaf(w.y, b[e], w.x * a[e])));
                        ss += v[j][e] * v[j][e];
                        mx = fmaxf(mx, fabsf(v[j][e]));
                    }
                }
            }
        } else {
            for (int c = lane; c < sd.C; c += 64) {
                const float t = blend1(w, (long long)c * sd.map.sc);
                ss += t * t;
                mx = fmaxf(mx, fabsf(t));
            }
        }
    }
    __device__ __forceinline__ void finish(const PanelSide& sd, const PanelParams& pp, int n, int blk, int rl, int gw, const float4& w, int lane)
    {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { ss += __shfl_xor(ss, m, 64); mx = fmaxf(mx, __shfl_xor(mx, m, 64)); }
        const float invn = (pp.normalize & 1) ? 1.f / fmaxf(sqrtf(ss), 1e-10f) : 1.f;       // norm(), modules.py:276
        const float rs = mx * invn > 0.f ? __builtin_ldexpf(1.f, -__builtin_amdgcn_frexp_expf(mx * invn)) : 1.f;
        const float inv = invn * rs;
        if (lane == 0) {
            sd.row_scale[((size_t)n * pp.nb + blk) * TP + rl] = 1.f / rs;
            if (sd.inv_out) sd.inv_out[gw] = invn;
        }
        if (pp.normalize & 2) return;                 // (tools: no row stores - what the stores cost)
        const int CP = sd.NCH * KC;
        half_t* dst = sd.panels + ((size_t)n * pp.nb + blk) * sd.NCH * (2 * TP * LDH) + rl * LDH;
        float* rows = sd.rows_out ? sd.rows_out + (size_t)gw * sd.C : nullptr;
        if constexpr (VW > 0) {
#pragma unroll
            for (int j = 0; j < MAXJ; ++j) {
                const int c = VW * lane + 64 * VW * j;
                if (c < CP) {                                                                // (zeros up to the end of the last chunk)
                    half_t* dh = dst + (size_t)(c >> 6) * (2 * TP * LDH) + (c & 63);
                    unsigned h0, l0;
                    split_f16_pair(v[j][0] * inv, v[j][1] * inv, h0, l0);
                    if constexpr (VW == 4) {
                        unsigned h1, l1;
                        split_f16_pair(v[j][2] * inv, v[j][3] * inv, h1, l1);
                        *reinterpret_cast<u32x2*>(dh) = u32x2{h0, h1};
                        *reinterpret_cast<u32x2*>(dh + TP * LDH) = u32x2{l0, l1};
                    } else {
                        *reinterpret_cast<unsigned*>(dh) = h0;
                        *reinterpret_cast<unsigned*>(dh + TP * LDH) = l0;
                    }
                    if (rows && c < sd.C) *reinterpret_cast<vec_t*>(rows + c) = v[j] * invn;
                }
            }
        } else {
            for (int c = lane; c < CP; c += 64) {
                const float t = c < sd.C ? blend1(w, (long long)c * sd.map.sc) : 0.f;
                unsigned h, l;
                split_f16_pair(t * inv, 0.f, h, l);
                half_t* dh = dst + (size_t)(c >> 6) * (2 * TP * LDH) + (c & 63);
                *reinterpret_cast<unsigned short*>(dh) = (unsigned short)(h & 0xffffu);
                *reinterpret_cast<unsigned short*>(dh + TP * LDH) = (unsigned short)(l & 0xffffu);
                if (rows && c < sd.C) rows[c] = t * invn;
            }
        }
    }
};

// VW2 < 0: one map
template <int VW1, int VW2>
__global__ void __launch_bounds__(256) sample_panels_kernel(const PanelParams pp)
{
    const GatherParams& p = pp.g;
    const int lane = threadIdx.x & 63;
    const int gw = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));       // wave-uniform: index, coordinates and taps are scalar work
    if (gw >= p.N * p.P) return;
    int n, q;
    long long img;
    int4 yx;
    float4 w;
    point_of(p, gw, n, q, img, yx, w);
    const int blk = q >> 7, rl = q & 127;
    RowSampler<VW1> a;
    RowSampler<(VW2 < 0 ? 0 : VW2)> b;
    a.load(pp.side[0], img, yx, w, lane);
    if constexpr (VW2 >= 0) b.load(pp.side[1], img, yx, w, lane);
    a.finish(pp.side[0], pp, n, blk, rl, gw, w, lane);
    if constexpr (VW2 >= 0) b.finish(pp.side[1], pp, n, blk, rl, gw, w, lane);
}

static int check_sample(const StegoMap* map, int32_t N, int32_t C, int32_t H, int32_t W, const float* coords, int32_t n_coords, int32_t S, const void* io)
{
    if (!map || !map->data || !coords || !io) return STEGO_ERR_NULL;
    if (N < 0 || C < 1 || H < 1 || W < 1 || S < 1 || n_coords < 1 || H > 32767 || W > 32767 || (long long)S * S > (1 << 20)) return STEGO_ERR_SHAPE;
    if ((long long)N * S * S >= (1ll << 31)) return STEGO_ERR_UNSUPPORTED;
    return STEGO_OK;
}

static GatherParams make_params(const StegoMap* map, const int64_t* index, int32_t N, int32_t C, int32_t H, int32_t W, const float* coords,
                                int32_t n_coords, int32_t S, float* io, float* dmap)
{
    GatherParams p;
    p.map = MapL{map->data, map->stride_n, map->stride_c, map->stride_h, map->stride_w};
    p.index = reinterpret_cast<const long long*>(index);
    p.coords = coords;
    p.io = io;
    p.dmap_p = dmap;
    p.rows_n = nullptr;
    p.inv = nullptr;
    p.extra = nullptr;
    p.extra_stride = 0;
    p.n_extra = 0;
    p.N = N; p.C = C; p.H = H; p.W = W; p.S = S; p.P = S * S; p.n_coords = n_coords;
    return p;
}

static int panel_vw(const StegoMap* map, int C, const float* rows_out)
{
    auto aligned = [&](int vw) {
        return map->stride_c == 1 && C % vw == 0 && C <= 256 * vw && map->stride_n % vw == 0 && map->stride_h % vw == 0 && map->stride_w % vw == 0 &&
               reinterpret_cast<uintptr_t>(map->data) % (4 * vw) == 0 && (!rows_out || reinterpret_cast<uintptr_t>(rows_out) % (4 * vw) == 0);
    };
    return aligned(4) ? 4 : aligned(2) ? 2 : 0;
}

static PanelSide panel_side(const StegoMap* map, int C, void* panels, float* row_scale, float* rows_out, float* inv_out)
{
    PanelSide sd;
    sd.map = MapL{map->data, map->stride_n, map->stride_c, map->stride_h, map->stride_w};
    sd.C = C;
    sd.NCH = (C + KC - 1) / KC;
    sd.panels = static_cast<half_t*>(panels);
    sd.row_scale = row_scale;
    sd.rows_out = rows_out;
    sd.inv_out = inv_out;
    return sd;
}

// one map (map2 == nullptr) or two maps of the same H x W sampled at the same points
hipError_t launch_sample_panels2(const StegoMap* map, int C, void* panels, float* row_scale, float* rows_out, float* inv_out,
                                 const StegoMap* map2, int C2, void* panels2, float* row_scale2, float* rows_out2, float* inv_out2,
                                 const long long* index, int N, int H, int W, const float* coords, int n_coords, int S, int normalize, hipStream_t stream)
{
    if (N == 0) return hipSuccess;
    PanelParams pp;
    pp.g = make_params(map, reinterpret_cast<const int64_t*>(index), N, C, H, W, coords, n_coords, S, nullptr, nullptr);
    pp.side[0] = panel_side(map, C, panels, row_scale, rows_out, inv_out);
    pp.side[1] = map2 ? panel_side(map2, C2, panels2, row_scale2, rows_out2, inv_out2) : pp.side[0];
    pp.nb = (pp.g.P + TP - 1) / TP;
    pp.normalize = (normalize ? 1 : 0) | ((knob(KNOB_DEBUG_SAMPLE) & (1 << 16)) ? 2 : 0);
    const int v1 = panel_vw(map, C, rows_out), v2 = map2 ? panel_vw(map2, C2, rows_out2) : -1;
    const dim3 grid((unsigned)(((long long)N * pp.g.P + 3) / 4)), block(256);
#define STEGO_SP_CASE(A, B) if (v1 == A && v2 == B) hipLaunchKernelGGL((sample_panels_kernel<A, B>), grid, block, 0, stream, pp)
    STEGO_SP_CASE(4, -1); else STEGO_SP_CASE(2, -1); else STEGO_SP_CASE(0, -1);
    else STEGO_SP_CASE(4, 4); else STEGO_SP_CASE(4, 2); else STEGO_SP_CASE(4, 0);
    else STEGO_SP_CASE(2, 4); else STEGO_SP_CASE(2, 2); else STEGO_SP_CASE(2, 0);
    else STEGO_SP_CASE(0, 4); else STEGO_SP_CASE(0, 2); else STEGO_SP_CASE(0, 0);
#undef STEGO_SP_CASE
    return hipGetLastError();
}

hipError_t launch_sample_panels(const StegoMap* map, const long long* index, int N, int C, int H, int W, const float* coords, int n_coords, int S,
                                int normalize, void* panels, float* row_scale, float* rows_out, float* inv_out, hipStream_t stream)
{
    return launch_sample_panels2(map, C, panels, row_scale, rows_out, inv_out, nullptr, 0, nullptr, nullptr, nullptr, nullptr, index, N, H, W, coords,
                                 n_coords, S, normalize, stream);
}

hipError_t launch_sample_scatter(const float* g_rows, const float* rows_n, const float* inv, const StegoMap* d_map, const long long* index, int N, int C,
                                 int H, int W, const float* coords, int n_coords, int S, const float* extra, int n_extra, long long extra_stride,
                                 hipStream_t stream)
{
    if (N == 0) return hipSuccess;
    GatherParams p = make_params(d_map, reinterpret_cast<const int64_t*>(index), N, C, H, W, coords, n_coords, S, const_cast<float*>(g_rows),
                                 const_cast<float*>(d_map->data));
    p.rows_n = rows_n;
    p.inv = inv;
    p.extra = extra;
    p.n_extra = extra ? n_extra : 0;
    p.extra_stride = extra_stride;
    if (knob(KNOB_DEBUG_BWD) & (1 << 19)) p.n_coords = -p.n_coords;
    const dim3 grid((unsigned)(((long long)N * p.P + 3) / 4)), block(256);
    hipLaunchKernelGGL(sample_scatter_kernel, grid, block, 0, stream, p);
    return hipGetLastError();
}

}  // namespace stego

using namespace stego;

extern "C" {

int stego_sample(const StegoMap* map, const int64_t* index, int32_t N, int32_t C, int32_t H, int32_t W, const float* coords,
                 int32_t n_coords, int32_t S, float* out, stego_stream_t stream)
{
    const int rc = check_sample(map, N, C, H, W, coords, n_coords, S, out);
    if (rc != STEGO_OK) return rc;
    if (N == 0) return STEGO_OK;
    const GatherParams p = make_params(map, index, N, C, H, W, coords, n_coords, S, out, nullptr);
    const bool vec = map->stride_c == 1 && C % 4 == 0 && map->stride_n % 4 == 0 && map->stride_h % 4 == 0 && map->stride_w % 4 == 0 &&
                     reinterpret_cast<uintptr_t>(map->data) % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0;
    const dim3 grid((unsigned)(((long long)N * p.P + 3) / 4)), block(256);
    if (vec) hipLaunchKernelGGL(sample_gather_kernel<true>, grid, block, 0, static_cast<hipStream_t>(stream), p);
    else hipLaunchKernelGGL(sample_gather_kernel<false>, grid, block, 0, static_cast<hipStream_t>(stream), p);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? STEGO_OK : STEGO_ERR_HIP + (int)e;
}

int stego_sample_panels(const StegoMap* map, const int64_t* index, int32_t N, int32_t C, int32_t H, int32_t W, const float* coords,
                        int32_t n_coords, int32_t S, int32_t normalize, void* panels, float* row_scale, float* rows_out, float* inv_out,
                        stego_stream_t stream)
{
    const int rc = check_sample(map, N, C, H, W, coords, n_coords, S, panels);
    if (rc != STEGO_OK) return rc;
    if (!row_scale) return STEGO_ERR_NULL;
    if (reinterpret_cast<uintptr_t>(panels) % 16 != 0) return STEGO_ERR_ALIGN;
    const hipError_t e = launch_sample_panels(map, reinterpret_cast<const long long*>(index), N, C, H, W, coords, n_coords, S, normalize, panels,
                                              row_scale, rows_out, inv_out, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? STEGO_OK : STEGO_ERR_HIP + (int)e;
}

int stego_sample_bwd(const float* g_out, const StegoMap* d_map, const int64_t* index, int32_t N, int32_t C, int32_t H, int32_t W,
                     const float* coords, int32_t n_coords, int32_t S, stego_stream_t stream)
{
    const int rc = check_sample(d_map, N, C, H, W, coords, n_coords, S, g_out);
    if (rc != STEGO_OK) return rc;
    if (N == 0) return STEGO_OK;
    const GatherParams p = make_params(d_map, index, N, C, H, W, coords, n_coords, S, const_cast<float*>(g_out), const_cast<float*>(d_map->data));
    const dim3 grid((unsigned)(((long long)N * p.P + 3) / 4)), block(256);
    hipLaunchKernelGGL(sample_scatter_kernel, grid, block, 0, static_cast<hipStream_t>(stream), p);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? STEGO_OK : STEGO_ERR_HIP + (int)e;
}

int stego_sample_bwd_rows(const float* g_rows, const float* rows_n, const float* inv, const StegoMap* d_map, const int64_t* index, int32_t N,
                          int32_t C, int32_t H, int32_t W, const float* coords, int32_t n_coords, int32_t S, stego_stream_t stream)
{
    const int rc = check_sample(d_map, N, C, H, W, coords, n_coords, S, g_rows);
    if (rc != STEGO_OK) return rc;
    if ((rows_n == nullptr) != (inv == nullptr)) return STEGO_ERR_NULL;
    if (N == 0) return STEGO_OK;
    GatherParams p = make_params(d_map, index, N, C, H, W, coords, n_coords, S, const_cast<float*>(g_rows), const_cast<float*>(d_map->data));
    p.rows_n = rows_n;
    p.inv = inv;
    const dim3 grid((unsigned)(((long long)N * p.P + 3) / 4)), block(256);
    hipLaunchKernelGGL(sample_scatter_kernel, grid, block, 0, static_cast<hipStream_t>(stream), p);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? STEGO_OK : STEGO_ERR_HIP + (int)e;
}

}
