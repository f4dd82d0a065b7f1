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
    int N, C, H, W, S, P, n_coords;
};

__device__ __forceinline__ void point_of(const GatherParams& p, int gw, int& n, int& q, long long& img, int4& yx, float4& w)
{
    n = gw / p.P;
    q = gw - n * p.P;
    img = p.index ? p.index[n] : (long long)n;
    const int hh = q / p.S, ww = q - hh * p.S;          // point (h, w) reads coords[w][h] (the permute of modules.py:288)
    const float* c = p.coords + ((size_t)(n % p.n_coords) * p.P + (size_t)ww * p.S + hh) * 2;
    make_taps(c[0], c[1], p.H, p.W, yx, w);
}

template <bool VEC>
__global__ void __launch_bounds__(256) sample_gather_kernel(const GatherParams p)
{
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
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
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
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
    for (int c = lane; c < p.C; c += 64) {
        const long long oc = (long long)c * p.map.sc;
        const float v = g[c];
        unsafeAtomicAdd(base + o0 + oc, w.x * v);
        unsafeAtomicAdd(base + o1 + oc, w.y * v);
        unsafeAtomicAdd(base + o2 + oc, w.z * v);
        unsafeAtomicAdd(base + o3 + oc, w.w * v);
    }
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
    p.N = N; p.C = C; p.H = H; p.W = W; p.S = S; p.P = S * S; p.n_coords = n_coords;
    return p;
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

}
