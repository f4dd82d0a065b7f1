// _stego_torchglue: what ties the C ABI (include/stego_corr.h) into a torch program without Python in the per-step path.
// Host code only - no kernels here; everything that computes is behind the C ABI in libstego_corr.so, which this module dlopen()s.
//
// (1) philox_state / ref_draws: the device generator's Philox state in its graph-safe form.  ContrastiveCorrelationLoss.forward draws
//     torch.rand x 2 and torch.randperm x neg_samples from the device's default generator (reference src/modules.py:366-367, :291-295,
//     :383); stego_ref_draws makes the same numbers in one launch from the generator's (seed, offset).  While a stream is being captured
//     that state lives in device memory that CUDAGraph::replay refreshes before every replay; a kernel has to read it there (seed
//     pointer, extragraph-offset pointer, offset inside the graph) - what ATen's own distribution kernels receive from
//     CUDAGeneratorImpl::philox_cuda_state(increment).  Python cannot reach it; this can.  The increment is registered with the
//     generator in the same call (eagerly: advances the offset; capturing: grows the graph's whole-graph increment), so the generator
//     moves exactly as the seven torch calls would move it.
// (3) head: the segmentation head's autograd function (stego_head_fwd / stego_head_bwd), same reason as (2).
// (2) corr_loss: ContrastiveCorrelationLoss.forward (src/modules.py:349-398, given the draws) as a torch::autograd::Function over
//     stego_corr_fwd_prepared / stego_corr_bwd.  The reference's training loop is eager (Lightning, train_segmentation.py:154-181): what
//     a user of the drop-in pays per step is the host side of this call, and the Python autograd.Function it replaces
//     (stego_amd/modules.py::_CorrLossFunction, kept for hosts without this module) cost more host time than both kernels take.
#include <torch/extension.h>
#include <ATen/hip/HIPGeneratorImpl.h>
#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime_api.h>
#include <dlfcn.h>
#include <cstring>
#include <map>
#include <set>
#include <mutex>
#include <string>
#include <tuple>
#include "../../include/stego_corr.h"
#include "../../include/stego_head.h"

namespace {

struct Lib {
    void* handle = nullptr;
    decltype(&stego_abi_version) abi_version;
    decltype(&stego_error_string) error_string;
    decltype(&stego_corr_workspace_bytes) workspace_bytes;
    decltype(&stego_corr_saved_ctx_bytes) saved_ctx_bytes;
    decltype(&stego_corr_bwd_workspace_bytes) bwd_workspace_bytes;
    decltype(&stego_corr_workspace_prepare) workspace_prepare;
    decltype(&stego_corr_workspace_prepare_now) workspace_prepare_now;
    decltype(&stego_corr_fwd_prepared) fwd_prepared;
    decltype(&stego_corr_bwd) bwd;
    decltype(&stego_ref_draws) ref_draws;
    decltype(&stego_ref_draws_indirect) ref_draws_indirect;
    decltype(&stego_ref_draws_advance) ref_draws_advance;
    decltype(&stego_head_fwd_workspace_bytes) head_fwd_ws;
    decltype(&stego_head_bwd_workspace_bytes) head_bwd_ws;
    decltype(&stego_head_fwd) head_fwd;
    decltype(&stego_head_bwd) head_bwd;
} L;

template <class F> void sym(F& f, const char* name)
{
    f = reinterpret_cast<F>(dlsym(L.handle, name));
    TORCH_CHECK(f != nullptr, "libstego_corr.so does not export ", name);
}

void bind(const std::string& path)
{
    L.handle = dlopen(path.c_str(), RTLD_NOW | RTLD_GLOBAL);
    TORCH_CHECK(L.handle != nullptr, "cannot load ", path, ": ", dlerror());
    sym(L.abi_version, "stego_abi_version");
    sym(L.error_string, "stego_error_string");
    sym(L.workspace_bytes, "stego_corr_workspace_bytes");
    sym(L.saved_ctx_bytes, "stego_corr_saved_ctx_bytes");
    sym(L.bwd_workspace_bytes, "stego_corr_bwd_workspace_bytes");
    sym(L.workspace_prepare, "stego_corr_workspace_prepare");
    sym(L.workspace_prepare_now, "stego_corr_workspace_prepare_now");
    sym(L.fwd_prepared, "stego_corr_fwd_prepared");
    sym(L.bwd, "stego_corr_bwd");
    sym(L.ref_draws, "stego_ref_draws");
    sym(L.ref_draws_indirect, "stego_ref_draws_indirect");
    sym(L.ref_draws_advance, "stego_ref_draws_advance");
    sym(L.head_fwd_ws, "stego_head_fwd_workspace_bytes");
    sym(L.head_bwd_ws, "stego_head_bwd_workspace_bytes");
    sym(L.head_fwd, "stego_head_fwd");
    sym(L.head_bwd, "stego_head_bwd");
    TORCH_CHECK(L.abi_version() == STEGO_ABI_VERSION, "libstego_corr.so has ABI ", L.abi_version(), ", this module was built for ",
                STEGO_ABI_VERSION);
}

void reset_workspaces();

// A failed or aborted launch can leave the hand-off words of a prepared workspace dirty: nothing cached survives an error.
void check(int rc, const char* what)
{
    if (rc != STEGO_OK) reset_workspaces();
    TORCH_CHECK(rc == STEGO_OK, what, ": ", L.error_string(rc), " (", rc, ")");
}

void require_device(const at::Tensor& t, const char* name)
{
    TORCH_CHECK(t.is_cuda(), "stego_amd runs on MI355X only: ", name, " is a ", t.device(), " tensor (no CPU fallback exists)");
}

hipStream_t current_stream(const at::Tensor& t)
{
    return c10::hip::getCurrentHIPStream(t.device().index()).stream();
}

bool capturing(hipStream_t s)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(s, &st) == hipSuccess && st == hipStreamCaptureStatusActive;
}

at::Tensor dense_f32(const at::Tensor& t)
{
    return (t.scalar_type() == at::kFloat ? t : t.to(at::kFloat)).contiguous();
}

// ------------------------------------------------------------------------------------------------ (1) the generator
// (captured, seed value or device pointer to it, offset value or device pointer to the extragraph offset, offset inside the graph)
std::tuple<bool, uint64_t, uint64_t, uint64_t> philox_state(const at::Generator& gen, int64_t increment)
{
    TORCH_CHECK(increment >= 0 && increment % 4 == 0, "increment must be a non-negative multiple of 4 (one Philox block)");
    auto* impl = at::check_generator<at::CUDAGeneratorImpl>(gen);
    at::PhiloxCudaState st;
    {
        std::lock_guard<std::mutex> lock(impl->mutex_);
        st = impl->philox_cuda_state(static_cast<uint64_t>(increment));
    }
    if (st.captured_)
        return {true, reinterpret_cast<uint64_t>(st.seed_.ptr), reinterpret_cast<uint64_t>(st.offset_.ptr), st.offset_intragraph_};
    return {false, st.seed_.val, st.offset_.val, 0};
}

// (coords1, coords2, perms) of a forward: torch.rand(B, S, S, 2) * 2 - 1 twice, super_perm(B) x n_neg - the generator's numbers
std::tuple<at::Tensor, at::Tensor, at::Tensor> ref_draws(const at::Generator& gen, int64_t B, int64_t S, int64_t n_neg, int64_t variant,
                                                         const at::Device& dev)
{
    TORCH_CHECK(L.handle, "bind() first");
    c10::DeviceGuard guard(dev);
    const auto f32 = at::TensorOptions().dtype(at::kFloat).device(dev);
    at::Tensor c1 = at::empty({B, S, S, 2}, f32), c2 = at::empty({B, S, S, 2}, f32);
    at::Tensor perms = at::empty({n_neg, B}, f32.dtype(at::kLong));
    const int64_t n_coord = c1.numel();
    const uint64_t adv = L.ref_draws_advance(n_coord, (int32_t)n_neg, (int32_t)B, (int32_t)variant);
    const auto st = philox_state(gen, (int64_t)adv);
    hipStream_t stream = c10::hip::getCurrentHIPStream(dev.index()).stream();
    if (std::get<0>(st))
        check(L.ref_draws_indirect(reinterpret_cast<const int64_t*>(std::get<1>(st)), reinterpret_cast<const int64_t*>(std::get<2>(st)),
                                   std::get<3>(st), (int32_t)variant, n_coord, (int32_t)n_neg, (int32_t)B, c1.data_ptr<float>(),
                                   c2.data_ptr<float>(), perms.data_ptr<int64_t>(), stream), "stego_ref_draws_indirect");
    else
        check(L.ref_draws(std::get<1>(st), std::get<2>(st), (int32_t)variant, n_coord, (int32_t)n_neg, (int32_t)B, c1.data_ptr<float>(),
                          c2.data_ptr<float>(), perms.data_ptr<int64_t>(), stream), "stego_ref_draws");
    return {c1, c2, perms};
}

// ------------------------------------------------------------------------------------------------ (2) the loss
// Forward workspaces are kept per (device, stream, descriptor): the fused forward hands data between workgroups through a few counters
// that must be zero when a launch starts and that every launch leaves zero again (stego_corr_workspace_prepare once, then one launch
// per call).  A workspace first met during a capture is prepared on the library's side stream (stego_corr_workspace_prepare_now)
// instead of as a memset node in every replay.
std::recursive_mutex ws_mutex;
std::map<std::tuple<int, uintptr_t, std::string>, at::Tensor> ws_cache;
std::set<std::tuple<int, uintptr_t, std::string>> ws_pinned;       // first used during a stream capture: never evicted

at::Tensor prepared_workspace(const StegoCorrDesc& d, const std::string& desc_bytes, const at::Device& dev, hipStream_t stream)
{
    std::lock_guard<std::recursive_mutex> lock(ws_mutex);
    const auto key = std::make_tuple((int)dev.index(), reinterpret_cast<uintptr_t>(stream), desc_bytes);
    auto it = ws_cache.find(key);
    if (it != ws_cache.end()) return it->second;
    if (ws_cache.size() >= 16) {
        // forget one workspace that no captured graph holds (a graph replays with the pointer it captured)
        for (auto e = ws_cache.begin(); e != ws_cache.end(); ++e)
            if (!ws_pinned.count(e->first)) { ws_cache.erase(e); break; }
    }
    const size_t n = L.workspace_bytes(&d);
    TORCH_CHECK(n > 0, "stego_corr_workspace_bytes: ", L.error_string(STEGO_ERR_UNSUPPORTED));
    at::Tensor ws = at::empty({(int64_t)n}, at::TensorOptions().dtype(at::kByte).device(dev));
    if (capturing(stream)) {
        ws_pinned.insert(key);
        check(L.workspace_prepare_now(&d, ws.data_ptr(), n), "stego_corr_workspace_prepare_now");
    } else {
        check(L.workspace_prepare(&d, ws.data_ptr(), n, stream), "stego_corr_workspace_prepare");
        check(L.workspace_prepare_now(nullptr, nullptr, 0), "stego_corr_workspace_prepare_now");      // (its side stream: made outside a capture)
    }
    ws_cache.emplace(key, ws);
    return ws;
}

void reset_workspaces()
{
    std::lock_guard<std::recursive_mutex> lock(ws_mutex);
    ws_cache.clear();
    ws_pinned.clear();
}

// (descriptor bytes, workspace) of every kept forward workspace: capi.event_counters_total() reads their two event words
std::vector<std::pair<py::bytes, at::Tensor>> workspaces()
{
    std::lock_guard<std::recursive_mutex> lock(ws_mutex);
    std::vector<std::pair<py::bytes, at::Tensor>> out;
    for (auto& e : ws_cache) out.emplace_back(py::bytes(std::get<2>(e.first)), e.second);
    return out;
}

StegoMap as_map(const at::Tensor& t, const char* name)
{
    TORCH_CHECK(t.scalar_type() == at::kFloat && t.dim() == 4, "expected a float32 [N,C,H,W] tensor for ", name);
    return StegoMap{t.data_ptr<float>(), t.stride(0), t.stride(1), t.stride(2), t.stride(3)};
}

const float* fptr(const at::Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }

using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

// mode 0: (pos_intra.mean(), pos_intra_cd, pos_inter.mean(), pos_inter_cd, neg_inter_loss, neg_inter_cd, neg_inter_loss.mean(),
//          means [3] = the three scalars as ONE autograd output: a weighted sum of them is differentiated from one root with one
//          coefficient vector as its upstream, modules.py _LazyLoss.backward)
// mode 1: (means [3], pos_intra_cd, pos_inter_cd, neg_inter_cd)
struct CorrLoss : public torch::autograd::Function<CorrLoss> {
    static variable_list forward(AutogradContext* ctx, const at::Tensor& feats, const at::Tensor& feats_pos, const at::Tensor& code,
                                 const at::Tensor& code_pos, const at::Tensor& coords1_in, const at::Tensor& coords2_in,
                                 const at::Tensor& perms_in, const std::string& desc_bytes, int64_t mode)
    {
        TORCH_CHECK(L.handle, "bind() first");
        TORCH_CHECK(desc_bytes.size() == sizeof(StegoCorrDesc), "descriptor size");
        StegoCorrDesc d;
        std::memcpy(&d, desc_bytes.data(), sizeof d);
        require_device(feats, "feats"); require_device(feats_pos, "feats_pos"); require_device(code, "code");
        require_device(code_pos, "code_pos"); require_device(coords1_in, "coords1"); require_device(coords2_in, "coords2");
        const bool need_grad = code.requires_grad() || code_pos.requires_grad();
        ctx->set_materialize_grads(false);       // an output nobody differentiated costs no zero-fill and no loads in the backward
        const at::Device dev = feats.device();
        c10::DeviceGuard guard(dev);
        hipStream_t stream = current_stream(feats);
        at::Tensor coords1 = dense_f32(coords1_in), coords2 = dense_f32(coords2_in), perms;
        if (d.n_neg > 0) {
            require_device(perms_in, "perms");
            perms = (perms_in.scalar_type() == at::kLong ? perms_in : perms_in.to(at::kLong)).contiguous();
        }
        const int64_t B = d.B, S = d.S, n_neg = d.n_neg;
        const auto f32 = at::TensorOptions().dtype(at::kFloat).device(dev);
        at::Tensor means = at::empty({3}, f32);
        at::Tensor intra_cd = at::empty({B, S, S, S, S}, f32), inter_cd = at::empty({B, S, S, S, S}, f32);
        at::Tensor neg_loss = at::empty({n_neg * B, S, S, S, S}, f32), neg_cd = at::empty({n_neg * B, S, S, S, S}, f32);
        at::Tensor saved_w, saved_mean, saved_ctx;
        if (need_grad) {
            saved_w = at::empty({(2 + n_neg) * B, S * S * S * S}, f32);
            saved_mean = at::empty({2 + n_neg}, f32);
            saved_ctx = at::empty({(int64_t)std::max<size_t>(L.saved_ctx_bytes(&d), 16)}, f32.dtype(at::kByte));
        }
        at::Tensor ws = prepared_workspace(d, desc_bytes, dev, stream);
        const StegoMap mf = as_map(feats, "feats"), mfp = as_map(feats_pos, "feats_pos"), mc = as_map(code, "code"),
                       mcp = as_map(code_pos, "code_pos");
        check(L.fwd_prepared(&d, &mf, &mfp, &mc, &mcp, coords1.data_ptr<float>(), coords2.data_ptr<float>(),
                             perms.defined() ? perms.data_ptr<int64_t>() : nullptr, means.data_ptr<float>(), intra_cd.data_ptr<float>(),
                             inter_cd.data_ptr<float>(), neg_loss.data_ptr<float>(), neg_cd.data_ptr<float>(),
                             need_grad ? saved_w.data_ptr<float>() : nullptr, need_grad ? saved_mean.data_ptr<float>() : nullptr,
                             need_grad ? saved_ctx.data_ptr() : nullptr, ws.data_ptr(), (size_t)ws.numel(), stream),
              "stego_corr_fwd_prepared");
        ctx->saved_data["desc"] = desc_bytes;
        ctx->saved_data["mode"] = mode;
        if (need_grad)
            ctx->save_for_backward({perms.defined() ? perms : at::Tensor(), intra_cd, inter_cd, neg_cd, saved_w, saved_mean, saved_ctx});
        if (mode == 1) return {means, intra_cd, inter_cd, neg_cd};
        return {means.select(0, 0), intra_cd, means.select(0, 1), inter_cd, neg_loss, neg_cd, means.select(0, 2), means};
    }

    static variable_list backward(AutogradContext* ctx, variable_list g)
    {
        for (const auto& t : g)
            TORCH_CHECK(!(t.defined() && t.requires_grad() && at::GradMode::is_enabled()),
                        "the correspondence loss is differentiable once (no double backward)");
        const std::string desc_bytes = ctx->saved_data["desc"].toStringRef();
        const int64_t mode = ctx->saved_data["mode"].toInt();
        StegoCorrDesc d;
        std::memcpy(&d, desc_bytes.data(), sizeof d);
        const auto saved = ctx->get_saved_variables();
        const at::Tensor &perms = saved[0], &intra_cd = saved[1], &inter_cd = saved[2], &neg_cd = saved[3], &saved_w = saved[4],
                         &saved_mean = saved[5], &saved_ctx = saved[6];
        at::Tensor g_intra, g_inter, g_neg, g_intra_cd, g_inter_cd, g_neg_cd;
        int32_t stride = 1;
        if (mode == 1) {
            if (g[0].defined()) {
                const at::Tensor gm = dense_f32(g[0]);
                g_intra = gm.narrow(0, 0, 1); g_inter = gm.narrow(0, 1, 1);
                if (d.n_neg > 0) g_neg = gm.narrow(0, 2, 1);
            }
            stride = -1;                                   // one device scalar: the upstream of the mean over the negative loss tensor
            g_intra_cd = g[1]; g_inter_cd = g[2]; g_neg_cd = g[3];
        } else {
            g_intra = g[0]; g_intra_cd = g[1]; g_inter = g[2]; g_inter_cd = g[3]; g_neg = g[4]; g_neg_cd = g[5];
            at::Tensor g_neg_mean = g[6];
            if (g.size() > 7 && g[7].defined()) {          // the three scalars differentiated through the vector output (added to the
                const at::Tensor gm = dense_f32(g[7]);     // scalar outputs' own upstreams, should a caller have used both)
                const at::Tensor m0 = gm.narrow(0, 0, 1), m1 = gm.narrow(0, 1, 1), m2 = gm.narrow(0, 2, 1);
                g_intra = g_intra.defined() ? g_intra.reshape({1}) + m0 : m0;
                g_inter = g_inter.defined() ? g_inter.reshape({1}) + m1 : m1;
                if (d.n_neg > 0) g_neg_mean = g_neg_mean.defined() ? g_neg_mean.reshape({1}) + m2 : m2;
            }
            bool neg_is_mean = false;
            if (g_neg_mean.defined() && d.n_neg > 0) {
                if (!g_neg.defined()) { g_neg = g_neg_mean.reshape({1}); neg_is_mean = true; }       // the training case
                else g_neg = g_neg + g_neg_mean / (double)g_neg.numel();                          // both the map and its mean
            }
            if (neg_is_mean) { stride = -1; g_neg = dense_f32(g_neg); }
            else if (g_neg.defined() && g_neg.numel() > 0) {
                bool expanded = true;
                for (int64_t s : g_neg.strides()) expanded = expanded && s == 0;
                if (expanded) stride = 0;                                                            // expanded scalar: the one element
                else g_neg = dense_f32(g_neg);
            } else g_neg = at::Tensor();
            if (g_intra.defined()) g_intra = dense_f32(g_intra);
            if (g_inter.defined()) g_inter = dense_f32(g_inter);
        }
        if (g_intra_cd.defined()) g_intra_cd = dense_f32(g_intra_cd);
        if (g_inter_cd.defined()) g_inter_cd = dense_f32(g_inter_cd);
        if (g_neg_cd.defined() && g_neg_cd.numel() > 0) g_neg_cd = dense_f32(g_neg_cd); else g_neg_cd = at::Tensor();
        const at::Device dev = saved_w.device();
        c10::DeviceGuard guard(dev);
        hipStream_t stream = current_stream(saved_w);
        const auto f32 = at::TensorOptions().dtype(at::kFloat).device(dev);
        at::Tensor d_code = at::empty({d.B, d.H, d.W, d.K}, f32), d_code_pos = at::empty({d.B, d.H, d.W, d.K}, f32);
        const size_t wsn = std::max<size_t>(L.bwd_workspace_bytes(&d), 16);
        at::Tensor ws = at::empty({(int64_t)wsn}, f32.dtype(at::kByte));
        check(L.bwd(&d, perms.defined() ? perms.data_ptr<int64_t>() : nullptr, saved_w.data_ptr<float>(), saved_mean.data_ptr<float>(),
                    saved_ctx.data_ptr(), intra_cd.data_ptr<float>(), inter_cd.data_ptr<float>(), neg_cd.data_ptr<float>(), fptr(g_intra),
                    fptr(g_inter), fptr(g_neg), stride, fptr(g_intra_cd), fptr(g_inter_cd), fptr(g_neg_cd), d_code.data_ptr<float>(),
                    d_code_pos.data_ptr<float>(), ws.data_ptr(), wsn, stream), "stego_corr_bwd");
        at::Tensor none;
        return {none, none, ctx->needs_input_grad(2) ? d_code.permute({0, 3, 1, 2}) : none,
                ctx->needs_input_grad(3) ? d_code_pos.permute({0, 3, 1, 2}) : none, none, none, none, none, none};
    }
};

std::vector<at::Tensor> corr_loss(const at::Tensor& feats, const at::Tensor& feats_pos, const at::Tensor& code, const at::Tensor& code_pos,
                                  const at::Tensor& coords1, const at::Tensor& coords2, const at::Tensor& perms, const std::string& desc,
                                  int64_t mode)
{
    return CorrLoss::apply(feats, feats_pos, code, code_pos, coords1, coords2, perms, desc, mode);
}

// ------------------------------------------------------------------------------------------------ (3) the segmentation head
// DinoFeaturizer.forward's tail (src/modules.py:108-116) as a torch::autograd::Function over stego_head_fwd / stego_head_bwd - the C++
// twin of stego_amd/featurizers.py::_NativeHeadFunction (kept for hosts without this module): two such calls per training step.
using OptT = c10::optional<at::Tensor>;
const float* optp(const OptT& t) { return (t.has_value() && t->defined()) ? t->data_ptr<float>() : nullptr; }

StegoHeadDesc head_desc(const at::Tensor& tokens, int64_t K, bool nonlinear)
{
    TORCH_CHECK(tokens.dim() == 3 && tokens.scalar_type() == at::kFloat && tokens.stride(2) == 1,
                "head: expected a float32 [B, HW, C] token tensor with contiguous channels");
    StegoHeadDesc d{};
    d.B = (int32_t)tokens.size(0); d.HW = (int32_t)tokens.size(1); d.C = (int32_t)tokens.size(2); d.K = (int32_t)K;
    d.nonlinear = nonlinear ? 1 : 0;
    d.tok_stride = tokens.stride(1);
    d.img_stride = d.B > 1 ? tokens.stride(0) : (int64_t)d.HW * tokens.stride(1);
    d.tokens_amax = nullptr;
    return d;
}

struct HeadFn : public torch::autograd::Function<HeadFn> {
    static variable_list forward(AutogradContext* ctx, const at::Tensor& tokens, const OptT& m1, const OptT& m2, const OptT& m3,
                                 const at::Tensor& w1, const at::Tensor& b1, const OptT& w21, const OptT& b21, const OptT& w22,
                                 const OptT& b22, bool want_feats, const OptT& tokens_amax)
    {
        TORCH_CHECK(L.handle, "bind() first");
        require_device(tokens, "tokens"); require_device(w1, "cluster1.weight");
        const bool nonlinear = w21.has_value() && w21->defined();
        bool need_grad = w1.requires_grad() || b1.requires_grad();
        for (const OptT* t : {&w21, &b21, &w22, &b22}) need_grad = need_grad || (t->has_value() && (*t)->defined() && (*t)->requires_grad());
        StegoHeadDesc d = head_desc(tokens, w1.size(0), nonlinear);
        if (tokens_amax.has_value() && tokens_amax->defined()) d.tokens_amax = reinterpret_cast<const uint32_t*>(tokens_amax->data_ptr());
        const at::Device dev = tokens.device();
        c10::DeviceGuard guard(dev);
        hipStream_t stream = current_stream(tokens);
        const auto f32 = at::TensorOptions().dtype(at::kFloat).device(dev);
        const int64_t B = d.B, HW = d.HW, C = d.C, K = d.K;
        at::Tensor code = at::empty({B, HW, K}, f32), feats, saved_h;
        if (want_feats) feats = at::empty({B, HW, C}, f32);
        if (nonlinear && need_grad) saved_h = at::empty({B * HW * C + 8}, f32);
        size_t nws = L.head_fwd_ws(&d);
        TORCH_CHECK(nws > 0, "stego_head_fwd: unsupported shape (C a multiple of 32, K <= 128, 16-byte token rows)");
        if (nonlinear && saved_h.defined()) nws -= ((size_t)B * HW * C * 4 + 255) / 256 * 256;      // (H lives in saved_h)
        at::Tensor ws = at::empty({(int64_t)nws}, f32.dtype(at::kByte));
        auto wp = [](const at::Tensor& t) { return t.data_ptr<float>(); };
        TORCH_CHECK(w1.is_contiguous() && b1.is_contiguous(), "head: contiguous parameters expected");
        check(L.head_fwd(&d, tokens.data_ptr<float>(), optp(m1), optp(m2), optp(m3), wp(w1), wp(b1), optp(w21), optp(b21), optp(w22),
                         optp(b22), code.data_ptr<float>(), feats.defined() ? feats.data_ptr<float>() : nullptr,
                         saved_h.defined() ? saved_h.data_ptr<float>() : nullptr, ws.data_ptr(), nws, stream), "stego_head_fwd");
        ctx->set_materialize_grads(false);
        ctx->saved_data["K"] = K;
        ctx->saved_data["nonlinear"] = nonlinear;
        if (need_grad) {
            auto ot = [](const OptT& t) { return (t.has_value() && t->defined()) ? *t : at::Tensor(); };
            ctx->save_for_backward({tokens, ot(m1), ot(m2), saved_h, ot(w22)});
        }
        if (!feats.defined()) return {code};             // (an undefined tensor cannot be an output: eval mode returns the code alone)
        ctx->mark_non_differentiable({feats});
        return {code, feats};
    }

    static variable_list backward(AutogradContext* ctx, variable_list g)
    {
        at::Tensor none;
        variable_list out(12, none);
        if (!g[0].defined()) return out;
        TORCH_CHECK(!(g[0].requires_grad() && at::GradMode::is_enabled()), "the head kernels are differentiable once (no double backward)");
        const auto saved = ctx->get_saved_variables();
        const at::Tensor &tokens = saved[0], &m1 = saved[1], &m2 = saved[2], &saved_h = saved[3], &w22 = saved[4];
        const bool nonlinear = ctx->saved_data["nonlinear"].toBool();
        const int64_t K = ctx->saved_data["K"].toInt();
        StegoHeadDesc d = head_desc(tokens, K, nonlinear);
        const at::Device dev = tokens.device();
        c10::DeviceGuard guard(dev);
        hipStream_t stream = current_stream(tokens);
        const auto f32 = at::TensorOptions().dtype(at::kFloat).device(dev);
        const int64_t C = d.C;
        at::Tensor dw1 = at::empty({K, C}, f32), db1 = at::empty({K}, f32), dw21, db21, dw22, db22;
        if (nonlinear) { dw21 = at::empty({C, C}, f32); db21 = at::empty({C}, f32); dw22 = at::empty({K, C}, f32); db22 = at::empty({K}, f32); }
        const at::Tensor d_code = dense_f32(g[0]);
        const size_t nws = L.head_bwd_ws(&d);
        at::Tensor ws = at::empty({(int64_t)nws}, f32.dtype(at::kByte));
        check(L.head_bwd(&d, tokens.data_ptr<float>(), fptr(m1), fptr(m2), nonlinear ? saved_h.data_ptr<float>() : nullptr, fptr(w22),
                         d_code.data_ptr<float>(), dw1.data_ptr<float>(), db1.data_ptr<float>(), nonlinear ? dw21.data_ptr<float>() : nullptr,
                         nonlinear ? db21.data_ptr<float>() : nullptr, nonlinear ? dw22.data_ptr<float>() : nullptr,
                         nonlinear ? db22.data_ptr<float>() : nullptr, ws.data_ptr(), nws, stream), "stego_head_bwd");
        out[4] = dw1; out[5] = db1; out[6] = dw21; out[7] = db21; out[8] = dw22; out[9] = db22;
        return out;
    }
};

std::vector<at::Tensor> head(const at::Tensor& tokens, const OptT& m1, const OptT& m2, const OptT& m3, const at::Tensor& w1,
                             const at::Tensor& b1, const OptT& w21, const OptT& b21, const OptT& w22, const OptT& b22, bool want_feats,
                             const OptT& tokens_amax)
{
    return HeadFn::apply(tokens, m1, m2, m3, w1, b1, w21, b21, w22, b22, want_feats, tokens_amax);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.def("bind", &bind, "dlopen libstego_corr.so and resolve the entry points this module calls");
    m.def("philox_state", &philox_state, "graph-safe Philox state of a device generator, advanced by `increment`");
    m.def("ref_draws", &ref_draws, "(coords1, coords2, perms): the generator's torch.rand x 2 / torch.randperm x n_neg, one launch");
    m.def("corr_loss", [](const at::Tensor& a, const at::Tensor& b, const at::Tensor& c, const at::Tensor& d, const at::Tensor& e,
                          const at::Tensor& f, const at::Tensor& g, const py::bytes& desc, int64_t mode) {
        return corr_loss(a, b, c, d, e, f, g, std::string(desc), mode);
    }, "ContrastiveCorrelationLoss.forward given the draws (autograd through stego_corr_bwd)");
    m.def("reset_workspaces", &reset_workspaces);
    m.def("workspaces", &workspaces, "[(descriptor bytes, workspace tensor)] of the kept forward workspaces");
    m.def("head", &head, "DinoFeaturizer's head (dropout masks given) -> [code [B, HW, K]] or [code, feats [B, HW, C]] (want_feats)");
}
