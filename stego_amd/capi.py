"""ctypes binding of libstego_corr.so (include/stego_corr.h) for torch tensors.

This is the ONLY compute backend of stego_amd: there is no PyTorch/CPU fallback.  If the
library is missing or the tensors are not on a HIP device the calls raise.
PyTorch is used for device memory (caching allocator), streams and autograd plumbing only.
"""
import ctypes
import os
from ctypes import POINTER, Structure, byref, c_float, c_int32, c_int64, c_size_t, c_void_p

import torch

from . import _build

ABI_VERSION = 7
FLAG_SHARED_DEVICE = 1          # StegoCorrDesc.flags
PREC_F32 = 0
PREC_F16X3 = 1
PREC_BF16X3 = PREC_F16X3        # old name of the split mode (it used bf16 halves until round 1, third design)


class StegoMap(Structure):
    _fields_ = [("data", c_void_p), ("stride_n", c_int64), ("stride_c", c_int64),
                ("stride_h", c_int64), ("stride_w", c_int64)]


class StegoCorrDesc(Structure):
    _fields_ = [("B", c_int32), ("C", c_int32), ("K", c_int32), ("H", c_int32), ("W", c_int32),
                ("S", c_int32), ("n_neg", c_int32), ("pointwise", c_int32), ("zero_clamp", c_int32),
                ("stabalize", c_int32), ("pos_intra_shift", c_float), ("pos_inter_shift", c_float),
                ("neg_inter_shift", c_float), ("precision", c_int32), ("flags", c_int32)]


# name -> (restype, argtypes); every symbol include/stego_corr.h declares
_P = c_void_p
class StegoVitDesc(Structure):
    """include/stego_vit.h"""
    _fields_ = [(n, c_int32) for n in ("B", "H", "W", "patch", "D", "depth", "heads", "hidden", "precision")]


VIT_F16, VIT_F16X3 = 0, 1


class StegoHeadDesc(Structure):
    """include/stego_head.h"""
    _fields_ = [("B", c_int32), ("HW", c_int32), ("C", c_int32), ("K", c_int32), ("nonlinear", c_int32),
                ("tok_stride", c_int64), ("img_stride", c_int64), ("tokens_amax", c_void_p)]


_H = POINTER(StegoHeadDesc)
_D = POINTER(StegoCorrDesc)
_M = POINTER(StegoMap)
_V = POINTER(StegoVitDesc)
SIGNATURES = {
    "stego_vit_param_count": (c_int32, [_V]),
    "stego_vit_weights_bytes": (c_size_t, [_V]),
    "stego_vit_workspace_bytes": (c_size_t, [_V]),
    "stego_vit_pack_weights": (c_int32, [_V, POINTER(ctypes.c_void_p), c_int32, _P, c_size_t, _P]),
    "stego_vit_forward": (c_int32, [_V, _P, _P, _P, _P, c_size_t, _P]),
    "stego_head_fwd_workspace_bytes": (c_size_t, [_H]),
    "stego_head_bwd_workspace_bytes": (c_size_t, [_H]),
    "stego_head_fwd": (c_int32, [_H] + [_P] * 10 + [_P] * 3 + [_P, c_size_t, _P]),
    "stego_head_bwd": (c_int32, [_H] + [_P] * 6 + [_P] * 6 + [_P, c_size_t, _P]),
    "stego_abi_version": (c_int32, []),
    "stego_debug_set": (c_int32, [c_int32, c_int32]),
    "stego_debug_occupy": (c_int32, [c_int32, c_int32, c_int32, _P]),
    "stego_error_string": (ctypes.c_char_p, [c_int32]),
    "stego_corr_workspace_bytes": (c_size_t, [_D]),
    "stego_corr_saved_ctx_bytes": (c_size_t, [_D]),
    "stego_corr_helper_workspace_bytes": (c_size_t, [_D]),
    "stego_corr_helper_saved_ctx_bytes": (c_size_t, [_D]),
    "stego_dense_corr_workspace_bytes": (c_size_t, [c_int32] * 6),
    "stego_dense_corr": (c_int32, [_M, _M] + [c_int32] * 7 + [_P, _P, c_size_t, _P]),
    "stego_sample": (c_int32, [_M, _P, c_int32, c_int32, c_int32, c_int32, _P, c_int32, c_int32, _P, _P]),
    "stego_sample_bwd": (c_int32, [_P, _M, _P, c_int32, c_int32, c_int32, c_int32, _P, c_int32, c_int32, _P]),
    "stego_sample_bwd_rows": (c_int32, [_P, _P, _P, _M, _P, c_int32, c_int32, c_int32, c_int32, _P, c_int32, c_int32, _P]),
    "stego_panel_image_bytes": (c_size_t, [c_int32, c_int32]),
    "stego_sample_panels": (c_int32, [_M, _P, c_int32, c_int32, c_int32, c_int32, _P, c_int32, c_int32, c_int32, _P, _P, _P, _P, _P]),
    "stego_dense_corr_panels": (c_int32, [_P, _P, c_int32, _P, _P, c_int32, c_int32, c_int32, c_int32, _P, _P, _P]),
    "stego_rowsum": (c_int32, [_P, ctypes.c_int64, c_int32, _P, _P]),
    "stego_loss_pointwise_fwd": (c_int32, [_P] * 4 + [c_int32] * 3 + [POINTER(c_float), c_float, c_float, c_int32, _P, _P, _P]),
    "stego_loss_pointwise_bwd": (c_int32, [_P] * 4 + [c_int32] * 3 + [POINTER(c_float), c_float, c_float, c_int32, _P, _P, _P, _P, _P]),
    "stego_knn_workspace_bytes": (c_size_t, [ctypes.c_int64, c_int32, c_int32, ctypes.c_int64]),
    "stego_knn_topk": (c_int32, [_P, ctypes.c_int64, c_int32, ctypes.c_int64, c_int32, c_int32, ctypes.c_int64, ctypes.c_int64,
                                 _P, _P, _P, c_size_t, _P]),
    "stego_corr_fwd": (c_int32, [_D] + [_M] * 4 + [_P] * 3 + [_P] * 8 + [_P, c_size_t, _P]),
    "stego_corr_fwd_prepared": (c_int32, [_D] + [_M] * 4 + [_P] * 3 + [_P] * 8 + [_P, c_size_t, _P]),
    "stego_corr_workspace_prepare": (c_int32, [_D, _P, c_size_t, _P]),
    "stego_corr_workspace_prepare_now": (c_int32, [_D, _P, c_size_t]),
    "stego_corr_fwd_launches": (c_int32, [_D] + [_M] * 4),
    "stego_corr_event_counters": (c_void_p, [_D, _P, c_size_t]),
    "stego_ref_draws": (c_int32, [ctypes.c_uint64, ctypes.c_uint64, c_int32, c_int64, c_int32, c_int32, _P, _P, _P, _P]),
    "stego_ref_draws_advance": (ctypes.c_uint64, [c_int64, c_int32, c_int32, c_int32]),
    "stego_tokens_from_cache": (c_int32, [_P, _P, c_int32, c_int32, c_int32, c_int32, _P, _P, _P]),
    "stego_ref_dropout_masks": (c_int32, [ctypes.c_uint64, ctypes.c_uint64, _P, _P, c_int32, c_int32, c_int64, c_float, _P, _P]),
    "stego_ref_dropout_masks_advance": (ctypes.c_uint64, [c_int64, c_int32, c_int32]),
    "stego_ref_draws_indirect": (c_int32, [_P, _P, ctypes.c_uint64, c_int32, c_int64, c_int32, c_int32, _P, _P, _P, _P]),
    "stego_fast_draws": (c_int32, [_P, ctypes.c_int64, c_int32, c_int32, _P, _P, _P, _P]),
    "stego_finish_draws": (c_int32, [_P, _P, ctypes.c_int64, POINTER(ctypes.c_void_p), c_int32, c_int32, _P, _P, _P, _P]),
    "stego_corr_fwd_profile": (c_int32, [_D] + [_M] * 4 + [_P] * 3 + [_P] * 8 + [_P, c_size_t, _P]
                               + [c_int32, POINTER(c_float)]),
    "stego_corr_bwd_workspace_bytes": (c_size_t, [_D]),
    "stego_corr_helper_bwd_workspace_bytes": (c_size_t, [_D]),
    "stego_corr_bwd": (c_int32, [_D] + [_P] + [_P] * 3 + [_P] * 3 + [_P, _P, _P, c_int32] + [_P] * 3
                       + [_P, _P] + [_P, c_size_t, _P]),
    "stego_corr_helper_fwd": (c_int32, [_D] + [_M] * 4 + [_P] * 5 + [_P, c_size_t, _P]),
    "stego_corr_helper_bwd": (c_int32, [_D] + [_P] * 6 + [_P, _P] + [_P, c_size_t, _P]),
}

_lib = None


def library_path():
    """The in-tree library - or the file STEGO_LIB_PATH names (the sanitizer build of tests/test_asan_host.py; tools)."""
    return os.environ.get("STEGO_LIB_PATH") or _build.LIB_PATH


def load():
    """dlopen the in-tree library (never builds implicitly on a GPU box: build() is explicit)."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(
            "stego_amd: %s is missing - the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc); there is no fallback path." % path)
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError if the .so is stale / symbol missing
        fn.restype = res
        fn.argtypes = args
    if lib.stego_abi_version() != ABI_VERSION:
        raise RuntimeError("stego_amd: ABI version mismatch, rebuild the library")
    _lib = lib
    return lib


KNOBS = {"STEGO_DEBUG": 0, "STEGO_DEBUG_SAMPLE": 1, "STEGO_DEBUG_BWD": 2, "STEGO_DEBUG_VIT": 3, "STEGO_DEBUG_KNN": 4,
         "STEGO_FWD_VARIANT": 5, "STEGO_SHARED_DEVICE": 6}


_shared_device_default = False


def set_shared_device(shared=True):
    """Default of the per-call flag STEGO_FLAG_SHARED_DEVICE for descriptors made by make_desc() from now on: other kernels
    (collectives) run on the device beside the loss, so the fused forward launches one workgroup per tile instead of one per
    compute unit (include/stego_corr.h).  A per-call property of the descriptor since ABI 3 - nothing in the library is global;
    `cfg.shared_device` (ContrastiveCorrelationLoss) or make_desc(..., shared_device=...) override this default."""
    global _shared_device_default
    _shared_device_default = bool(shared)


def debug_set(name, value):
    """Measurement knob of the library (tools only; the library reads the environment once, at load)."""
    _check(load().stego_debug_set(KNOBS[name], int(value)))


def _check(rc):
    if rc != 0:
        _WS_CACHE.clear()            # a kept workspace may hold dirty hand-off words after a failed launch
        _WS_PINNED.clear()
        if _torchglue_mod:           # ... and so may the ones the C++ autograd function keeps
            _torchglue_mod.reset_workspaces()
        msg = load().stego_error_string(rc).decode()
        raise RuntimeError("libstego_corr error %d: %s" % (rc, msg))


def _require_dev(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("stego_amd runs on MI355X only: got a %s tensor (no CPU fallback exists)" % t.device)


def _map(t):
    if t.dtype != torch.float32 or t.dim() != 4:
        raise RuntimeError("expected a float32 [N,C,H,W] tensor, got %s %s" % (t.dtype, tuple(t.shape)))
    s = t.stride()
    return StegoMap(t.data_ptr(), s[0], s[1], s[2], s[3])


def _ptr(t):
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """hipStream_t of torch's current stream on the current device (the raw accessor: torch.cuda.current_stream() costs ~10 us of
    Python per call, three calls per training step)."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


class _on_device:
    """`with torch.cuda.device(dev)` only when dev is not already current (the context manager is ~8 us of Python)."""
    __slots__ = ("guard",)

    def __init__(self, dev):
        idx = dev.index if dev.index is not None else 0
        self.guard = None if (_raw_device is not None and _raw_device() == idx) else torch.cuda.device(dev)

    def __enter__(self):
        if self.guard is not None:
            self.guard.__enter__()

    def __exit__(self, *exc):
        if self.guard is not None:
            return self.guard.__exit__(*exc)
        return False


def make_desc(B, C, K, H, W, S, n_neg, cfg, shifts, precision=PREC_F32, shared_device=None):
    if shared_device is None:
        shared_device = getattr(cfg, "shared_device", None)
    if shared_device is None:
        shared_device = _shared_device_default
    return StegoCorrDesc(B, C, K, H, W, S, n_neg, int(bool(cfg.pointwise)), int(bool(cfg.zero_clamp)),
                         int(bool(cfg.stabalize)), float(shifts[0]), float(shifts[1]), float(shifts[2]),
                         int(precision), FLAG_SHARED_DEVICE if shared_device else 0)


def occupy(n_workgroups, lds_bytes=64 * 1024, microseconds=80, stream=None):
    """Test hook (stego_debug_occupy): a stand-in for a foreign kernel holding compute units while the loss runs."""
    _check(load().stego_debug_occupy(int(n_workgroups), int(lds_bytes), int(microseconds),
                                     stream.cuda_stream if stream is not None else _stream()))


def _dense(t, dtype):
    if t.dtype != dtype:
        t = t.to(dtype)
    return t if t.is_contiguous() else t.contiguous()


def _empty_bytes(n, dev):
    return torch.empty(max(int(n), 16), dtype=torch.uint8, device=dev)


# Forward workspaces are kept (per device, stream and descriptor): the fused forward hands data between workgroups through
# a few counters in the workspace that must be zero when a launch starts and that every launch leaves zero again.  A kept
# workspace is prepared once (stego_corr_workspace_prepare) and then costs neither an allocation nor a memset per call.
_WS_CACHE = {}
_WS_PINNED = set()        # keys of workspaces first used during a stream capture
_SIDE_READY = set()
_WS_CACHE_MAX = 16


def reset_workspaces():
    """Forget the kept forward workspaces (call after a launch that failed: its counters may be dirty)."""
    _WS_CACHE.clear()
    _WS_PINNED.clear()
    if _torchglue_mod:
        _torchglue_mod.reset_workspaces()


def _prepared_ws(lib, desc, dev):
    n = int(lib.stego_corr_workspace_bytes(byref(desc)))
    stream = _stream()
    key = (dev.index, int(stream or 0), n, bytes(desc))
    ws = _WS_CACHE.get(key)
    if ws is None:
        if len(_WS_CACHE) >= _WS_CACHE_MAX:
            # forget the oldest workspace that no captured graph holds (a graph replays with the pointer it captured: its workspace
            # must live as long as this cache does)
            for k in _WS_CACHE:
                if k not in _WS_PINNED:
                    _WS_CACHE.pop(k)
                    break
        ws = _empty_bytes(n, dev)
        with _on_device(dev):
            if torch.cuda.is_current_stream_capturing():
                _WS_PINNED.add(key)
                # a workspace first met while a graph is being captured (the capture stream is a stream of its own): prepared NOW on
                # the library's side stream, not as a memset node that every replay would repeat in front of the forward (launches
                # leave their counters zero, so once is enough)
                _check(lib.stego_corr_workspace_prepare_now(byref(desc), _ptr(ws), ws.numel()))
            else:
                _check(lib.stego_corr_workspace_prepare(byref(desc), _ptr(ws), ws.numel(), stream))
                if dev.index not in _SIDE_READY:       # the library's side stream for this device: created outside any capture
                    _check(lib.stego_corr_workspace_prepare_now(None, None, 0))
                    _SIDE_READY.add(dev.index)
        _WS_CACHE[key] = ws
    return ws


def _fwd_buffers(lib, desc, dev, need_grad, flat=False, keep_ws=False):
    B, S, n_neg = desc.B, desc.S, desc.n_neg
    f32 = dict(dtype=torch.float32, device=dev)
    shp = (S ** 4,) if flat else (S, S, S, S)
    loss_means = torch.empty(3, **f32)
    intra_cd = torch.empty(B, *shp, **f32)
    inter_cd = torch.empty(B, *shp, **f32)
    neg_loss = torch.empty(n_neg * B, *shp, **f32)
    neg_cd = torch.empty(n_neg * B, *shp, **f32)
    saved_w = torch.empty((2 + n_neg) * B, S ** 4, **f32) if need_grad else None
    saved_mean = torch.empty(2 + n_neg, **f32) if need_grad else None
    saved_ctx = _empty_bytes(lib.stego_corr_saved_ctx_bytes(byref(desc)), dev) if need_grad else None
    ws = _prepared_ws(lib, desc, dev) if keep_ws else _empty_bytes(lib.stego_corr_workspace_bytes(byref(desc)), dev)
    return loss_means, intra_cd, inter_cd, neg_loss, neg_cd, saved_w, saved_mean, saved_ctx, ws


def corr_fwd(desc, feats, feats_pos, code, code_pos, coords1, coords2, perms, need_grad):
    """stego_corr_fwd on torch tensors. Returns (loss_means[3], intra_cd, inter_cd, neg_loss, neg_cd,
    saved) where saved = (saved_w, saved_mean, saved_ctx) or None."""
    _require_dev(feats, feats_pos, code, code_pos, coords1, coords2, perms)
    lib = load()
    dev = feats.device
    coords1 = _dense(coords1, torch.float32)
    coords2 = _dense(coords2, torch.float32)
    perms = _dense(perms, torch.int64) if desc.n_neg else None
    (loss_means, intra_cd, inter_cd, neg_loss, neg_cd, saved_w, saved_mean, saved_ctx, ws) = _fwd_buffers(
        lib, desc, dev, need_grad, keep_ws=True)
    mf, mfp, mc, mcp = _map(feats), _map(feats_pos), _map(code), _map(code_pos)
    with _on_device(dev):
        _check(lib.stego_corr_fwd_prepared(byref(desc), byref(mf), byref(mfp), byref(mc), byref(mcp),
                                  _ptr(coords1), _ptr(coords2), _ptr(perms),
                                  _ptr(loss_means), _ptr(intra_cd), _ptr(inter_cd), _ptr(neg_loss), _ptr(neg_cd),
                                  _ptr(saved_w), _ptr(saved_mean), _ptr(saved_ctx), _ptr(ws), ws.numel(), _stream()))
    saved = (saved_w, saved_mean, saved_ctx) if need_grad else None
    return loss_means, intra_cd, inter_cd, neg_loss, neg_cd, saved


def finish_draws(u1, u2, raw_perms, B):
    """(coords1, coords2, perms) from torch.rand x2 and the n_neg torch.randperm results, one launch (stego_finish_draws)."""
    lib = load()
    dev = u1.device
    n_neg = len(raw_perms)
    c1 = torch.empty_like(u1)
    c2 = torch.empty_like(u2)
    perms = torch.empty(n_neg, B, dtype=torch.int64, device=dev)
    arr = (ctypes.c_void_p * max(n_neg, 1))(*[r.data_ptr() for r in raw_perms])
    with _on_device(dev):
        _check(lib.stego_finish_draws(_ptr(u1), _ptr(u2), u1.numel(), arr, n_neg, B, _ptr(c1), _ptr(c2), _ptr(perms), _stream()))
    return c1, c2, perms


_torchglue_mod = None


def torchglue():
    """The in-tree torch extension (csrc/torch_glue_ext.cpp): the loss as a C++ autograd function over the C ABI and the device
    generator's graph-safe Philox state - or None when it has not been built (then the Python autograd.Function runs, and draws under
    stream capture stay the torch calls)."""
    global _torchglue_mod
    if _torchglue_mod is None:
        path = _build.TORCHGLUE_PATH
        if not os.path.exists(path):
            _torchglue_mod = False
        else:
            import importlib.util
            load()
            try:
                spec = importlib.util.spec_from_file_location("_stego_torchglue", path)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                mod.bind(library_path())      # (the SAME file ctypes loaded: with STEGO_LIB_PATH set two copies would each own their globals - ADVICE round 5)
                _torchglue_mod = mod
            except (ImportError, OSError, RuntimeError) as e:      # e.g. a stale build against another torch: the product runs without it
                import warnings
                warnings.warn("stego_amd: %s does not load (%s: %s) - using the Python autograd function; rebuild with "
                              "stego_amd._build.build_torchglue(force=True)" % (path, type(e).__name__, str(e)[:200]))
                _torchglue_mod = False
    return _torchglue_mod or None


_ADVANCE = {}


def ref_draws(gen, shape, n_neg, B, variant, dev):
    """(coords1, coords2, perms) = the reference's torch.rand x 2 / torch.randperm x n_neg draws from ONE launch (stego_ref_draws),
    bit for bit, advancing `gen` (the device's torch.Generator) exactly as the torch calls would.  While the current stream is being
    captured the kernel reads the generator's state where CUDAGraph.replay puts it (stego_ref_draws_indirect; needs torchglue())."""
    ext = torchglue()
    if ext is not None and len(shape) == 4 and shape[1] == shape[2] and shape[3] == 2 and shape[0] == B:
        return ext.ref_draws(gen, B, shape[1], n_neg, variant, dev)
    lib = load()
    c1 = torch.empty(shape, dtype=torch.float32, device=dev)
    c2 = torch.empty(shape, dtype=torch.float32, device=dev)
    perms = torch.empty(n_neg, B, dtype=torch.int64, device=dev)
    with _on_device(dev):
        adv = int(lib.stego_ref_draws_advance(c1.numel(), n_neg, B, variant))
        if ext is not None:
            captured, seed, off, intra = ext.philox_state(gen, adv)
            if captured:
                _check(lib.stego_ref_draws_indirect(seed, off, intra, variant, c1.numel(), n_neg, B, _ptr(c1), _ptr(c2), _ptr(perms), _stream()))
            else:
                _check(lib.stego_ref_draws(seed, off, variant, c1.numel(), n_neg, B, _ptr(c1), _ptr(c2), _ptr(perms), _stream()))
            return c1, c2, perms
        seed, off = gen.initial_seed(), gen.get_offset()
        _check(lib.stego_ref_draws(seed & (2 ** 64 - 1), off, variant, c1.numel(), n_neg, B, _ptr(c1), _ptr(c2), _ptr(perms), _stream()))
    gen.set_offset(off + adv)
    return c1, c2, perms


def ref_dropout_masks(gen, n_masks, numel, keep_prob, variant, dev):
    """n_masks x `x.new_empty(B, C, 1, 1).bernoulli_(keep_prob).div_(keep_prob)` (what nn.Dropout2d draws per call) from ONE launch,
    bit for bit, advancing `gen` like the torch calls -> float32 [n_masks, numel].  Needs torchglue() (the generator's Philox state)."""
    lib = load()
    ext = torchglue()
    if ext is None:
        raise RuntimeError("ref_dropout_masks needs the torch glue extension")
    out = torch.empty(n_masks, numel, dtype=torch.float32, device=dev)
    with _on_device(dev):
        adv = int(lib.stego_ref_dropout_masks_advance(numel, n_masks, variant))
        captured, seed, off, intra = ext.philox_state(gen, adv)
        if captured:
            _check(lib.stego_ref_dropout_masks(0, intra, seed, off, variant, n_masks, numel, float(keep_prob), _ptr(out), _stream()))
        else:
            _check(lib.stego_ref_dropout_masks(seed, off, None, None, variant, n_masks, numel, float(keep_prob), _ptr(out), _stream()))
    return out


def fast_draws(seed, shape, n_neg, B):
    """(coords1, coords2, perms) of the reference's distributions from ONE launch (stego_fast_draws); `seed`: int64 [1] on the
    device, drawn by the caller from the torch generator.  Not the reference's random stream (cfg.fast_draws)."""
    lib = load()
    dev = seed.device
    c1 = torch.empty(shape, dtype=torch.float32, device=dev)
    c2 = torch.empty(shape, dtype=torch.float32, device=dev)
    perms = torch.empty(n_neg, B, dtype=torch.int64, device=dev)
    with _on_device(dev):
        _check(lib.stego_fast_draws(_ptr(seed), c1.numel(), n_neg, B, _ptr(c1), _ptr(c2), _ptr(perms), _stream()))
    return c1, c2, perms


def event_counters(desc, ws):
    """(tiles that gave up waiting for their anchor, negative tiles whose old_mean rendezvous timed out) since the forward workspace `ws`
    was prepared (stego_corr_event_counters: two device words; this copies them - a synchronisation, tools and tests only)."""
    lib = load()
    p = lib.stego_corr_event_counters(byref(desc), _ptr(ws), ws.numel())
    if not p:
        raise RuntimeError("stego_corr_event_counters: bad descriptor or workspace")
    off = int(p) - ws.data_ptr()
    w = ws[off: off + 8].view(torch.int32).cpu()
    return int(w[0]), int(w[1])


def event_counters_total():
    """Sum of event_counters() over every forward workspace this process keeps (the Python wrapper's and the C++ autograd function's):
    (tiles that sampled their anchor themselves, old_mean rendezvous that timed out) since those workspaces were prepared.  Both are zero
    in normal operation; non-zero = the single-launch forward keeps taking its fallback paths (device shared or partitioned without
    cfg.shared_device / STEGO_FLAG_SHARED_DEVICE): correct, ~30 us slower per step.  A synchronisation: call it every few hundred steps."""
    g = r = 0
    seen = []
    for key, ws in list(_WS_CACHE.items()):
        seen.append((key[3], ws))
    mod = torchglue()
    if mod is not None and hasattr(mod, "workspaces"):
        seen += list(mod.workspaces())
    for dbytes, ws in seen:
        desc = StegoCorrDesc.from_buffer_copy(dbytes)
        if desc.S * desc.S > 128:                   # the multi-launch path (csrc/corr_wide.hip) has no hand-off words
            continue
        a, b = event_counters(desc, ws)
        g += a
        r += b
    return g, r


def corr_fwd_launches(desc, feats, feats_pos, code, code_pos):
    """Kernel launches stego_corr_fwd_prepared needs for these maps: 1 = the fused forward, 3 = sample / tile / finalize."""
    lib = load()
    mf, mfp, mc, mcp = _map(feats), _map(feats_pos), _map(code), _map(code_pos)
    return int(lib.stego_corr_fwd_launches(byref(desc), byref(mf), byref(mfp), byref(mc), byref(mcp)))


def corr_fwd_profile(desc, feats, feats_pos, code, code_pos, coords1, coords2, perms, need_grad, iters):
    """stego_corr_fwd_profile: mean HIP-event duration (ms) of (sample kernel, tile kernel, finalize kernel)."""
    _require_dev(feats, feats_pos, code, code_pos, coords1, coords2, perms)
    lib = load()
    dev = feats.device
    (loss_means, intra_cd, inter_cd, neg_loss, neg_cd, saved_w, saved_mean, saved_ctx, ws) = _fwd_buffers(
        lib, desc, dev, need_grad, flat=True)
    mf, mfp, mc, mcp = _map(feats), _map(feats_pos), _map(code), _map(code_pos)
    ms = (c_float * 3)()
    with _on_device(dev):
        _check(lib.stego_corr_fwd_profile(byref(desc), byref(mf), byref(mfp), byref(mc), byref(mcp),
                                          _ptr(coords1), _ptr(coords2), _ptr(perms) if desc.n_neg else None,
                                          _ptr(loss_means), _ptr(intra_cd), _ptr(inter_cd), _ptr(neg_loss),
                                          _ptr(neg_cd), _ptr(saved_w), _ptr(saved_mean), _ptr(saved_ctx),
                                          _ptr(ws), ws.numel(), _stream(), int(iters), ms))
    return ms[0], ms[1], ms[2]


def corr_bwd(desc, code, code_pos, coords1, coords2, perms, saved, intra_cd, inter_cd, neg_cd,
             g_intra, g_inter, g_neg_loss, g_intra_cd, g_inter_cd, g_neg_cd, neg_is_mean=False):
    """stego_corr_bwd. g_* may be None. Returns (d_code, d_code_pos) as [B,K,H,W] views of channels-last buffers.
    neg_is_mean: g_neg_loss is ONE device scalar, the upstream of loss_means[2] (stride -1 of the C ABI)."""
    saved_w, saved_mean, saved_ctx = saved
    _require_dev(code, code_pos, saved_w)
    lib = load()
    dev = code.device
    B, K, H, W = desc.B, desc.K, desc.H, desc.W
    perms = _dense(perms, torch.int64) if desc.n_neg else None
    stride = 1
    if neg_is_mean:
        stride = -1
        g_neg_loss = None if g_neg_loss is None else _dense(g_neg_loss, torch.float32)
    elif g_neg_loss is not None and g_neg_loss.numel() > 0:
        if all(s == 0 for s in g_neg_loss.stride()):
            stride = 0                      # expanded scalar: hand over the one element
        else:
            g_neg_loss = _dense(g_neg_loss, torch.float32)
    else:
        g_neg_loss = None
    g_intra = None if g_intra is None else _dense(g_intra, torch.float32)
    g_inter = None if g_inter is None else _dense(g_inter, torch.float32)
    g_intra_cd = None if g_intra_cd is None else _dense(g_intra_cd, torch.float32)
    g_inter_cd = None if g_inter_cd is None else _dense(g_inter_cd, torch.float32)
    g_neg_cd = None if (g_neg_cd is None or g_neg_cd.numel() == 0) else _dense(g_neg_cd, torch.float32)
    d_code = torch.empty(B, H, W, K, dtype=torch.float32, device=dev)
    d_code_pos = torch.empty(B, H, W, K, dtype=torch.float32, device=dev)
    ws = _empty_bytes(lib.stego_corr_bwd_workspace_bytes(byref(desc)), dev)
    with _on_device(dev):
        _check(lib.stego_corr_bwd(byref(desc), _ptr(perms),
                                  _ptr(saved_w), _ptr(saved_mean), _ptr(saved_ctx),
                                  _ptr(intra_cd), _ptr(inter_cd), _ptr(neg_cd),
                                  _ptr(g_intra), _ptr(g_inter), _ptr(g_neg_loss), stride,
                                  _ptr(g_intra_cd), _ptr(g_inter_cd), _ptr(g_neg_cd),
                                  _ptr(d_code), _ptr(d_code_pos), _ptr(ws), ws.numel(), _stream()))
    return d_code.permute(0, 3, 1, 2), d_code_pos.permute(0, 3, 1, 2)


def helper_fwd(desc, f1, f2, c1, c2, need_grad):
    _require_dev(f1, f2, c1, c2)
    lib = load()
    N, S1, S2 = desc.B, desc.H, desc.W
    dev = f1.device
    f32 = dict(dtype=torch.float32, device=dev)
    loss = torch.empty(N, S1, S2, S1, S2, **f32)
    cd = torch.empty(N, S1, S2, S1, S2, **f32)
    saved_w = torch.empty(N, (S1 * S2) ** 2, **f32) if need_grad else None
    saved_mean = torch.empty(1, **f32) if need_grad else None
    saved_ctx = _empty_bytes(lib.stego_corr_helper_saved_ctx_bytes(byref(desc)), dev) if need_grad else None
    ws = _empty_bytes(lib.stego_corr_helper_workspace_bytes(byref(desc)), dev)
    m1, m2, m3, m4 = _map(f1), _map(f2), _map(c1), _map(c2)
    with _on_device(dev):
        _check(lib.stego_corr_helper_fwd(byref(desc), byref(m1), byref(m2), byref(m3), byref(m4),
                                         _ptr(loss), _ptr(cd), _ptr(saved_w), _ptr(saved_mean), _ptr(saved_ctx),
                                         _ptr(ws), ws.numel(), _stream()))
    return loss, cd, ((saved_w, saved_mean, saved_ctx) if need_grad else None)


def helper_bwd(desc, c1, c2, saved, cd, g_loss, g_cd):
    saved_w, saved_mean, saved_ctx = saved
    _require_dev(c1, c2, saved_w)
    lib = load()
    N, K, S1, S2 = desc.B, desc.K, desc.H, desc.W
    dev = c1.device
    g_loss = None if g_loss is None else _dense(g_loss, torch.float32)
    g_cd = None if g_cd is None else _dense(g_cd, torch.float32)
    d1 = torch.empty(N, S1, S2, K, dtype=torch.float32, device=dev)
    d2 = torch.empty(N, S1, S2, K, dtype=torch.float32, device=dev)
    ws = _empty_bytes(lib.stego_corr_helper_bwd_workspace_bytes(byref(desc)), dev)
    with _on_device(dev):
        _check(lib.stego_corr_helper_bwd(byref(desc), _ptr(saved_w), _ptr(saved_mean),
                                         _ptr(saved_ctx), _ptr(cd), _ptr(g_loss), _ptr(g_cd), _ptr(d1), _ptr(d2),
                                         _ptr(ws), ws.numel(), _stream()))
    return d1.permute(0, 3, 1, 2), d2.permute(0, 3, 1, 2)


def knn_topk(x, k=30, normalize=False, q_begin=0, q_count=None, return_sims=False):
    """All-pairs cosine top-k (reference precompute_knns.py:86-96) of the rows of ``x`` [N, D] (fp32, HIP device).
    Returns int64 [q_count, k] (and the fp32 similarities)."""
    _require_dev(x)
    if x.dim() != 2 or x.dtype != torch.float32:
        raise ValueError("knn_topk expects a 2-D float32 tensor")
    if x.stride(1) != 1:
        x = x.contiguous()
    lib = load()
    N, D = x.shape
    q_count = N - q_begin if q_count is None else q_count
    nws = lib.stego_knn_workspace_bytes(N, D, k, q_count)
    if nws == 0:
        raise RuntimeError("stego_knn_topk: unsupported shape (need 1 <= k <= 32, k <= N < 2^31)")
    dev = x.device
    ws = _empty_bytes(nws, dev)
    idx = torch.empty(q_count, k, dtype=torch.int64, device=dev)
    sims = torch.empty(q_count, k, dtype=torch.float32, device=dev) if return_sims else None
    with _on_device(dev):
        _check(lib.stego_knn_topk(_ptr(x), N, D, x.stride(0), k, 1 if normalize else 0, q_begin, q_count, _ptr(idx), _ptr(sims),
                                  _ptr(ws), ws.numel(), _stream()))
    return (idx, sims) if return_sims else idx


def dense_corr(a, b, normalize=False):
    """tensor_correlation (reference modules.py:283-284), optionally on norm()'ed maps: a [B,C,H1,W1], b [B,C,H2,W2]
    (fp32, HIP device, any strides) -> [B,H1,W1,H2,W2].  Forward only."""
    _require_dev(a, b)
    if a.dim() != 4 or b.dim() != 4 or a.dtype != torch.float32 or b.dtype != torch.float32:
        raise ValueError("dense_corr expects two 4-D float32 maps")
    if a.shape[:2] != b.shape[:2]:
        raise ValueError("dense_corr: batch / channel mismatch %s vs %s" % (tuple(a.shape), tuple(b.shape)))
    lib = load()
    B, C, H1, W1 = a.shape
    H2, W2 = b.shape[2:]
    dev = a.device
    out = torch.empty(B, H1, W1, H2, W2, dtype=torch.float32, device=dev)
    ws = _empty_bytes(lib.stego_dense_corr_workspace_bytes(B, C, H1, W1, H2, W2), dev)
    ma, mb = _map(a), _map(b)
    with _on_device(dev):
        _check(lib.stego_dense_corr(byref(ma), byref(mb), B, C, H1, W1, H2, W2, 1 if normalize else 0, _ptr(out), _ptr(ws),
                                    ws.numel(), _stream()))
    return out


def sample(t, coords, index=None):
    """sample(t[index], coords) of the reference (modules.py:287-288) without the indexed copy of `t`: t [M, C, H, W] (fp32, HIP device, any
    strides), coords [Nc, S, S, 2], index int64 [N] or None (N = M) -> [N, C, S, S] (a view of channels-last rows [N, S*S, C]).  Row n
    uses the coordinates of row n % Nc."""
    _require_dev(t, coords)
    if t.dim() != 4 or t.dtype != torch.float32 or coords.dim() != 4 or coords.shape[-1] != 2 or coords.shape[1] != coords.shape[2]:
        raise ValueError("sample expects a float32 [M, C, H, W] map and [Nc, S, S, 2] coordinates")
    lib = load()
    M, C, H, W = t.shape
    S = int(coords.shape[1])
    N = M if index is None else int(index.numel())
    coords = coords.contiguous().float()
    if index is not None:
        index = index.contiguous().to(torch.int64)
    out = torch.empty(N, S * S, C, dtype=torch.float32, device=t.device)
    m = _map(t)
    with _on_device(t.device):
        _check(lib.stego_sample(byref(m), _ptr(index) if index is not None else None, N, C, H, W, _ptr(coords), int(coords.shape[0]), S,
                                _ptr(out), _stream()))
    return out.view(N, S, S, C).permute(0, 3, 1, 2)


def loss_pointwise_fwd(fd, cd, shifts, cmin, cmax, pointwise, rowsum=None):
    """The elementwise part of helper() (modules.py:330-345) over all pair-sets: fd, cd [n_sets, B, P, P] contiguous fp32 ->
    (neg_loss [n_sets - 2, B, P, P], per-set loss sums [n_sets], rowsum [n_sets * B * P], old_mean [n_sets]).  rowsum: the row sums of fd when
    the correlation kernel already took them (dense_corr_panels(want_rowsum=True))."""
    _require_dev(fd, cd)
    lib = load()
    n_sets, B, P = int(fd.shape[0]), int(fd.shape[1]), int(fd.shape[2])
    dev = fd.device
    rows = n_sets * B * P
    have_rowsum = rowsum is not None
    if not have_rowsum:
        rowsum = torch.empty(rows, dtype=torch.float32, device=dev)
    sh = (c_float * 3)(*[float(x) for x in shifts])
    neg_loss = torch.empty(max(n_sets - 2, 0), B, P, P, dtype=torch.float32, device=dev)
    loss_rowsum = torch.empty(rows, dtype=torch.float32, device=dev)
    with _on_device(dev):
        if pointwise:
            if not have_rowsum:
                _check(lib.stego_rowsum(_ptr(fd), rows, P, _ptr(rowsum), _stream()))
            old_mean = rowsum.view(n_sets, -1).sum(1) / float(B * P * P)
        else:
            old_mean = torch.zeros(n_sets, dtype=torch.float32, device=dev)
        _check(lib.stego_loss_pointwise_fwd(_ptr(fd), _ptr(cd), _ptr(rowsum), _ptr(old_mean), n_sets, B, P, sh, float(cmin), float(cmax),
                                            1 if pointwise else 0, _ptr(neg_loss) if n_sets > 2 else None, _ptr(loss_rowsum), _stream()))
    return neg_loss, loss_rowsum.view(n_sets, -1).sum(1), rowsum, old_mean


def loss_pointwise_bwd(fd, cd, rowsum, old_mean, shifts, cmin, cmax, pointwise, g_neg, g_sums):
    """d cd of loss_pointwise_fwd: g_neg (upstream of neg_loss: dense, an expanded scalar, or None), g_sums (upstream of the sums, or None)."""
    lib = load()
    n_sets, B, P = int(fd.shape[0]), int(fd.shape[1]), int(fd.shape[2])
    sh = (c_float * 3)(*[float(x) for x in shifts])
    g_cd = torch.empty_like(cd)
    dense = bcast = None
    if g_neg is not None and g_neg.numel() > 0:
        if all(st == 0 for st in g_neg.stride()):
            bcast = g_neg                                   # ONE value behind every element (the backward of .mean() / .sum())
        else:
            dense = g_neg.contiguous().float()
    if g_sums is not None:
        g_sums = g_sums.contiguous().float()
    with _on_device(fd.device):
        _check(lib.stego_loss_pointwise_bwd(_ptr(fd), _ptr(cd), _ptr(rowsum), _ptr(old_mean), n_sets, B, P, sh, float(cmin), float(cmax),
                                            1 if pointwise else 0, _ptr(dense) if dense is not None else None,
                                            _ptr(bcast) if bcast is not None else None, _ptr(g_sums) if g_sums is not None else None,
                                            _ptr(g_cd), _stream()))
    return g_cd


def sample_bwd(g_out, like, coords, index=None):
    """Adjoint of sample(): g_out [N, C, S, S] -> the gradient of the map (shape and memory format of `like`), atomic fp32 adds."""
    _require_dev(g_out, coords)
    lib = load()
    M, C, H, W = like.shape
    S = int(coords.shape[1])
    N = M if index is None else int(index.numel())
    g = g_out.permute(0, 2, 3, 1).contiguous()                     # [N, S, S, C] rows
    coords = coords.contiguous().float()
    if index is not None:
        index = index.contiguous().to(torch.int64)
    d = torch.zeros_like(like, dtype=torch.float32)               # (preserves the strides of a dense `like`: channels-last stays channels-last)
    m = _map(d)
    with _on_device(like.device):
        _check(lib.stego_sample_bwd(_ptr(g), byref(m), _ptr(index) if index is not None else None, N, C, H, W, _ptr(coords),
                                    int(coords.shape[0]), S, _stream()))
    return d


def sample_bwd_rows(g_rows, d_map, coords, index=None, rows_n=None, inv=None):
    """Adjoint of sample() for gradients that are already channels-last rows: g_rows [N, P, C] contiguous is ADDED into d_map (the map's
    gradient, fp32, any strides, zeroed by the caller).  With rows_n / inv (the normalised rows and 1 / max(|row|, eps) stego_sample_panels
    saved) g_rows is the gradient of the NORMALISED rows and the backward of norm() (F.normalize, modules.py:275-276) is applied on the way:
    d row = inv * (g - rows_n <rows_n, g>), or inv * g where the row's norm was below eps."""
    _require_dev(g_rows, d_map)
    lib = load()
    M, C, H, W = d_map.shape
    S = int(coords.shape[1])
    N = int(g_rows.shape[0])
    if g_rows.dim() != 3 or not g_rows.is_contiguous() or g_rows.shape[1] != S * S or g_rows.shape[2] != C or g_rows.dtype != torch.float32:
        raise ValueError("sample_bwd_rows expects contiguous float32 [N, S*S, C] rows")
    if (rows_n is None) != (inv is None) or (rows_n is not None and (not rows_n.is_contiguous() or not inv.is_contiguous()
                                                                   or rows_n.shape != g_rows.shape or inv.numel() != N * S * S)):
        raise ValueError("sample_bwd_rows: rows_n [N, P, C] and inv [N, P] go together, contiguous")
    coords = coords.contiguous().float()
    if index is not None:
        index = index.contiguous().to(torch.int64)
        if index.numel() != N:
            raise ValueError("sample_bwd_rows: one index per row image")
    elif N != M:
        raise ValueError("sample_bwd_rows: %d row images for a map of %d" % (N, M))
    m = _map(d_map)
    with _on_device(d_map.device):
        _check(lib.stego_sample_bwd_rows(_ptr(g_rows), _ptr(rows_n) if rows_n is not None else None, _ptr(inv) if inv is not None else None,
                                         byref(m), _ptr(index) if index is not None else None, N, C, H, W, _ptr(coords),
                                         int(coords.shape[0]), S, _stream()))
    return d_map


class PanelSet:
    """Prepared operands of the dense-correspondence kernel for `images` point sets of P points x C channels (include/stego_corr.h,
    stego_sample_panels): split-fp16 operand images + one scale per row."""

    def __init__(self, images, C, P, device):
        lib = load()
        self.images, self.C, self.P = int(images), int(C), int(P)
        self.image_bytes = int(lib.stego_panel_image_bytes(self.C, self.P))
        if self.image_bytes <= 0:
            raise ValueError("PanelSet: bad shape C = %d, P = %d" % (C, P))
        self.rows = (self.P + 127) // 128 * 128
        self.panels = torch.empty(self.images * self.image_bytes, dtype=torch.uint8, device=device)
        self.row_scale = torch.empty(self.images, self.rows, dtype=torch.float32, device=device)


def sample_panels(pset, first, t, coords, index=None, normalize=True, rows_out=None, inv_out=None):
    """Images [first, first + N) of `pset` = norm(sample(t if index is None else t[index], coords)) (modules.py:275-276, 287-288, 384-385) as
    operands of dense_corr_panels; rows_out [N, P, C] / inv_out [N, P] (contiguous fp32, optional) receive the normalised rows and
    1 / max(|row|, eps)."""
    _require_dev(t, coords)
    if t.dim() != 4 or t.dtype != torch.float32 or coords.dim() != 4 or coords.shape[-1] != 2 or coords.shape[1] != coords.shape[2]:
        raise ValueError("sample_panels expects a float32 [M, C, H, W] map and [Nc, S, S, 2] coordinates")
    lib = load()
    M, C, H, W = t.shape
    S = int(coords.shape[1])
    N = M if index is None else int(index.numel())
    if C != pset.C or S * S != pset.P or first < 0 or first + N > pset.images:
        raise ValueError("sample_panels: the panel set holds %d images of %d x %d" % (pset.images, pset.P, pset.C))
    for o, shape in ((rows_out, (N, S * S, C)), (inv_out, (N, S * S))):
        if o is not None and (tuple(o.shape) != shape or not o.is_contiguous() or o.dtype != torch.float32):
            raise ValueError("sample_panels: optional outputs are contiguous float32 %s" % (shape,))
    coords = coords.contiguous().float()
    if index is not None:
        index = index.contiguous().to(torch.int64)
    m = _map(t)
    with _on_device(t.device):
        _check(lib.stego_sample_panels(byref(m), _ptr(index) if index is not None else None, N, C, H, W, _ptr(coords), int(coords.shape[0]), S,
                                       1 if normalize else 0, pset.panels.data_ptr() + first * pset.image_bytes, _ptr(pset.row_scale[first:]),
                                       _ptr(rows_out) if rows_out is not None else None, _ptr(inv_out) if inv_out is not None else None, _stream()))


def dense_corr_panels(pa, images_a, pb, N, want_rowsum=False):
    """out[n] = (image n % images_a of pa) . (image n of pb)^T, [N, P, P] fp32: tensor_correlation (modules.py:283-284) of one set of anchors
    against the second operands of N / images_a pair-sets.  want_rowsum: also returns sum_j out[n, i, j] as [N * P]."""
    lib = load()
    if pa.C != pb.C or images_a > pa.images or N > pb.images or images_a < 1 or N < 1:
        raise ValueError("dense_corr_panels: operand sets do not match")
    out = torch.empty(N, pa.P, pb.P, dtype=torch.float32, device=pa.panels.device)
    rowsum = torch.empty(N * pa.P, dtype=torch.float32, device=out.device) if want_rowsum else None
    with _on_device(out.device):
        _check(lib.stego_dense_corr_panels(_ptr(pa.panels), _ptr(pa.row_scale), images_a, _ptr(pb.panels), _ptr(pb.row_scale), N, pa.C, pa.P, pb.P,
                                           _ptr(out), _ptr(rowsum), _stream()))
    return (out, rowsum) if want_rowsum else out


# ------------------------------------------------------------------ segmentation head (include/stego_head.h)
def head_desc(tokens, K, nonlinear):
    """tokens: fp32 [B, HW, C] view with a dense channel axis (e.g. feat[:, 1:, :] of the backbone: the class token is skipped by
    the view's offset, the image stride stays (1 + HW) * C)."""
    if tokens.dim() != 3 or tokens.dtype != torch.float32 or tokens.stride(2) != 1:
        raise RuntimeError("head: expected a float32 [B, HW, C] token tensor with contiguous channels")
    B, HW, C = tokens.shape
    return StegoHeadDesc(B, HW, C, int(K), 1 if nonlinear else 0, tokens.stride(1), tokens.stride(0) if B > 1 else HW * tokens.stride(1), None)


def tokens_from_cache(table, index, skip_rows=1):
    """stego_tokens_from_cache: float32 [n, ntok, D] rows `index` of the fp16 token table + a device word (int32 [1]) with the float bits
    of the largest magnitude of the rows >= skip_rows (what StegoHeadDesc.tokens_amax takes)."""
    _require_dev(table, index)
    if table.dtype != torch.float16 or table.dim() != 3 or not table.is_contiguous():
        raise RuntimeError("tokens_from_cache: expected a contiguous float16 [items, ntok, D] table")
    lib = load()
    dev = table.device
    index = _dense(index, torch.int64)
    n, ntok, D = int(index.numel()), table.shape[1], table.shape[2]
    out = torch.empty(n, ntok, D, dtype=torch.float32, device=dev)
    amax = torch.empty(1, dtype=torch.int32, device=dev)
    with _on_device(dev):
        _check(lib.stego_tokens_from_cache(_ptr(table), _ptr(index), n, ntok, D, int(skip_rows), _ptr(out), _ptr(amax), _stream()))
    return out, amax


def head_fwd(tokens, masks, w1, b1, w21, b21, w22, b22, need_grad, want_feats, tokens_amax=None):
    """stego_head_fwd.  masks = (m1, m2, m3) fp32 [B, C] each or None.  Returns (code [B, HW, K], feats_out [B, HW, C] or None,
    saved_h or None)."""
    _require_dev(tokens, w1, b1)
    lib = load()
    dev = tokens.device
    nonlinear = w21 is not None
    d = head_desc(tokens, w1.shape[0], nonlinear)
    if tokens_amax is not None:                  # (the magnitude word of exactly these tokens: tokens_from_cache)
        d.tokens_amax = tokens_amax.data_ptr()
    B, HW, C, K = d.B, d.HW, d.C, d.K
    code = torch.empty(B, HW, K, dtype=torch.float32, device=dev)
    feats = torch.empty(B, HW, C, dtype=torch.float32, device=dev) if want_feats else None
    saved_h = torch.empty(B * HW * C + 8, dtype=torch.float32, device=dev) if (nonlinear and need_grad) else None
    nws = int(lib.stego_head_fwd_workspace_bytes(byref(d)))
    if nws == 0:
        raise RuntimeError("stego_head_fwd: unsupported shape (C a multiple of 32, K <= 128, 16-byte token rows)")
    if nonlinear and saved_h is not None:       # H lives in saved_h: the workspace only holds the scale words and the split weights
        nws -= (B * HW * C * 4 + 255) // 256 * 256
    ws = _empty_bytes(nws, dev)
    m1, m2, m3 = masks if masks is not None else (None, None, None)
    with _on_device(dev):
        _check(lib.stego_head_fwd(byref(d), _ptr(tokens), _ptr(m1), _ptr(m2), _ptr(m3), _ptr(w1), _ptr(b1), _ptr(w21), _ptr(b21),
                                  _ptr(w22), _ptr(b22), _ptr(code), _ptr(feats), _ptr(saved_h), _ptr(ws), ws.numel(), _stream()))
    return code, feats, saved_h


def head_bwd(tokens, masks, saved_h, w22, d_code, K):
    """stego_head_bwd -> (dw1, db1, dw21, db21, dw22, db22) (the cluster2 entries None for a linear head)."""
    _require_dev(tokens, d_code)
    lib = load()
    dev = tokens.device
    nonlinear = saved_h is not None
    d = head_desc(tokens, K, nonlinear)
    C = d.C
    f32 = dict(dtype=torch.float32, device=dev)
    dw1, db1 = torch.empty(K, C, **f32), torch.empty(K, **f32)
    dw21 = db21 = dw22 = db22 = None
    if nonlinear:
        dw21, db21, dw22, db22 = torch.empty(C, C, **f32), torch.empty(C, **f32), torch.empty(K, C, **f32), torch.empty(K, **f32)
    d_code = _dense(d_code, torch.float32)
    ws = _empty_bytes(lib.stego_head_bwd_workspace_bytes(byref(d)), dev)
    m1, m2 = (masks[0], masks[1]) if masks is not None else (None, None)
    with _on_device(dev):
        _check(lib.stego_head_bwd(byref(d), _ptr(tokens), _ptr(m1), _ptr(m2), _ptr(saved_h), _ptr(w22), _ptr(d_code),
                                  _ptr(dw1), _ptr(db1), _ptr(dw21), _ptr(db21), _ptr(dw22), _ptr(db22), _ptr(ws), ws.numel(), _stream()))
    return dw1, db1, dw21, db21, dw22, db22
