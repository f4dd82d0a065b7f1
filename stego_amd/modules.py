"""Host-side mirror of the reference's ``src/modules.py`` hot-path surface.

Same names, argument meaning and return values as the reference so that
``train_segmentation.py`` is a drop-in:

* ``ContrastiveCorrelationLoss(cfg)`` - ``.forward`` (reference modules.py:349-398) and
  ``.helper`` (:325-347) run on the hand-written HIP kernels in ``csrc/`` through the C ABI
  (``include/stego_corr.h``); the RNG draws (coords1, coords2, super_perm x neg_samples) are
  made here with torch in the reference's order (:366, :367, :383) so loss curves overlap.
* ``norm`` :275, ``average_norm`` :279, ``tensor_correlation`` :283, ``sample`` :287,
  ``super_perm`` :291, ``sample_nonzero_locations`` :298 - thin torch-on-device helpers the
  other reference scripts import by name (they are not on the hot path; the fused kernel
  does its own sampling/normalisation/contraction).

There is no CPU / pure-PyTorch fallback for the loss: non-HIP tensors raise.
"""
import logging
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd.function import once_differentiable

from . import capi
from .featurizers import (ClusterLookup, ContrastiveCRFLoss, Decoder, DinoFeaturizer, DoubleConv,  # noqa: F401
                          FeaturePyramidNet, LambdaLayer, NetWithActivations, ResizeAndClassify)


# ------------------------------------------------------------------ small named helpers
def norm(t):
    """reference modules.py:275-276."""
    return F.normalize(t, dim=1, eps=1e-10)


def average_norm(t):
    """reference modules.py:279-280."""
    return t / t.square().sum(1, keepdim=True).sqrt().mean()


class _SampleFunction(torch.autograd.Function):
    """sample(t[index], coords) on the native gather (stego_sample) with its adjoint (stego_sample_bwd): no permuted copy of the maps."""

    @staticmethod
    def forward(ctx, t, coords, index):
        ctx.save_for_backward(coords, index if index is not None else coords.new_empty(0))
        ctx.has_index = index is not None
        ctx.like = t
        return capi.sample(t.detach(), coords, index)

    @staticmethod
    def backward(ctx, g):
        coords, index = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            return None, None, None
        return capi.sample_bwd(g, ctx.like, coords, index if ctx.has_index else None), None, None


def sample_indexed(t, coords, index=None):
    """sample(t if index is None else t[index], coords.repeat(...)) (modules.py:287-288, :384-385): native on a HIP device (no copy of the
    permuted maps; row n of the result uses coords[n % len(coords)]), the reference's expression elsewhere."""
    if t.is_cuda and t.dtype == torch.float32 and coords.shape[1] == coords.shape[2]:
        return _SampleFunction.apply(t, coords, index)
    if index is not None:
        t = t[index]
        coords = coords.repeat(t.shape[0] // coords.shape[0], 1, 1, 1)
    return sample(t, coords)


class _PointwiseLossFunction(torch.autograd.Function):
    """helper()'s elementwise part (modules.py:330-345) for all pair-sets in three launches (stego_rowsum, stego_loss_pointwise_fwd /
    _bwd): (fd [sets, B, P, P] - no gradient, the reference computes it under no_grad -, cd [sets, B, P, P]) ->
    (neg_loss [sets - 2, B, P, P], the sum of every set's loss [sets])."""

    @staticmethod
    def forward(ctx, fd, cd, shifts, cmin, cmax, pointwise, fd_rowsum=None):
        fd, cdc = fd.contiguous(), cd.detach().contiguous()
        neg_loss, sums, rowsum, old_mean = capi.loss_pointwise_fwd(fd, cdc, shifts, cmin, cmax, pointwise, fd_rowsum)
        ctx.save_for_backward(fd, cdc, rowsum, old_mean)
        ctx.args = (shifts, cmin, cmax, pointwise)
        return neg_loss, sums

    @staticmethod
    def backward(ctx, g_neg, g_sums):
        fd, cdc, rowsum, old_mean = ctx.saved_tensors
        shifts, cmin, cmax, pointwise = ctx.args
        return None, capi.loss_pointwise_bwd(fd, cdc, rowsum, old_mean, shifts, cmin, cmax, pointwise, g_neg, g_sums), None, None, None, None, None


class _CodeCorrFunction(torch.autograd.Function):
    """cd of ALL pair-sets for shapes the fused kernels do not take (modules.py:335 with :369-385 in front of it):
        image n of the operand set = norm(sample(orig_code, coords1)) [0, B) | norm(sample(orig_code_pos, coords2)) [B, 2 B) |
                                     norm(sample(orig_code[perm_k], coords2)) [2 B + k B, ...)
        cd[n] = rows(n % B) . rows(n)^T
    Forward: three sampling launches that write the dense kernel's operands directly (stego_sample_panels) + one correlation launch.
    Backward: two batched GEMMs against the saved normalised rows, then the backward of norm() and of the sampling in one scatter launch
    per source (stego_sample_bwd_rows)."""

    @staticmethod
    def forward(ctx, orig_code, orig_code_pos, coords1, coords2, idx):
        B, K = orig_code.shape[:2]
        P = int(coords1.shape[1]) ** 2
        n_img = 2 * B + (int(idx.numel()) if idx is not None else 0)
        dev = orig_code.device
        oc, ocp = orig_code.detach(), orig_code_pos.detach()
        pset = capi.PanelSet(n_img, K, P, dev)
        cn = torch.empty(n_img, P, K, dtype=torch.float32, device=dev)
        inv = torch.empty(n_img, P, dtype=torch.float32, device=dev)
        capi.sample_panels(pset, 0, oc, coords1, None, True, cn[:B], inv[:B])
        capi.sample_panels(pset, B, ocp, coords2, None, True, cn[B:2 * B], inv[B:2 * B])
        if n_img > 2 * B:
            capi.sample_panels(pset, 2 * B, oc, coords2, idx, True, cn[2 * B:], inv[2 * B:])
        ctx.save_for_backward(cn, inv, coords1, coords2, idx if idx is not None else coords1.new_empty(0))
        ctx.has_idx = idx is not None
        ctx.like = (orig_code, orig_code_pos)
        ctx.set_materialize_grads(False)
        return capi.dense_corr_panels(pset, B, pset, n_img)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        if g is None:
            return None, None, None, None, None
        cn, inv, coords1, coords2, idx = ctx.saved_tensors
        orig_code, orig_code_pos = ctx.like
        n_img, P, K = cn.shape
        B = orig_code.shape[0]
        n_sets = n_img // B
        g = g.contiguous().view(n_sets, B, P, P)
        # the gradient of every image's normalised rows: as a second operand (g^T . anchors), the anchors' also as the first (sum over the sets)
        d_rows = torch.matmul(g.transpose(2, 3), cn[:B]).view(n_img, P, K)
        d_rows[:B] += torch.bmm(g.view(n_img, P, P), cn).view(n_sets, B, P, K).sum(0)
        d_code = d_pos = None
        if ctx.needs_input_grad[0]:
            d_code = torch.zeros_like(orig_code, dtype=torch.float32)
            capi.sample_bwd_rows(d_rows[:B], d_code, coords1, None, cn[:B], inv[:B])
            if ctx.has_idx:
                capi.sample_bwd_rows(d_rows[2 * B:], d_code, coords2, idx, cn[2 * B:], inv[2 * B:])
        if ctx.needs_input_grad[1]:
            d_pos = torch.zeros_like(orig_code_pos, dtype=torch.float32)
            capi.sample_bwd_rows(d_rows[B:2 * B], d_pos, coords2, None, cn[B:2 * B], inv[B:2 * B])
        return d_code, d_pos, None, None, None


class _DenseCorrFunction(torch.autograd.Function):
    """tensor_correlation with gradients, all three contractions on the native dense-correspondence kernel:
        out[n,h,w,i,j] = sum_c a[n,c,h,w] b[n,c,i,j]
        dA[n,c,h,w]    = sum_ij g[n,h,w,i,j] b[n,c,i,j]   = tensor_correlation(g seen as a map with channels (i,j), b^T)
        dB[n,c,i,j]    = sum_hw g[n,h,w,i,j] a[n,c,h,w]   = tensor_correlation(g seen as a map with channels (h,w), a^T)
    The adjoints are the same kernel on strided VIEWS (StegoMap carries element strides): nothing is transposed in memory."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        ctx.set_materialize_grads(False)
        return capi.dense_corr(a.detach(), b.detach())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        if g is None:
            return None, None
        a, b = a.detach(), b.detach()
        N, C, H1, W1 = a.shape
        H2, W2 = b.shape[2:]
        g = g.contiguous().view(N, H1 * W1, H2 * W2)
        da = db = None
        if C <= 128 and H1 * W1 <= 1024 and H2 * W2 <= 1024:
            # the adjoints of a CODE correlation over sampled points (the loss at cfg.feature_samples > 11): N small batched GEMMs
            # [C x P] . [P x P] - plain library GEMMs (rocBLAS through torch.bmm).  On the native kernel each first re-lays its "map" g -
            # 59 MB at S = 16 - into fp16 operand panels: 0.25 ms per adjoint for 2 GFLOP (tools/exp/generic_loop.py)
            if ctx.needs_input_grad[0]:
                da = torch.bmm(b.flatten(2), g.transpose(1, 2)).view(N, C, H1, W1)
            if ctx.needs_input_grad[1]:
                db = torch.bmm(a.flatten(2), g).view(N, C, H2, W2)
            return da, db
        if ctx.needs_input_grad[0]:
            gq = g.permute(0, 2, 1).unflatten(2, (H1, W1))             # [N, (i,j), H1, W1]: "channels" = positions of b
            bt = b.flatten(2).permute(0, 2, 1).unsqueeze(-1)           # [N, (i,j), C, 1]
            da = capi.dense_corr(gq, bt).squeeze(-1).permute(0, 3, 1, 2)        # [N, H1, W1, C, 1] -> [N, C, H1, W1]
        if ctx.needs_input_grad[1]:
            gp = g.unflatten(2, (H2, W2))                              # [N, (h,w), H2, W2]: "channels" = positions of a
            at = a.flatten(2).permute(0, 2, 1).unsqueeze(-1)           # [N, (h,w), C, 1]
            db = capi.dense_corr(gp, at).squeeze(-1).permute(0, 3, 1, 2)        # [N, H2, W2, C, 1] -> [N, C, H2, W2]
        return da, db


def tensor_correlation(a, b):
    """reference modules.py:283-284.  float32 4-D maps on a HIP device run on the native dense-correspondence kernel
    (stego_dense_corr), with gradients (both adjoints are the same kernel on strided views: _DenseCorrFunction); anything
    else (CPU tensors - the reference's own path for tests / plotting - or other dtypes) is the reference's einsum.  This
    helper is not on the training hot path: the fused loss has its own contraction."""
    if a.is_cuda and b.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.dim() == 4 and b.dim() == 4:
        if torch.is_grad_enabled() and (a.requires_grad or b.requires_grad):
            return _DenseCorrFunction.apply(a, b)
        return capi.dense_corr(a, b)
    return torch.einsum("nchw,ncij->nhwij", a, b)


def sample(t: torch.Tensor, coords: torch.Tensor):
    """reference modules.py:287-288."""
    return F.grid_sample(t, coords.permute(0, 2, 1, 3), padding_mode='border', align_corners=True)


def _unfix(perm):
    """`perm[perm == arange] += 1; perm % size` of modules.py:293-295 as pure elementwise ops: the reference's boolean
    mask indexing makes the host wait for the device (nonzero) on every call - 5 calls per step were 0.7 ms of CPU time
    per step here, more than the whole loss on the GPU.  Same values."""
    size = perm.shape[-1]
    return (perm + (perm == torch.arange(size, device=perm.device)).to(perm.dtype)) % size


def super_perm(size: int, device: torch.device):
    """reference modules.py:291-295 (TorchScript there; same draws from the same generator)."""
    return _unfix(torch.randperm(size, device=device, dtype=torch.long))


def sample_nonzero_locations(t, target_size):
    """reference modules.py:298-311 (salience-guided coords; cfg.use_salience, off by default)."""
    nonzeros = torch.nonzero(t)
    coords = torch.zeros(target_size, dtype=nonzeros.dtype, device=nonzeros.device)
    n = target_size[1] * target_size[2]
    for i in range(t.shape[0]):
        selected_nonzeros = nonzeros[nonzeros[:, 0] == i]
        if selected_nonzeros.shape[0] == 0:
            selected_coords = torch.randint(t.shape[1], size=(n, 2), device=nonzeros.device)
        else:
            selected_coords = selected_nonzeros[torch.randint(len(selected_nonzeros), size=(n,)), 1:]
        coords[i, :, :, :] = selected_coords.reshape(target_size[1], target_size[2], 2)
    coords = coords.to(torch.float32) / t.shape[1]
    coords = coords * 2 - 1
    return torch.flip(coords, dims=[-1])


# ----------------------------------------------------------------------- autograd glue
_backend = capi      # tests may swap this for an oracle-backed double to exercise host logic on CPU

# (device index, numel of a coords tensor, n_neg, B) -> variant of stego_ref_draws that reproduces the installed torch, or -1
_REF_DRAW_VARIANTS = {}


def _device_generator(dev):
    if not torch.cuda.default_generators:            # (filled by the lazy CUDA / HIP initialisation)
        torch.cuda.init()
    return torch.cuda.default_generators[dev.index if dev.index is not None else torch.cuda.current_device()]


def _torch_draws(shape, n, B, dev):
    """The reference's draws as the reference makes them (modules.py:366-367, :383): seven torch calls on the device generator."""
    out = [torch.rand(shape, device=dev), torch.rand(shape, device=dev)]
    out += [torch.randperm(B, device=dev, dtype=torch.long) for _ in range(n)]
    return out


def ref_draw_variant(shape, n, B, dev):
    """Which variant of the one-launch draws (stego_ref_draws) is bit-identical to THIS torch build's rand / randperm for these sizes -
    found by running both from the same generator state once (the generator is left where it was); -1 = none (the torch calls stay).
    The kernel restates ATen internals; this check is what makes relying on them safe across torch versions."""
    numel = 1
    for s_ in shape:
        numel *= int(s_)
    key = (dev.index, numel, int(n), int(B))
    v = _REF_DRAW_VARIANTS.get(key)
    if v is not None:
        return v
    v = -1
    gen = _device_generator(dev)
    try:
        state = gen.get_state()
        try:
            ref = _torch_draws(shape, n, B, dev)
            off_ref = gen.get_offset()
            want1, want2 = ref[0] * 2 - 1, ref[1] * 2 - 1
            want_p = _unfix(torch.stack(ref[2:])) if n else None
            for cand in range(8):
                gen.set_state(state)
                c1, c2, perms = _backend.ref_draws(gen, shape, n, B, cand, dev)
                if gen.get_offset() == off_ref and torch.equal(c1, want1) and torch.equal(c2, want2) and \
                        (n == 0 or torch.equal(perms, want_p)):
                    v = cand
                    break
        finally:
            gen.set_state(state)
    except (RuntimeError, AttributeError):
        v = -1
    _REF_DRAW_VARIANTS[key] = v
    _report_variant("the RNG draws of ContrastiveCorrelationLoss.forward (torch.rand x 2 + torch.randperm x %d, B = %d)" % (n, B),
                    "stego_ref_draws", v, "~%d tiny launches per step instead of one" % (6 + 5 * int(n)))
    return v


_REPORTED_FALLBACKS = set()


def _report_variant(what, kernel, variant, cost):
    """Says once per kind which restatement of the installed torch's generator arithmetic was selected - or, as a WARNING, that none
    matched and the torch calls stay (a torch upgrade must not return the product path to dozens of launches per step silently)."""
    log = logging.getLogger("stego_amd")
    if variant >= 0:
        log.info("%s: variant %d of %s reproduces torch %s bit for bit (checked once per size)", what, variant, kernel, torch.__version__)
    elif (what, kernel) not in _REPORTED_FALLBACKS:
        _REPORTED_FALLBACKS.add((what, kernel))
        warnings.warn("stego_amd: no variant of %s reproduces torch %s for %s - the torch calls are kept (same numbers, %s)"
                      % (kernel, torch.__version__, what, cost))


def _usable_ref_draw_variant(shape, n, B, dev):
    """ref_draw_variant(), or - while the current stream is being captured, where the self-check cannot run - its cached answer, provided
    the generator's graph-safe state is reachable (capi.torchglue())."""
    if not torch.cuda.is_current_stream_capturing():
        return ref_draw_variant(shape, n, B, dev)
    numel = 1
    for s_ in shape:
        numel *= int(s_)
    if getattr(_backend, "torchglue", lambda: None)() is None:
        return -1
    return _REF_DRAW_VARIANTS.get((dev.index, numel, int(n), int(B)), -1)


def _native_autograd(cfg):
    """The C++ autograd function of the loss (capi.torchglue()) unless cfg.native_autograd is False, the module has not been built, or a
    test has swapped the backend: then the Python autograd.Function classes below run - same C ABI calls, same results."""
    if _backend is not capi or not getattr(cfg, "native_autograd", True):
        return None
    return capi.torchglue()


def _precision_of(cfg):
    name = getattr(cfg, "corr_precision", "f16x3")
    if name in ("f32", "fp32", 0):
        return capi.PREC_F32
    if name in ("f16x3", "bf16x3", 1):          # "bf16x3": the split mode's former name
        return capi.PREC_F16X3
    raise ValueError("unknown corr_precision %r" % (name,))


def as_channels_last(t, min_numel=1 << 16):
    """Host-side layout policy.  DinoFeaturizer hands over channels-last strided views (modules.py:97), which the
    kernels read with one coalesced run per bilinear tap.  In an NCHW-contiguous map a tap is C isolated 4-byte
    words (measured 2.4x slower end to end), so large NCHW maps are re-laid out once (one device copy, ~35 us for
    BASELINE config 2); small ones go through the kernels' generic strided path as they are."""
    if t.dim() != 4 or t.stride(1) == 1 or t.numel() < min_numel:
        return t
    return t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)


def _relayout_threshold(code):
    """Code dimensions above 72 exist on the fused (channels-last) path only: such maps are always re-laid out, however small."""
    return 0 if code.shape[1] > 72 else 1 << 16


class _LossFamily:
    """The three scalar outputs of ONE forward call - pos_intra_loss.mean(), pos_inter_loss.mean(), neg_inter_loss.mean(), as the
    autograd outputs of the loss op - and the device constants a weighted sum of them needs as upstream gradients."""
    _const = {}
    _pinned = set()
    __slots__ = ("outs", "means")

    def __init__(self, o_intra, o_inter, o_neg, means=None):
        self.outs = (o_intra, o_inter, o_neg)
        self.means = means          # the same three scalars as the op's ONE vector output [3] (the C++ autograd function's), or None

    _CONST_MAX = 256                  # cached device constants (per value / triple, device, dtype)

    @classmethod
    def _put(cls, key, make, like):
        """A bounded cache of device constants whose tensors are NEVER replaced while alive: at the bound the least recently used entry's
        VALUE is overwritten in place (copy_) and the entry re-keyed - a captured graph that baked the old tensor's address in as an upstream
        gradient keeps a valid pointer; entries made or used during a stream capture are pinned (never re-used), as the workspace cache
        does (_WS_PINNED).  (scheduled loss weights changed the key every step and the cache grew without bound.)"""
        t = cls._const.get(key)
        capturing = torch.cuda.is_available() and like.is_cuda and torch.cuda.is_current_stream_capturing()
        if t is not None:
            cls._const[key] = cls._const.pop(key)            # most recently used last
            if capturing:
                cls._pinned.add(key)
            return t
        if capturing:
            return None                    # (no allocation + fill inside a capture: the caller takes the plain path)
        if len(cls._const) >= cls._CONST_MAX:
            victim = next((k for k in cls._const if k not in cls._pinned and k[1:] == key[1:] and
                           isinstance(k[0], tuple) == isinstance(key[0], tuple)), None)     # (same device, dtype and shape: scalar or [3])
            if victim is not None:
                t = cls._const.pop(victim)
                t.copy_(make().to(t.device))
                cls._const[key] = t
                return t
        t = cls._const[key] = make()
        return t

    @classmethod
    def const(cls, value, like):
        return cls._put((float(value), like.device, like.dtype), lambda: torch.full((), float(value), device=like.device, dtype=like.dtype), like)

    @classmethod
    def const3(cls, coeffs, like):
        """The coefficient vector [3] as a device tensor (cached per value triple): the upstream of the vector output."""
        return cls._put((coeffs, like.device, like.dtype), lambda: torch.tensor(coeffs, device=like.device, dtype=like.dtype), like)


def _is_number(x):
    return isinstance(x, (int, float)) and not isinstance(x, bool)


class _LazyLoss:
    """c0 * pos_intra + c1 * pos_inter + c2 * neg_inter of ONE forward call, not yet evaluated: what the reference's training step
    writes as ``(pos_inter_weight * pos_inter_loss + pos_intra_weight * pos_intra_loss + neg_inter_weight * neg_inter_loss) *
    correspondence_weight`` (train_segmentation.py:178-181) - five scalar kernels forward and four backward around a 50 us loss.

    THE CONTRACT, in one place.  This is NOT a tensor: it only answers (a) multiplication / division by a Python number, negation,
    addition / subtraction of another combination of the SAME forward call (all of which stay lazy), and (b) ``.backward()`` without an
    explicit gradient, which hands the three coefficients to the loss op's backward as its upstream gradients (no kernel at all).
    EVERYTHING else - any torch function, any other operand, any attribute or method of a tensor (``.item()``, ``float()``,
    ``.detach()``, ``+ other_loss``, ``torch.stack([...])``, logging) - first evaluates the sum with the plain torch ops on the plain
    tensors (``materialize()``: exactly the expression the caller wrote, with its normal autograd graph) and then behaves as that
    tensor does: same values, same gradients, only slower.  Tested op by op against the plain expression (tests/test_parity_gpu.py)."""
    __slots__ = ("_fam", "_c", "_value")

    def __init__(self, fam, coeffs):
        self._fam, self._c, self._value = fam, tuple(float(c) for c in coeffs), None

    # ---- the lazy algebra
    def _scaled(self, k):
        return _LazyLoss(self._fam, [c * k for c in self._c])

    def __mul__(self, k):
        return self._scaled(k) if _is_number(k) else self.materialize() * k

    __rmul__ = __mul__

    def __truediv__(self, k):
        return self._scaled(1.0 / k) if _is_number(k) and k != 0 else self.materialize() / k

    def __neg__(self):
        return self._scaled(-1.0)

    def __pos__(self):
        return self

    def _same(self, other):
        if isinstance(other, _LossScalar) and getattr(other, "_stego_fam", None) is self._fam:
            other = other._lazy()
        return other if isinstance(other, _LazyLoss) and other._fam is self._fam else None

    def __add__(self, other):
        o = self._same(other)
        if o is not None:
            return _LazyLoss(self._fam, [a + b for a, b in zip(self._c, o._c)])
        if _is_number(other) and other == 0:       # (``loss = 0; loss += ...``: train_segmentation.py:146,181)
            return self
        return self.materialize() + (other.materialize() if isinstance(other, _LazyLoss) else other)

    __radd__ = __add__

    def __sub__(self, other):
        o = self._same(other)
        if o is not None:
            return _LazyLoss(self._fam, [a - b for a, b in zip(self._c, o._c)])
        return self.materialize() - (other.materialize() if isinstance(other, _LazyLoss) else other)

    def __rsub__(self, other):
        return (-self).__add__(other)

    # ---- evaluation: the caller's expression on the plain tensors
    def materialize(self):
        if self._value is None:
            with torch._C.DisableTorchFunctionSubclass():
                v = None
                for c, o in zip(self._c, self._fam.outs):
                    if o is None or c == 0.0:
                        continue
                    t = o.as_subclass(torch.Tensor) * c
                    v = t if v is None else v + t
                if v is None:
                    ref = next(o for o in self._fam.outs if o is not None)
                    v = ref.as_subclass(torch.Tensor) * 0.0
            self._value = v
        return self._value

    def backward(self, gradient=None, retain_graph=None, create_graph=False, inputs=None):
        if gradient is None and not create_graph and self._value is None:
            fam = self._fam
            m = fam.means
            if m is not None and m.requires_grad:      # one root, one upstream: the coefficient vector
                g = _LossFamily.const3(tuple(0.0 if o is None else c for c, o in zip(self._c, fam.outs)), m)
                if g is not None:
                    return torch.autograd.backward(m, g, retain_graph=retain_graph, inputs=inputs)
            outs, grads = [], []
            for c, o in zip(self._c, fam.outs):
                if o is None or c == 0.0 or not o.requires_grad:
                    continue
                g = _LossFamily.const(c, o)
                if g is None:
                    outs = None
                    break
                outs.append(o)
                grads.append(g)
            if outs:
                return torch.autograd.backward(outs, grads, retain_graph=retain_graph, inputs=inputs)
        return self.materialize().backward(gradient, retain_graph=retain_graph, create_graph=create_graph, inputs=inputs)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        def plain(x):
            if isinstance(x, _LazyLoss):
                return x.materialize()
            if isinstance(x, (list, tuple)):
                return type(x)(plain(y) for y in x)
            return x
        return func(*plain(args), **{k: plain(v) for k, v in (kwargs or {}).items()})

    def __getattr__(self, name):               # anything a tensor has and this does not: the tensor's
        return getattr(self.materialize(), name)

    def __float__(self):
        return float(self.materialize())

    def __repr__(self):
        return "_LazyLoss(%s)" % (repr(self.materialize()),)

    def __format__(self, spec):
        return format(self.materialize(), spec)

    def __bool__(self):
        return bool(self.materialize())

    def __int__(self):
        return int(self.materialize())

    def __array__(self, dtype=None):
        return self.materialize().detach().cpu().numpy() if dtype is None else self.materialize().detach().cpu().numpy().astype(dtype)

    __hash__ = object.__hash__


def _delegate(name):
    def f(self, *args, **kwargs):
        return getattr(self.materialize(), name)(*[a.materialize() if isinstance(a, _LazyLoss) else a for a in args], **kwargs)
    f.__name__ = name
    return f


# the remaining operators of a tensor (Python looks special methods up on the type, not through __getattr__): the tensor's own
for _n in ("__abs__", "__pow__", "__rpow__", "__rtruediv__", "__floordiv__", "__rfloordiv__", "__mod__", "__rmod__", "__matmul__",
           "__rmatmul__", "__lt__", "__le__", "__gt__", "__ge__", "__eq__", "__ne__", "__getitem__", "__len__", "__iter__",
           "__invert__", "__and__", "__or__", "__xor__", "__complex__", "__index__", "__round__"):
    setattr(_LazyLoss, _n, _delegate(_n))
del _n


class _LossScalar(torch.Tensor):
    """One of the three scalar outputs (a REAL tensor: value, storage and autograd node are the op's own).  Multiplied by a Python number,
    or added to a sibling, it answers with a _LazyLoss (see its contract); ``.mean()`` of the 0-dim scalar is the scalar itself
    (train_segmentation.py:169-171 takes it); every other use is the plain tensor's."""

    def _lazy(self):
        return _LazyLoss(self._stego_fam, [1.0 if i == self._stego_idx else 0.0 for i in range(3)])

    # The operators of the training step's expression, answered without the __torch_function__ dispatch (a few us each on the host);
    # every case they do not take goes to the tensor's own operator, i.e. through __torch_function__ below - the same answers.
    def __mul__(self, k):
        if _is_number(k) and getattr(self, "_stego_fam", None) is not None:
            c = [0.0, 0.0, 0.0]
            c[self._stego_idx] = k
            return _LazyLoss(self._stego_fam, c)
        return torch.Tensor.__mul__(self, k)

    def __rmul__(self, k):
        if _is_number(k) and getattr(self, "_stego_fam", None) is not None:
            c = [0.0, 0.0, 0.0]
            c[self._stego_idx] = k
            return _LazyLoss(self._stego_fam, c)
        return torch.Tensor.__rmul__(self, k)

    def mean(self, *args, **kwargs):
        if not args and not kwargs and self.dim() == 0 and getattr(self, "_stego_fam", None) is not None:
            return self
        return torch.Tensor.mean(self, *args, **kwargs)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if not kwargs and args and isinstance(args[0], _LossScalar) and getattr(args[0], "_stego_fam", None) is not None:
            me = args[0]
            if func in (torch.Tensor.mean, torch.mean) and len(args) == 1 and me.dim() == 0:
                return me
            if len(args) == 2:
                other = args[1]
                if func in (torch.Tensor.mul, torch.Tensor.__mul__, torch.Tensor.__rmul__, torch.mul) and _is_number(other):
                    return me._lazy() * other
                if func in (torch.Tensor.div, torch.Tensor.__truediv__, torch.true_divide, torch.div) and _is_number(other) and other != 0:
                    return me._lazy() / other
                if func in (torch.Tensor.add, torch.Tensor.__add__, torch.Tensor.__radd__, torch.add) and \
                        (isinstance(other, _LazyLoss) or (isinstance(other, _LossScalar) and getattr(other, "_stego_fam", None) is me._stego_fam)):
                    return me._lazy() + other
                if func in (torch.Tensor.sub, torch.Tensor.__sub__, torch.sub) and \
                        (isinstance(other, _LazyLoss) or (isinstance(other, _LossScalar) and getattr(other, "_stego_fam", None) is me._stego_fam)):
                    return me._lazy() - other
            if len(args) == 1 and func in (torch.Tensor.neg, torch.Tensor.__neg__, torch.neg):
                return -me._lazy()
        elif not kwargs and len(args) == 2 and isinstance(args[1], _LossScalar) and getattr(args[1], "_stego_fam", None) is not None:
            me, other = args[1], args[0]       # number * scalar reaches here as Tensor.__rmul__(me, number): handled above; this is torch.mul(number, me)
            if func in (torch.mul,) and _is_number(other):
                return me._lazy() * other
        with torch._C.DisableTorchFunctionSubclass():
            out = func(*args, **kwargs)
        return out


def _lazy_scalars(o_intra, o_inter, o_neg, means=None):
    """Wrap the op's three scalar outputs (any may be None) as _LossScalar siblings of one family.  The family holds the PLAIN outputs
    and the wrappers hold the family - no reference cycle through a tensor's __dict__ (torch.cuda.graph() runs gc.collect() on entry;
    collecting such a cycle crashed the interpreter)."""
    plain = tuple(o if o is None or type(o) is torch.Tensor else o.as_subclass(torch.Tensor) for o in (o_intra, o_inter, o_neg))
    fam = _LossFamily(*plain, means=means)
    members = []
    for i, o in enumerate(plain):
        if o is None:
            members.append(None)
            continue
        w = o.as_subclass(_LossScalar)
        w._stego_idx = i
        w._stego_fam = fam
        members.append(w)
    return members


class _NegLossMap(torch.Tensor):
    """The negative loss tensor forward() returns (modules.py:390, ``torch.cat(neg_losses)``), carrying the mean the forward launch
    computed anyway.  The reference's training step only ever takes ``neg_inter_loss.mean()`` (train_segmentation.py:176): that call
    is answered with the kernel's scalar - no 9 MB reduction pass, and the backward gets one device scalar instead of a dense
    upstream (which would also push it onto the slower dense-upstream kernels).  Every other use behaves like the plain tensor
    (results are plain tensors; gradients flow to the same autograd node)."""

    def mean(self, *args, **kwargs):           # (the training step's call, answered without the __torch_function__ dispatch)
        if not args and not kwargs:
            m = getattr(self, "_stego_mean", None)
            if m is not None:
                return m
        return torch.Tensor.mean(self, *args, **kwargs)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in (torch.Tensor.mean, torch.mean) and len(args) == 1 and not kwargs:
            m = getattr(args[0], "_stego_mean", None)
            if m is not None:
                return m
        # identity conversions keep the shortcut alive: neg.float().mean(), neg.contiguous().mean(), neg.to(torch.float32).mean() and
        # neg.view(-1) / flatten() / reshape(-1) before .mean() are the same tensor to the reference's reader
        if func in _NEG_IDENTITY and args and getattr(args[0], "_stego_mean", None) is not None:
            with torch._C.DisableTorchFunctionSubclass():
                out = func(*args, **kwargs)
            if isinstance(out, torch.Tensor) and out.dtype == args[0].dtype and out.numel() == args[0].numel() and \
                    out.device == args[0].device:
                out = out.as_subclass(_NegLossMap)
                out._stego_mean = args[0]._stego_mean
            return out
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)


_NEG_IDENTITY = (torch.Tensor.float, torch.Tensor.contiguous, torch.Tensor.to, torch.Tensor.view, torch.Tensor.reshape,
                 torch.Tensor.flatten, torch.flatten, torch.reshape)


class _CorrLossFunction(torch.autograd.Function):
    """ContrastiveCorrelationLoss.forward as one op: stego_corr_fwd / stego_corr_bwd.  Seventh output: the mean over the negative
    loss tensor (loss_means[2]), for _NegLossMap."""

    @staticmethod
    def forward(ctx, feats, feats_pos, code, code_pos, coords1, coords2, perms, desc):
        need_grad = bool(code.requires_grad or code_pos.requires_grad)
        ctx.set_materialize_grads(False)       # an output nobody differentiated costs no zero-fill and no loads in the backward
        mn = _relayout_threshold(code)
        (loss_means, intra_cd, inter_cd, neg_loss, neg_cd, saved) = _backend.corr_fwd(
            desc, as_channels_last(feats.detach(), mn), as_channels_last(feats_pos.detach(), mn),
            as_channels_last(code.detach(), mn), as_channels_last(code_pos.detach(), mn), coords1, coords2, perms, need_grad)
        ctx.desc = desc
        if need_grad:
            ctx.n_saved = len(saved)
            ctx.save_for_backward(code, code_pos, coords1, coords2, perms, intra_cd, inter_cd, neg_cd, *saved)
        return loss_means[0], intra_cd, loss_means[1], inter_cd, neg_loss, neg_cd, loss_means[2]

    @staticmethod
    @once_differentiable
    def backward(ctx, g_intra, g_intra_cd, g_inter, g_inter_cd, g_neg_loss, g_neg_cd, g_neg_mean):
        code, code_pos, coords1, coords2, perms, intra_cd, inter_cd, neg_cd = ctx.saved_tensors[:8]
        saved = tuple(ctx.saved_tensors[8:])
        neg_is_mean = False
        if g_neg_mean is not None and ctx.desc.n_neg > 0:
            if g_neg_loss is None:
                g_neg_loss, neg_is_mean = g_neg_mean.reshape(1), True        # the training case: one device scalar
            else:                                                            # both the map and its mean were differentiated
                g_neg_loss = g_neg_loss + g_neg_mean / float(g_neg_loss.numel())
        d_code, d_code_pos = _backend.corr_bwd(ctx.desc, code.detach(), code_pos.detach(), coords1, coords2, perms,
                                               saved, intra_cd, inter_cd, neg_cd,
                                               g_intra, g_inter, g_neg_loss, g_intra_cd, g_inter_cd, g_neg_cd,
                                               neg_is_mean=neg_is_mean)
        return (None, None,
                d_code if ctx.needs_input_grad[2] else None,
                d_code_pos if ctx.needs_input_grad[3] else None,
                None, None, None, None)


class _CorrLossMeansFunction(torch.autograd.Function):
    """The same op returning the three loss MEANS as one tensor [3] = (pos_intra.mean(), pos_inter.mean(),
    cat(neg).mean()) - all three come out of the forward kernel - so that a training step combines them with one dot
    product and the backward gets three device scalars (no .mean() pass over the negative loss tensor, no expand)."""

    @staticmethod
    def forward(ctx, feats, feats_pos, code, code_pos, coords1, coords2, perms, desc):
        need_grad = bool(code.requires_grad or code_pos.requires_grad)
        ctx.set_materialize_grads(False)
        mn = _relayout_threshold(code)
        (loss_means, intra_cd, inter_cd, _neg_loss, neg_cd, saved) = _backend.corr_fwd(
            desc, as_channels_last(feats.detach(), mn), as_channels_last(feats_pos.detach(), mn),
            as_channels_last(code.detach(), mn), as_channels_last(code_pos.detach(), mn), coords1, coords2, perms, need_grad)
        ctx.desc = desc
        if need_grad:
            ctx.save_for_backward(code, code_pos, coords1, coords2, perms, intra_cd, inter_cd, neg_cd, *saved)
        return loss_means, intra_cd, inter_cd, neg_cd

    @staticmethod
    @once_differentiable
    def backward(ctx, g_means, g_intra_cd, g_inter_cd, g_neg_cd):
        code, code_pos, coords1, coords2, perms, intra_cd, inter_cd, neg_cd = ctx.saved_tensors[:8]
        saved = tuple(ctx.saved_tensors[8:])
        if g_means is not None:
            g_means = g_means.contiguous()
        gi = None if g_means is None else g_means[0:1]
        ge = None if g_means is None else g_means[1:2]
        gn = None if (g_means is None or ctx.desc.n_neg == 0) else g_means[2:3]
        d_code, d_code_pos = _backend.corr_bwd(ctx.desc, code.detach(), code_pos.detach(), coords1, coords2, perms,
                                               saved, intra_cd, inter_cd, neg_cd, gi, ge, gn, g_intra_cd, g_inter_cd, g_neg_cd,
                                               neg_is_mean=True)
        return (None, None,
                d_code if ctx.needs_input_grad[2] else None,
                d_code_pos if ctx.needs_input_grad[3] else None,
                None, None, None, None)


class _HelperFunction(torch.autograd.Function):
    """ContrastiveCorrelationLoss.helper on pre-sampled tensors: stego_corr_helper_fwd / _bwd."""

    @staticmethod
    def forward(ctx, f1, f2, c1, c2, desc):
        need_grad = bool(c1.requires_grad or c2.requires_grad)
        ctx.set_materialize_grads(False)
        loss, cd, saved = _backend.helper_fwd(desc, f1.detach(), f2.detach(), c1.detach(), c2.detach(), need_grad)
        ctx.desc = desc
        if need_grad:
            ctx.save_for_backward(c1, c2, cd, *saved)
        return loss, cd

    @staticmethod
    @once_differentiable
    def backward(ctx, g_loss, g_cd):
        c1, c2, cd = ctx.saved_tensors[:3]
        d1, d2 = _backend.helper_bwd(ctx.desc, c1.detach(), c2.detach(), tuple(ctx.saved_tensors[3:]), cd, g_loss, g_cd)
        return (None, None, d1 if ctx.needs_input_grad[2] else None, d2 if ctx.needs_input_grad[3] else None, None)


# --------------------------------------------------------------------------- the loss
_PAIR_SET_BOUND = {}


def _pair_set_bound(device=None):
    """Largest batch of the single-launch forward: the tiles of one pair-set run at the same time, one per compute unit
    (fused_supported, csrc/corr_fused.hip: device_cu_count() & ~7) - 256 on a whole MI355X, fewer on a partitioned one."""
    if not torch.cuda.is_available():
        return 256
    idx = torch.cuda.current_device() if device is None or getattr(device, "index", None) is None else device.index
    n = _PAIR_SET_BOUND.get(idx)
    if n is None:       # per device: a partitioned or different device in the same process has its own compute-unit count
        n = _PAIR_SET_BOUND[idx] = max(8, torch.cuda.get_device_properties(idx).multi_processor_count & ~7)
    return n


class ContrastiveCorrelationLoss(nn.Module):
    """Drop-in for the reference class (modules.py:314-398); same cfg keys:
    feature_samples, neg_samples, pointwise, zero_clamp, stabalize, use_salience,
    pos_intra_shift, pos_inter_shift, neg_inter_shift.  Optional extra key
    ``corr_precision`` ('f16x3' default | 'f32')."""

    def __init__(self, cfg, ):
        super(ContrastiveCorrelationLoss, self).__init__()
        self.cfg = cfg

    def standard_scale(self, t):
        t1 = t - t.mean()
        t2 = t1 / t1.std()
        return t2

    def helper(self, f1, f2, c1, c2, shift):
        """(loss, cd) for already-sampled f1,f2 [N,C,S1,S2] / c1,c2 [N,K,S1,S2] (modules.py:325-347)."""
        N, C, S1, S2 = f1.shape
        K = c1.shape[1]
        if f1.is_cuda and (S1 * S2 > 128 or K > 72):               # limits of stego_corr_helper_fwd (include/stego_corr.h)
            return self.generic_helper(f1, f2, c1, c2, shift)
        desc = capi.make_desc(N, C, K, S1, S2, S2, 0, self.cfg, (shift, shift, shift), _precision_of(self.cfg))
        return _HelperFunction.apply(f1, f2, c1, c2, desc)

    def draw_coords(self, orig_feats, orig_salience, orig_salience_pos):
        """The coords1/coords2 draws of modules.py:355-367, same order, same generator."""
        cfg = self.cfg
        coord_shape = [orig_feats.shape[0], cfg.feature_samples, cfg.feature_samples, 2]
        dev = orig_feats.device
        if cfg.use_salience:
            coords1_nonzero = sample_nonzero_locations(orig_salience, coord_shape)
            coords2_nonzero = sample_nonzero_locations(orig_salience_pos, coord_shape)
            coords1_reg = torch.rand(coord_shape, device=dev) * 2 - 1
            coords2_reg = torch.rand(coord_shape, device=dev) * 2 - 1
            mask = (torch.rand(coord_shape[:-1], device=dev) > .1).unsqueeze(-1).to(torch.float32)
            coords1 = coords1_nonzero * mask + coords1_reg * (1 - mask)
            coords2 = coords2_nonzero * mask + coords2_reg * (1 - mask)
        else:
            coords1 = torch.rand(coord_shape, device=dev) * 2 - 1
            coords2 = torch.rand(coord_shape, device=dev) * 2 - 1
        return coords1, coords2

    @staticmethod
    def fused_kernels_cover(B, C, K, H, W, S, device=None):
        """Does the hand-written loss path (stego_corr_fwd / _bwd, include/stego_corr.h "Limits of this build") take this shape?
        S * S <= 128 sample points per image: K <= 72 on any layout, 72 < K <= 128 on the single-launch forward (any parity, ViT widths, B and the
        map within its bounds); 128 < S * S <= 256 (cfg.feature_samples 12 .. 16): any K <= 128, any layout.  Everything else is computed by
        generic_forward()."""
        if K > 128 or H > 32767 or W > 32767:
            return False
        if S * S > 128:
            # 129 .. 256 points per image (cfg.feature_samples 12 .. 16): the multi-launch kernels of csrc/corr_wide.hip behind the same entry
            # points (round 5)
            return S <= 16 and (2 + 253) * B <= 65535
        if K > 72:
            return C in (192, 384, 768) and B <= _pair_set_bound(device) and H <= 256 and W <= 256
        return True

    def generic_helper(self, f1, f2, c1, c2, shift):
        """modules.py:325-347 with its einsums on the native dense-correspondence kernel (tensor_correlation -> stego_dense_corr,
        forward and both adjoints) and the elementwise work in torch: the path of shapes the fused kernels do not cover."""
        cfg = self.cfg
        with torch.no_grad():
            fd = tensor_correlation(norm(f1), norm(f2))
            if cfg.pointwise:
                old_mean = fd.mean()
                fd -= fd.mean([3, 4], keepdim=True)
                fd = fd - fd.mean() + old_mean
        cd = tensor_correlation(norm(c1), norm(c2))
        min_val = 0.0 if cfg.zero_clamp else -9999.0
        if cfg.stabalize:
            loss = -cd.clamp(min_val, .8) * (fd - shift)
        else:
            loss = -cd.clamp(min_val) * (fd - shift)
        return loss, cd

    def generic_forward(self, orig_feats, orig_feats_pos, orig_code, orig_code_pos, coords1, coords2, perms):
        """modules.py:369-398 for any shape (any cfg.feature_samples, any cfg.dim) the kernels behind stego_corr_fwd do not take (fused_kernels_cover:
        cfg.dim > 128, feature_samples > 16).  Same six return values.  On a HIP device: the samplers that write the
        dense-correspondence kernel's operands (stego_sample_panels: the fp32 rows of the sampled features never exist), both correlation
        tensors of ALL 2 + neg_samples pair-sets in one launch each, helper()'s elementwise part in three launches, the adjoints as batched
        GEMMs, norm + sampling backward in one scatter per source - composed here, gradient through autograd (host-bound: ~0.6 ms per step at
        B = 32, S = 16; round 4: 3.8 ms, the reference's loop of helper() calls 8.5 ms; tools/exp/generic_time.py).  Elsewhere (CPU tensors in
        tests, non-square coordinate grids): the reference's expressions in torch, all pair-sets as one batch."""
        cfg = self.cfg
        B = orig_feats.shape[0]
        n_neg = int(perms.shape[0]) if perms is not None else 0
        n_sets = 2 + n_neg
        min_val = 0.0 if cfg.zero_clamp else -9999.0
        idx = perms.reshape(-1) if n_neg else None                       # [n_neg * B]
        if orig_feats.is_cuda and orig_feats.dtype == torch.float32 and orig_code.dtype == torch.float32 and coords1.shape[1] == coords1.shape[2] \
                and coords1.shape == coords2.shape:
            # native (round 5).  The negatives read the maps of image perm_n[b] at pair (n, b)'s coords2[b] through an index, not through the
            # permuted copies of modules.py:384-385 (5 x (38.5 + 7) MB at BASELINE config 2, the largest single cost of the reference's step);
            # the sampler normalises (norm(), :275-276) and writes the dense kernel's split-fp16 OPERANDS - the fp32 rows of the sampled features
            # (88 MB at S = 16) never exist, the anchors are one operand set against all pair-sets' second operands (no repeat / cat copies);
            # helper()'s elementwise part for all pair-sets is three launches (stego_rowsum / stego_loss_pointwise_fwd / _bwd)
            S = int(coords1.shape[1])
            P = S * S
            C = orig_feats.shape[1]
            with torch.no_grad():
                fset = capi.PanelSet(n_sets * B, C, P, orig_feats.device)
                capi.sample_panels(fset, 0, orig_feats, coords1)
                capi.sample_panels(fset, B, orig_feats_pos, coords2)
                if n_neg:
                    capi.sample_panels(fset, 2 * B, orig_feats, coords2, idx)
                fd, fd_rowsum = capi.dense_corr_panels(fset, B, fset, n_sets * B, want_rowsum=bool(cfg.pointwise))[:2] if cfg.pointwise \
                    else (capi.dense_corr_panels(fset, B, fset, n_sets * B), None)
                del fset
            cd = _CodeCorrFunction.apply(orig_code, orig_code_pos, coords1, coords2, idx).view(n_sets, B, S, S, S, S)
            neg, sums = _PointwiseLossFunction.apply(fd.view(n_sets, B, P, P), cd.view(n_sets, B, P, P),
                                                     (cfg.pos_intra_shift, cfg.pos_inter_shift, cfg.neg_inter_shift), min_val,
                                                     0.8 if cfg.stabalize else 3.0e38, bool(cfg.pointwise), fd_rowsum)
            cnt = float(B * P * P)
            if n_neg:
                neg_inter_loss, neg_inter_cd = neg.view(n_neg * B, S, S, S, S), cd[2:].flatten(0, 1)
            else:
                neg_inter_loss = neg_inter_cd = cd.new_zeros(0, S, S, S, S)
            return sums[0] / cnt, cd[0], sums[1] / cnt, cd[1], neg_inter_loss, neg_inter_cd
        feats, code = sample_indexed(orig_feats, coords1), sample_indexed(orig_code, coords1)
        f2, c2 = [feats, sample_indexed(orig_feats_pos, coords2)], [code, sample_indexed(orig_code_pos, coords2)]
        if n_neg:
            f2.append(sample_indexed(orig_feats, coords2, idx))
            c2.append(sample_indexed(orig_code, coords2, idx))
        f2, c2 = torch.cat(f2), torch.cat(c2)
        f1, c1 = feats.repeat(n_sets, 1, 1, 1), code.repeat(n_sets, 1, 1, 1)
        S1, S2 = feats.shape[2:]
        per_set = lambda t: t.view(n_sets, B, S1, S2, S1, S2)            # noqa: E731
        with torch.no_grad():
            fd = per_set(tensor_correlation(norm(f1), norm(f2)))
            if cfg.pointwise:                                            # helper(), modules.py:331-333, per pair-set
                # (a set's mean as the mean of its row means - equal counts: the same number, and a reduction with B S^2 outputs per
                # set instead of ONE, which torch runs on a handful of workgroups: 0.45 ms each at S = 16)
                row = fd.mean([4, 5], keepdim=True)
                old_mean = row.mean(dim=(1, 2, 3), keepdim=True)
                fd -= row
                fd = fd - fd.mean([4, 5], keepdim=True).mean(dim=(1, 2, 3), keepdim=True) + old_mean
            # fd - shift with the pair-set's own shift (Python numbers: no host-to-device copy, the step stays capturable in a graph)
            fds = torch.empty_like(fd)
            torch.sub(fd[0], cfg.pos_intra_shift, out=fds[0])
            torch.sub(fd[1], cfg.pos_inter_shift, out=fds[1])
            if n_neg:
                torch.sub(fd[2:], cfg.neg_inter_shift, out=fds[2:])
        cd = per_set(tensor_correlation(norm(c1), norm(c2)))
        loss = -(cd.clamp(min_val, .8) if cfg.stabalize else cd.clamp(min_val)) * fds
        if n_neg:
            neg_inter_loss, neg_inter_cd = loss[2:].flatten(0, 1), cd[2:].flatten(0, 1)
        else:
            neg_inter_loss = neg_inter_cd = cd.new_zeros(0, S1, S2, S1, S2)
        return loss[0].mean(), cd[0], loss[1].mean(), cd[1], neg_inter_loss, neg_inter_cd

    def forward_explicit(self, orig_feats, orig_feats_pos, orig_code, orig_code_pos, coords1, coords2, perms):
        """forward() with the RNG draws supplied by the caller (perms: int64 [neg_samples, B])."""
        cfg = self.cfg
        B, C, H, W = orig_feats.shape
        K = orig_code.shape[1]
        S = cfg.feature_samples
        if orig_feats.is_cuda and not self.fused_kernels_cover(B, C, K, H, W, S, orig_feats.device):
            return self.generic_forward(orig_feats, orig_feats_pos, orig_code, orig_code_pos, coords1, coords2, perms)
        n_neg = int(perms.shape[0]) if perms is not None else 0
        if perms is None:
            perms = torch.zeros(0, B, dtype=torch.long, device=orig_feats.device)
        desc = capi.make_desc(B, C, K, H, W, S, n_neg, cfg,
                              (cfg.pos_intra_shift, cfg.pos_inter_shift, cfg.neg_inter_shift), _precision_of(cfg))
        ext = _native_autograd(cfg)
        if ext is not None:            # the same op as a C++ autograd function (csrc/torch_glue_ext.cpp): a third of the host time
            mn = _relayout_threshold(orig_code)
            out = ext.corr_loss(as_channels_last(orig_feats, mn), as_channels_last(orig_feats_pos, mn), as_channels_last(orig_code, mn),
                                as_channels_last(orig_code_pos, mn), coords1, coords2, perms, bytes(desc), 0)
        else:
            out = _CorrLossFunction.apply(orig_feats, orig_feats_pos, orig_code, orig_code_pos, coords1, coords2, perms, desc)
        neg_loss = out[4]
        # OPT-IN (round 6: the default is the reference's contract - three plain 0-dim tensors, so that ``w * loss + ...`` in the
        # reference's own training_step (train_segmentation.py:179-181,227) is a torch.Tensor): cfg.lazy_loss_sums = True wraps them
        lazy = getattr(cfg, "lazy_loss_sums", False)
        o_intra, o_inter, o_neg = _lazy_scalars(out[0], out[2], out[6] if n_neg > 0 else None, out[7] if len(out) > 7 else None) \
            if lazy else (out[0], out[2], out[6])
        if n_neg > 0:
            neg_loss = neg_loss.as_subclass(_NegLossMap)          # (an alias: same storage, same autograd node)
            neg_loss._stego_mean = o_neg
        return o_intra, out[1], o_inter, out[3], neg_loss, out[5]

    def forward(self,
                orig_feats: torch.Tensor, orig_feats_pos: torch.Tensor,
                orig_salience: torch.Tensor, orig_salience_pos: torch.Tensor,
                orig_code: torch.Tensor, orig_code_pos: torch.Tensor,
                ):
        coords1, coords2, perms = self.draw(orig_feats, orig_salience, orig_salience_pos)
        return self.forward_explicit(orig_feats, orig_feats_pos, orig_code, orig_code_pos, coords1, coords2, perms)

    def means(self, orig_feats, orig_feats_pos, orig_salience, orig_salience_pos, orig_code, orig_code_pos):
        """forward() for a training step that only needs the three loss means (train_segmentation.py:176-181 takes
        .mean() of each): returns (means [3] = (pos_intra, pos_inter, neg_inter), pos_intra_cd, pos_inter_cd, neg_inter_cd).
        Same draws, same kernel; the negative loss tensor is not handed back, its mean comes out of the forward launch."""
        coords1, coords2, perms = self.draw(orig_feats, orig_salience, orig_salience_pos)
        cfg = self.cfg
        B, C, H, W = orig_feats.shape
        if orig_feats.is_cuda and not self.fused_kernels_cover(B, C, orig_code.shape[1], H, W, cfg.feature_samples, orig_feats.device):
            o = self.generic_forward(orig_feats, orig_feats_pos, orig_code, orig_code_pos, coords1, coords2, perms)
            neg_mean = o[4].mean() if o[4].numel() else o[0].new_zeros(())
            return torch.stack([o[0], o[2], neg_mean]), o[1], o[3], o[5]
        n_neg = int(perms.shape[0]) if perms is not None else 0
        if perms is None:
            perms = torch.zeros(0, B, dtype=torch.long, device=orig_feats.device)
        desc = capi.make_desc(B, C, orig_code.shape[1], H, W, cfg.feature_samples, n_neg, cfg,
                              (cfg.pos_intra_shift, cfg.pos_inter_shift, cfg.neg_inter_shift), _precision_of(cfg))
        ext = _native_autograd(cfg)
        if ext is not None:
            mn = _relayout_threshold(orig_code)
            return tuple(ext.corr_loss(as_channels_last(orig_feats, mn), as_channels_last(orig_feats_pos, mn),
                                       as_channels_last(orig_code, mn), as_channels_last(orig_code_pos, mn), coords1, coords2, perms,
                                       bytes(desc), 1))
        return _CorrLossMeansFunction.apply(orig_feats, orig_feats_pos, orig_code, orig_code_pos, coords1, coords2, perms, desc)

    def total(self, orig_feats, orig_feats_pos, orig_salience, orig_salience_pos, orig_code, orig_code_pos, weights):
        """weights = (pos_intra_weight, pos_inter_weight, neg_inter_weight) -> (the weighted sum of the three loss means as one
        dot product, means [3] detached for logging, pos_intra_cd, pos_inter_cd, neg_inter_cd)."""
        m, icd, ecd, ncd = self.means(orig_feats, orig_feats_pos, orig_salience, orig_salience_pos, orig_code, orig_code_pos)
        key = (tuple(float(w) for w in weights), m.device)
        w = self._weights.get(key) if hasattr(self, "_weights") else None
        if w is None:
            if not hasattr(self, "_weights"):
                self._weights = {}
            w = torch.tensor(key[0], dtype=torch.float32, device=m.device)
            self._weights[key] = w
        return torch.dot(m, w), m.detach(), icd, ecd, ncd

    def draw(self, orig_feats, orig_salience, orig_salience_pos):
        """The RNG draws of forward() (modules.py:355-367, 382-385) -> (coords1, coords2, perms or None)."""
        B = orig_feats.shape[0]
        dev = orig_feats.device
        cfg = self.cfg
        if dev.type == "cuda" and not cfg.use_salience and getattr(cfg, "fast_draws", False) and hasattr(_backend, "fast_draws"):
            # OPT-IN (cfg.fast_draws, default off): the same distributions from ONE kernel keyed by 64 bits of the torch
            # generator - not the reference's random stream, ~30 launches fewer per step (torch.randperm alone is 5 kernels)
            seed = torch.randint(-2 ** 63, 2 ** 63 - 1, (1,), dtype=torch.int64, device=dev)
            shape = [B, cfg.feature_samples, cfg.feature_samples, 2]
            coords1, coords2, perms = _backend.fast_draws(seed, shape, cfg.neg_samples, B)
            if cfg.neg_samples == 0:
                perms = None
        elif dev.type == "cuda" and not cfg.use_salience and hasattr(_backend, "ref_draws") and B <= 2048 and \
                getattr(cfg, "one_launch_draws", True) and \
                (variant := _usable_ref_draw_variant([B, cfg.feature_samples, cfg.feature_samples, 2], cfg.neg_samples, B, dev)) >= 0:
            # DEFAULT on a HIP device: the reference's draws - the same numbers torch.rand x 2 and torch.randperm x neg_samples give
            # from this generator state, the generator advanced by the same amount - from ONE launch (stego_ref_draws) instead of
            # ~30 tiny kernels.  Checked against the real torch calls once per process and size (ref_draw_variant).  Under stream
            # capture the generator's state lives on the device: the kernel reads it there (stego_ref_draws_indirect, through the
            # in-tree torch extension); a size that was never drawn eagerly before the capture keeps the torch calls below.
            shape = [B, cfg.feature_samples, cfg.feature_samples, 2]
            gen = _device_generator(dev)
            coords1, coords2, perms = _backend.ref_draws(gen, shape, cfg.neg_samples, B, variant, dev)
            if cfg.neg_samples == 0:
                perms = None
        elif dev.type == "cuda" and not cfg.use_salience and hasattr(_backend, "finish_draws"):
            # The reference's draws in the reference's order on the device generator (:366, :367, :383): torch.rand x2, then
            # one torch.randperm per negative; "* 2 - 1" and the super_perm fix-up run in ONE kernel (stego_finish_draws)
            # instead of nine.  (Forking the ~30 tiny draw kernels onto side streams while a HIP graph is captured was
            # measured SLOWER: 352 vs 268 us per replay - cross-branch edges cost more than the kernels they overlap.)
            shape = [B, cfg.feature_samples, cfg.feature_samples, 2]
            n = cfg.neg_samples
            out = _torch_draws(shape, n, B, dev)
            coords1, coords2, perms = _backend.finish_draws(out[0], out[1], out[2:], B)
            if n == 0:
                perms = None
        else:
            coords1, coords2 = self.draw_coords(orig_feats, orig_salience, orig_salience_pos)
            # :382-383 - one randperm per negative from the device generator (the reference's draws), one batched fix-up
            raw = [torch.randperm(B, device=dev, dtype=torch.long) for _ in range(cfg.neg_samples)]
            perms = _unfix(torch.stack(raw)) if raw else None
        return coords1, coords2, perms
