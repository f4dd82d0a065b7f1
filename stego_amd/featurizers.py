"""Featurizers, probes and the optional CRF loss: the rest of the reference's ``src/modules.py``
surface (``from modules import *`` is how its scripts get them).  Stock PyTorch-ROCm: these
produce/consume the hot path's tensors but are not on it.

Same constructor signatures, attribute names and state-dict keys as the reference so that its
checkpoints (``net.cluster1.0.*``, ``net.cluster2.{0,2}.*``, ``cluster_probe.clusters`` ...)
load unchanged:
  LambdaLayer :8-14, DinoFeaturizer :17-118, ResizeAndClassify :121-131, ClusterLookup :134-161,
  FeaturePyramidNet :164-252, DoubleConv :255-272, Decoder :401-413, NetWithActivations :416-434,
  ContrastiveCRFLoss :437-469.
"""
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import capi, dino_vit, vit_native


class LambdaLayer(nn.Module):
    def __init__(self, lambd):
        super().__init__()
        self.lambd = lambd

    def forward(self, x):
        return self.lambd(x)


class TokenCache:
    """Backbone tokens of a fixed-crop dataset kept in HBM ("optionally cache features for the fixed
    five-crop dataset").  The backbone is frozen and the reference's `img` / `img_pos` of dataset index i are the same
    pixels every epoch (data.py:520-565: deterministic resize + crop; only `img_aug` is random), so their tokens are a
    pure function of i: computed once, stored as fp16 (ViT-S/8 at 224^2: 603 KB per image - a 118 k-image epoch is 71 GB,
    a quarter of one MI355X's 288 GB), served from memory afterwards.  Under data parallelism every rank holds the WHOLE
    table: the reference's DistributedSampler reshuffles globally every epoch, so every rank meets every index over time (a
    rank-stable partition would let the table shard with the data, at the price of a different sampling order).  A
    training step then skips the backbone: 8.4 -> ~1.3 ms at B = 32 pairs (tools/bench_step.py)."""

    def __init__(self, n_items, ntok, dim, device, dtype=torch.float16):
        self.tokens = torch.empty(n_items, ntok, dim, dtype=dtype, device=device)
        self.filled = torch.zeros(n_items, dtype=torch.bool, device=device)
        self.complete = False
        self.misses = 0
        self.last = None              # (tokens fp32, magnitude word) of the latest fetch on a HIP device

    def clear(self):
        """Forget everything (the backbone's weights changed)."""
        self.filled.zero_()
        self.complete = False

    def fetch(self, index, img, compute):
        """tokens fp32 [B, ntok, D] for dataset indices `index` (long [B]); `compute(img_subset)` runs the backbone for
        the rows not cached yet.  The miss test is a host sync, paid only until the table is full."""
        index = index.to(self.tokens.device)
        if not self.complete:
            missing = ~self.filled[index]
            if bool(missing.any()):
                rows = index[missing]
                self.tokens[rows] = compute(img[missing]).to(self.tokens.dtype)
                self.filled[rows] = True
                self.misses += int(missing.sum())
            self.complete = bool(self.filled.all())
        if self.tokens.is_cuda and self.tokens.dtype == torch.float16:
            # one pass: rows -> fp32 + the largest magnitude of the patch tokens (what the native head prescales by: it would read all
            # tokens again for it)
            out, amax = capi.tokens_from_cache(self.tokens, index, skip_rows=1)
            self.last = (out, amax)
            return out
        return self.tokens[index].float()


class _TokenLinear(torch.autograd.Function):
    """F.linear over a [B, h, w, C] token tensor whose weight gradient is reduced per image first:
    dW = sum_b dY_b^T X_b as one batched GEMM (K = h*w per image) + a sum over the batch, instead of one GEMM with
    K = B*h*w = 50 176 and a 70..384-wide output, for which the BLAS library picks a 45 TFLOP/s kernel (1.2 ms per step
    for the three weight gradients of the head, tools/bench_step.py).  Same math, fp32."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return F.linear(x, w, b)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        B = x.shape[0]
        g3 = g.reshape(B, -1, g.shape[-1])
        gx = g.matmul(w) if ctx.needs_input_grad[0] else None
        gw = torch.bmm(g3.transpose(1, 2), x.reshape(B, -1, x.shape[-1])).sum(0) if ctx.needs_input_grad[1] else None
        gb = g3.sum((0, 1)) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return gx, gw, gb


# (device index, B * C, n, keep probability) -> variant of stego_ref_dropout_masks that reproduces the installed torch, or -1
_MASK_VARIANTS = {}


def _mask_variant(net, x, n, q):
    """Which variant of the one-launch Dropout2d masks is bit-identical to THIS torch build's bernoulli_ / div_ for these sizes: found by
    running both from the same generator state once (the generator is left where it was); -1 = none.  Under stream capture (where the
    check cannot run) only a cached answer is used."""
    from .modules import _device_generator
    dev = x.device
    key = (dev.index, x.shape[0] * x.shape[1], int(n), float(q))
    v = _MASK_VARIANTS.get(key)
    if v is not None:
        return v
    if torch.cuda.is_current_stream_capturing():
        return -1
    v = -1
    try:
        gen = _device_generator(dev)
        state = gen.get_state()
        try:
            want = torch.stack([net._feature_noise(x).view(-1) for _ in range(n)])
            off = gen.get_offset()
            for cand in range(4):
                gen.set_state(state)
                got = capi.ref_dropout_masks(gen, n, key[1], q, cand, dev)
                if gen.get_offset() == off and torch.equal(got, want):
                    v = cand
                    break
        finally:
            gen.set_state(state)
    except (RuntimeError, AttributeError):
        v = -1
    _MASK_VARIANTS[key] = v
    from .modules import _report_variant
    _report_variant("the Dropout2d masks of the segmentation head (%d masks of %d channels)" % (n, key[1]), "stego_ref_dropout_masks", v,
                    "%d bernoulli_ / div_ launches per step instead of one" % (2 * int(n)))
    return v


class _NativeHeadFunction(torch.autograd.Function):
    """The segmentation head on the hand-written kernels of include/stego_head.h (modules.py:108-116): the three 1x1 convolutions as
    split-fp16 GEMMs over the token matrix with the Dropout2d masks applied while the tokens are staged, the dropped-out feature
    map written on the way; backward = the six parameter gradients (the backbone is frozen: the tokens get none)."""

    @staticmethod
    def forward(ctx, tokens, m1, m2, m3, w1, b1, w21, b21, w22, b22, want_feats, tokens_amax=None):
        from . import capi
        need_grad = any(t is not None and t.requires_grad for t in (w1, b1, w21, b21, w22, b22))
        masks = (m1, m2, m3) if m1 is not None else None
        det = lambda t: None if t is None else t.detach()
        code, feats, saved_h = capi.head_fwd(tokens.detach(), masks, det(w1), det(b1), det(w21), det(b21), det(w22), det(b22),
                                             need_grad, want_feats, tokens_amax)
        ctx.set_materialize_grads(False)
        ctx.K = w1.shape[0]
        ctx.nonlinear = w21 is not None
        if need_grad:
            ctx.save_for_backward(tokens, m1, m2, saved_h, w22)
        if feats is not None:
            ctx.mark_non_differentiable(feats)
        return code, feats

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_code, _g_feats):
        from . import capi
        if g_code is None:
            return (None,) * 12
        tokens, m1, m2, saved_h, w22 = ctx.saved_tensors
        masks = (m1, m2, None) if m1 is not None else None
        dw1, db1, dw21, db21, dw22, db22 = capi.head_bwd(tokens, masks, saved_h if ctx.nonlinear else None,
                                                          w22.detach() if w22 is not None else None, g_code, ctx.K)
        return None, None, None, None, dw1, db1, dw21, db21, dw22, db22, None, None


class DinoFeaturizer(nn.Module):
    """Frozen DINO ViT + trainable 1x1-conv segmentation head.  forward(img) -> (feats, code) with
    feats a channels-last strided VIEW [B,C,h,w] of the tokens - the layout the loss kernels want."""

    _N_FEATS = {"vit_tiny": 192, "vit_small": 384, "vit_base": 768}

    def __init__(self, dim, cfg):
        super().__init__()
        self.cfg = cfg
        self.dim = dim
        self.patch_size = cfg.dino_patch_size
        self.feat_type = cfg.dino_feat_type
        arch = cfg.model_type
        if arch not in self._N_FEATS or self.patch_size not in (8, 16):
            raise ValueError("Unknown arch and patch size")
        self.model = dino_vit.ARCHS[arch](patch_size=self.patch_size, num_classes=0)
        for p in self.model.parameters():
            p.requires_grad = False
        self.model.eval()
        if torch.cuda.is_available():
            self.model.cuda()
        self.dropout = nn.Dropout2d(p=.1)
        self.token_cache = None       # TokenCache (enable_token_cache) or None
        self._native = None           # vit_native.NativeViT, built on first use on a HIP device
        self.backbone_path = None     # "native" | "torch": which path the last forward took
        # new weights (also when loaded through a parent module's load_state_dict) -> re-pack the backbone on next use
        def _new_weights(module, _incompatible):
            if module._native:
                module._native.invalidate()
            if module.token_cache is not None:
                module.token_cache.clear()          # tokens of the OLD backbone
        self.register_load_state_dict_post_hook(_new_weights)

        weights = getattr(cfg, "pretrained_weights", None)
        if weights is not None:
            sd = torch.load(weights, map_location="cpu")
            sd = sd.get("teacher", sd)
            sd = {k.replace("module.", "").replace("backbone.", ""): v for k, v in sd.items()}
            msg = self.model.load_state_dict(sd, strict=False)
            print("Pretrained weights found at {} and loaded with msg: {}".format(weights, msg))
        else:
            # the reference downloads the public DINO checkpoint here (modules.py:59-62); this build is
            # offline, so the backbone stays randomly initialised (benchmarks / tests use synthetic data)
            warnings.warn("DinoFeaturizer: no cfg.pretrained_weights and no network: DINO backbone is randomly initialised")

        self.n_feats = self._N_FEATS[arch]
        self.cluster1 = self.make_clusterer(self.n_feats)
        self.proj_type = cfg.projection_type
        if self.proj_type == "nonlinear":
            self.cluster2 = self.make_nonlinear_clusterer(self.n_feats)

    def make_clusterer(self, in_channels):
        return nn.Sequential(nn.Conv2d(in_channels, self.dim, (1, 1)))

    def make_nonlinear_clusterer(self, in_channels):
        return nn.Sequential(nn.Conv2d(in_channels, in_channels, (1, 1)), nn.ReLU(),
                             nn.Conv2d(in_channels, self.dim, (1, 1)))

    def _tokens(self, img, n):
        """feat[0] (and qkv[0]) of get_intermediate_feat (modules.py:88-89).  On a HIP device the frozen backbone runs on the
        hand-written kernels of include/stego_vit.h - DEFAULT (cfg.native_backbone, True unless set False) in precision
        cfg.backbone_precision = "f16x3": split-fp16 operands with three MFMAs per product, fp32 accumulation, statistics and
        residual stream; its error against the fp64 model is at or below the fp32 torch model's own (tests/test_vit_native.py:
        4e-7 .. 4e-5 depending on the network), at half the fp32 torch module's time.  "f16" (opt-in) is the plain fp16-operand
        path: another 2x faster, 5e-4 .. 1e-3 relative L2 with single elements up to 4e-2 off - outside the fp32 class.  The torch
        module is kept for the variants the kernels do not build: n > 1, feat_type 'KK' (needs the block's qkv), CPU tensors,
        geometries outside include/stego_vit.h."""
        native_ok = (img.is_cuda and n == 1 and self.feat_type == "feat" and img.dtype == torch.float32
                     and getattr(self.cfg, "native_backbone", True) and vit_native.supported(self.model))
        if native_ok:
            if self._native is None:
                self._native = vit_native.NativeViT(self.model, precision=getattr(self.cfg, "backbone_precision", "f16x3"))
            if self._native.shape_supported(*[int(img.shape[i]) for i in (0, 2, 3)]):      # (else: the torch module below)
                self.backbone_path = "native"
                return self._native.forward_tokens(img), None
        self.backbone_path = "torch"
        feat, _, qkv = self.model.get_intermediate_feat(img, n=n)
        return feat[0], qkv[0]


    def enable_token_cache(self, n_items, img_hw, device, dtype=torch.float16):
        """Keep the backbone tokens of dataset items 0..n_items-1 in device memory (see TokenCache)."""
        ntok = 1 + (img_hw[0] // self.patch_size) * (img_hw[1] // self.patch_size)
        self.token_cache = TokenCache(n_items, ntok, self.n_feats, device, dtype)
        return self.token_cache

    def forward(self, img, n=1, return_class_feat=False, cache_index=None):
        if self.model.training:              # (modules.py:80 calls eval() on every forward: ~130 modules walked in Python, 0.45 ms per call)
            self.model.eval()
        with torch.no_grad():
            assert img.shape[2] % self.patch_size == 0 and img.shape[3] % self.patch_size == 0
            if cache_index is not None and self.token_cache is not None and n == 1 and self.feat_type == "feat":
                feat, qkv = self.token_cache.fetch(cache_index, img, lambda sub: self._tokens(sub, 1)[0]), None
            else:
                feat, qkv = self._tokens(img, n)
            fh, fw = img.shape[2] // self.patch_size, img.shape[3] // self.patch_size
            if return_class_feat:
                return feat[:, :1, :].reshape(feat.shape[0], 1, 1, -1).permute(0, 3, 1, 2)
            if self.feat_type == "feat":
                image_feat = feat[:, 1:, :].reshape(feat.shape[0], fh, fw, -1).permute(0, 3, 1, 2)
            elif self.feat_type == "KK":
                k = qkv[1, :, :, 1:, :]
                Bn, heads, _, d = k.shape
                image_feat = k.reshape(Bn, heads, fh, fw, d).permute(0, 1, 4, 2, 3).reshape(Bn, heads * d, fh, fw)
            else:
                raise ValueError("Unknown feat type:{}".format(self.feat_type))

        if self.proj_type is not None:
            if self._native_head_ok(image_feat):
                amax = None
                if cache_index is not None and self.token_cache is not None and self.token_cache.last is not None and \
                        self.token_cache.last[0] is feat:
                    amax = self.token_cache.last[1]
                return self._head_native(image_feat, amax)
            code = self._head(image_feat)
        else:
            code = image_feat
        return (self.dropout(image_feat) if self.cfg.dropout else image_feat), code

    def _native_head_ok(self, image_feat):
        """The hand-written head kernels (include/stego_head.h) take channels-last fp32 token maps on a HIP device; cfg.native_head
        (default True) turns them off."""
        if not (image_feat.is_cuda and image_feat.dtype == torch.float32 and image_feat.dim() == 4 and image_feat.stride(1) == 1
                and getattr(self.cfg, "native_head", True) and image_feat.shape[1] % 32 == 0 and self.dim <= 128
                and image_feat.stride(3) % 4 == 0 and image_feat.stride(0) % 4 == 0
                and image_feat.stride(2) == image_feat.shape[3] * image_feat.stride(3)):
            return False
        # the limits of head_check (csrc/head_fused.hip: at least 16 tokens per image, 32-bit byte offsets): maps beyond them - a 2 x 2
        # token map of a 32-pixel crop, say - run through the torch head (self._head), as they did before the native head existed
        B, C, fh, fw = image_feat.shape
        return fh * fw >= 16 and B * fh * fw < (1 << 30) and B * fh * fw * C < (1 << 30)

    def _head_native(self, image_feat, tokens_amax=None):
        """forward()'s tail on the native head: (feats, code) exactly as modules.py:108-116 returns them.  The three Dropout2d draws are
        made with the torch calls F.dropout2d makes, in the reference's order (cluster1's input :109, cluster2's :111, the returned
        map :114), so a seeded run consumes the generator like the reference; the masks go to the kernel as [B, C] scale vectors."""
        B, C, fh, fw = image_feat.shape
        tok = image_feat.permute(0, 2, 3, 1).reshape(B, fh * fw, C) if image_feat.is_contiguous(memory_format=torch.channels_last) \
            else image_feat.permute(0, 2, 3, 1).flatten(1, 2)          # a VIEW [B, hw, C] (the class token stays skipped by strides)
        nonlinear = self.proj_type == "nonlinear"
        m1 = m2 = m3 = None
        if self.training:
            n_draw = 1 + int(nonlinear) + int(bool(self.cfg.dropout))
            noise = self._feature_noises(image_feat, n_draw)           # the reference's draws, in its order, one launch when possible
            m1 = noise[0]
            if nonlinear:
                m2 = noise[1]
            if self.cfg.dropout:
                m3 = noise[n_draw - 1]
            if m2 is None:
                m2 = m1                                                 # (unused by a linear head; keeps the argument list dense)
            if m3 is None:
                m3 = torch.ones_like(m1) if self.cfg.dropout else m1
        want_feats = bool(self.cfg.dropout and self.training)
        c1 = self.cluster1[0]
        w1, b1 = c1.weight.view(c1.out_channels, c1.in_channels), c1.bias
        w21 = b21 = w22 = b22 = None
        if nonlinear:
            c20, c22 = self.cluster2[0], self.cluster2[2]
            w21, b21 = c20.weight.view(c20.out_channels, c20.in_channels), c20.bias
            w22, b22 = c22.weight.view(c22.out_channels, c22.in_channels), c22.bias
        ext = capi.torchglue() if getattr(self.cfg, "native_autograd", True) else None
        if ext is not None and hasattr(ext, "head"):       # the same op as a C++ autograd function (csrc/torch_glue_ext.cpp)
            outs = ext.head(tok, m1, m2, m3, w1, b1, w21, b21, w22, b22, want_feats, tokens_amax)
            code, feats = outs[0], (outs[1] if len(outs) > 1 else None)
        else:
            code, feats = _NativeHeadFunction.apply(tok, m1, m2, m3, w1, b1, w21, b21, w22, b22, want_feats, tokens_amax)
        code = code.view(B, fh, fw, self.dim).permute(0, 3, 1, 2)
        feats = feats.view(B, fh, fw, C).permute(0, 3, 1, 2) if feats is not None else image_feat
        return feats, code

    def _feature_noises(self, x, n):
        """n consecutive Dropout2d channel masks [B, C] for x: the torch calls (_feature_noise), or - on a HIP device with the torch glue
        extension - the same numbers from ONE launch (stego_ref_dropout_masks: 3 masks are 9 tiny torch kernels otherwise), the generator
        advanced identically.  Checked against the real torch calls once per process and size (_mask_variant); any mismatch, a missing
        extension or cfg.one_launch_draws = False keep the torch calls."""
        B, C = x.shape[0], x.shape[1]
        q = 1.0 - float(self.dropout.p)
        if x.is_cuda and 0.0 < q <= 1.0 and getattr(self.cfg, "one_launch_draws", True) and capi.torchglue() is not None:
            v = _mask_variant(self, x, n, q)
            if v >= 0:
                from .modules import _device_generator
                return capi.ref_dropout_masks(_device_generator(x.device), n, B * C, q, v, x.device)
        return [self._feature_noise(x).view(B, C) for _ in range(n)]

    def _feature_noise(self, x):
        """The channel mask of nn.Dropout2d, drawn exactly as ATen's feature dropout draws it (a [B,C,1,1] tensor filled
        by bernoulli_(1-p), then divided by 1-p), so a seeded run consumes the generator like the reference's
        self.dropout(image_feat) at modules.py:109-111."""
        p = self.dropout.p
        return x.new_empty(x.shape[0], x.shape[1], 1, 1).bernoulli_(1 - p).div_(1 - p)

    def _head(self, image_feat):
        """code = cluster1(dropout(f)) [+ cluster2(dropout(f))]  (modules.py:108-112).  The token features arrive as a
        channels-last view, so the 1x1 convolutions are GEMMs over the token matrix: F.linear with the conv weights
        viewed as [out, in] (same parameters, same state dict) instead of MIOpen convolutions on a strided NCHW view
        (measured 2.3 -> see tools/bench_step.py); the code map leaves as a channels-last view too, the layout the loss
        kernels read with one run per tap."""
        if image_feat.stride(1) != 1:                      # e.g. feat_type 'KK': plain convolutions
            code = self.cluster1(self.dropout(image_feat))
            if self.proj_type == "nonlinear":
                code = code + self.cluster2(self.dropout(image_feat))
            return code
        B, C = image_feat.shape[:2]
        tok = image_feat.permute(0, 2, 3, 1)               # [B, h, w, C]

        def drop(t):
            return t * self._feature_noise(image_feat).view(B, 1, 1, C) if self.training else t

        def lin(conv, t):
            return _TokenLinear.apply(t, conv.weight.view(conv.out_channels, conv.in_channels), conv.bias)

        code = lin(self.cluster1[0], drop(tok))
        if self.proj_type == "nonlinear":
            code = code + lin(self.cluster2[2], F.relu(lin(self.cluster2[0], drop(tok))))
        return code.permute(0, 3, 1, 2)


class ResizeAndClassify(nn.Module):
    def __init__(self, dim: int, size: int, n_classes: int):
        super().__init__()
        self.size = size
        self.predictor = nn.Sequential(nn.Conv2d(dim, n_classes, (1, 1)), nn.LogSoftmax(1))

    def forward(self, x):
        return F.interpolate(self.predictor(x), self.size, mode="bilinear", align_corners=False)


class ClusterLookup(nn.Module):
    """Cosine k-means probe on the code (trained on detached codes)."""

    def __init__(self, dim: int, n_classes: int):
        super().__init__()
        self.n_classes = n_classes
        self.dim = dim
        self.clusters = nn.Parameter(torch.randn(n_classes, dim))

    def reset_parameters(self):
        with torch.no_grad():
            self.clusters.copy_(torch.randn(self.n_classes, self.dim))

    def forward(self, x, alpha, log_probs=False):
        inner = torch.einsum("bchw,nc->bnhw", F.normalize(x, dim=1), F.normalize(self.clusters, dim=1))
        if alpha is None:
            probs = F.one_hot(torch.argmax(inner, dim=1), self.clusters.shape[0]).permute(0, 3, 1, 2).to(torch.float32)
        else:
            probs = F.softmax(inner * alpha, dim=1)
        if log_probs:
            return F.log_softmax(inner * alpha, dim=1)
        return -(probs * inner).sum(1).mean(), probs


class DoubleConv(nn.Module):
    def __init__(self, in_channels, out_channels, mid_channels=None):
        super().__init__()
        mid = mid_channels or out_channels
        self.double_conv = nn.Sequential(
            nn.Conv2d(in_channels, mid, kernel_size=3, padding=1), nn.BatchNorm2d(mid), nn.ReLU(),
            nn.Conv2d(mid, out_channels, kernel_size=3, padding=1), nn.BatchNorm2d(out_channels), nn.ReLU())

    def forward(self, x):
        return self.double_conv(x)


class NetWithActivations(nn.Module):
    """Runs the children of `model` in order and keeps the activations of `layer_nums`."""

    def __init__(self, model, layer_nums):
        super().__init__()
        self.layers = nn.ModuleList(model.children())
        self.layer_nums = {(len(self.layers) + l) if l < 0 else l for l in layer_nums}

    def forward(self, x):
        acts = {}
        for i, layer in enumerate(self.layers):
            x = layer(x)
            if i in self.layer_nums:
                acts[i] = x
        return acts


class FeaturePyramidNet(nn.Module):
    """ResNet-trunk alternative featurizer (cfg.arch == 'feature-pyramid'); `cut_model` is the trunk whose
    children 5,6,7 give the 28/14/7-resolution activations.  forward(x) -> (low_res_feats, clusters)."""

    @staticmethod
    def _helper(x):
        return F.interpolate(x, 56, mode="bilinear", align_corners=False).unsqueeze(-1)

    def make_clusterer(self, in_channels):
        return nn.Sequential(nn.Conv2d(in_channels, self.dim, (1, 1)), LambdaLayer(FeaturePyramidNet._helper))

    def make_nonlinear_clusterer(self, in_channels):
        return nn.Sequential(nn.Conv2d(in_channels, in_channels, (1, 1)), nn.ReLU(),
                             nn.Conv2d(in_channels, in_channels, (1, 1)), nn.ReLU(),
                             nn.Conv2d(in_channels, self.dim, (1, 1)), LambdaLayer(FeaturePyramidNet._helper))

    def __init__(self, granularity, cut_model, dim, continuous):
        super().__init__()
        assert granularity in {1, 2, 3, 4}
        self.layer_nums = [5, 6, 7]
        self.spatial_resolutions = [7, 14, 28, 56]
        self.feat_channels = [2048, 1024, 512, 3]
        self.extra_channels = [128, 64, 32, 32]
        self.granularity = granularity
        self.encoder = NetWithActivations(cut_model, self.layer_nums)
        self.dim = dim
        self.continuous = continuous
        self.n_feats = self.dim
        self.up = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False)
        fc, ec = self.feat_channels, self.extra_channels
        self.cluster1 = self.make_clusterer(fc[0])
        self.cluster1_nl = self.make_nonlinear_clusterer(fc[0])
        if granularity >= 2:
            self.conv2 = DoubleConv(fc[0] + fc[1], ec[1])
            self.cluster2 = self.make_clusterer(ec[1])
        if granularity >= 3:
            self.conv3 = DoubleConv(ec[1] + fc[2], ec[2])
            self.cluster3 = self.make_clusterer(ec[2])
        if granularity >= 4:
            self.conv4 = DoubleConv(ec[2] + fc[3], ec[3])
            self.cluster4 = self.make_clusterer(ec[3])

    def c(self, x, y):
        return torch.cat([x, y], dim=1)

    def forward(self, x):
        with torch.no_grad():
            feats = self.encoder(x)
        low = feats[self.layer_nums[-1]]
        heads = [self.cluster1(low)]
        f = low
        if self.granularity >= 2:
            f = self.conv2(self.c(self.up(f), feats[self.layer_nums[-2]]))
            heads.append(self.cluster2(f))
        if self.granularity >= 3:
            f = self.conv3(self.c(self.up(f), feats[self.layer_nums[-3]]))
            heads.append(self.cluster3(f))
        if self.granularity >= 4:
            size = self.spatial_resolutions[-1]
            f = self.conv4(self.c(self.up(f), F.interpolate(x, (size, size), mode="bilinear", align_corners=False)))
            heads.append(self.cluster4(f))
        avg = torch.cat(heads, 4).mean(4)
        return low, (avg if self.continuous else torch.log_softmax(avg, 1))


class Decoder(nn.Module):
    def __init__(self, code_channels, feat_channels):
        super().__init__()
        self.linear = nn.Conv2d(code_channels, feat_channels, (1, 1))
        self.nonlinear = nn.Sequential(nn.Conv2d(code_channels, code_channels, (1, 1)), nn.ReLU(),
                                       nn.Conv2d(code_channels, code_channels, (1, 1)), nn.ReLU(),
                                       nn.Conv2d(code_channels, feat_channels, (1, 1)))

    def forward(self, x):
        return self.linear(x) + self.nonlinear(x)


class ContrastiveCRFLoss(nn.Module):
    """Optional CRF-style pairwise loss on randomly chosen pixels (cfg.crf_weight, 0 by default)."""

    def __init__(self, n_samples, alpha, beta, gamma, w1, w2, shift):
        super().__init__()
        self.n_samples, self.alpha, self.beta, self.gamma = n_samples, alpha, beta, gamma
        self.w1, self.w2, self.shift = w1, w2, shift

    def forward(self, guidance, clusters):
        dev = clusters.device
        assert guidance.shape[0] == clusters.shape[0] and guidance.shape[2:] == clusters.shape[2:]
        h, w = guidance.shape[2], guidance.shape[3]
        coords = torch.cat([torch.randint(0, h, size=[1, self.n_samples], device=dev),
                            torch.randint(0, w, size=[1, self.n_samples], device=dev)], 0)
        g = guidance[:, :, coords[0], coords[1]]
        d_xy = (coords.unsqueeze(-1) - coords.unsqueeze(1)).square().sum(0).unsqueeze(0)
        d_g = (g.unsqueeze(-1) - g.unsqueeze(2)).square().sum(1)
        kernel = self.w1 * torch.exp(-d_xy / (2 * self.alpha) - d_g / (2 * self.beta)) + \
            self.w2 * torch.exp(-d_xy / (2 * self.gamma)) - self.shift
        c = clusters[:, :, coords[0], coords[1]]
        return -(torch.einsum("nka,nkb->nab", c, c) * kernel)
