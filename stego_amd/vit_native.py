"""Native (hand-written HIP, include/stego_vit.h) forward of the frozen DINO backbone.

Host-side mirror of what ``DinoFeaturizer.forward`` needs from the reference's
``VisionTransformer.get_intermediate_feat(img, n=1)`` (src/dino/vision_transformer.py:225-237, called at
src/modules.py:88): the normalised tokens of the last block.  The reference also returns the attention maps and the
qkv tensor of that block; ``feat_type == "feat"`` (the default, train_config.yml) never looks at them, so this path
does not build them - callers that want ``feat_type == "KK"`` or ``n > 1`` keep using the torch module.

Weights are packed once per (H, W) from a ``dino_vit.VisionTransformer`` (same parameter names as the DINO
checkpoints); torch is used for device memory and for the bicubic ``interpolate_pos_encoding`` only.
"""
import ctypes

import torch

from . import capi


def _params_of(model, H, W):
    """The fp32 tensors stego_vit_pack_weights() takes, in its order (see include/stego_vit.h)."""
    with torch.no_grad():
        D = model.embed_dim
        ps = model.patch_embed.patch_size
        probe = torch.empty(1, 1 + (H // ps) * (W // ps), D, device=model.pos_embed.device)
        pos = model.interpolate_pos_encoding(probe, H, W)[0]          # vision_transformer.py:171-193
        out = [model.patch_embed.proj.weight.reshape(D, -1), model.patch_embed.proj.bias, model.cls_token.reshape(D), pos]
        for blk in model.blocks:
            out += [blk.norm1.weight, blk.norm1.bias, blk.attn.qkv.weight, blk.attn.qkv.bias, blk.attn.proj.weight,
                    blk.attn.proj.bias, blk.norm2.weight, blk.norm2.bias, blk.mlp.fc1.weight, blk.mlp.fc1.bias,
                    blk.mlp.fc2.weight, blk.mlp.fc2.bias]
        out += [model.norm.weight, model.norm.bias]
        return [t.detach().to(torch.float32).contiguous() for t in out]


def supported(model):
    D = model.embed_dim
    blk = model.blocks[0]
    return (D % 64 == 0 and D <= 768 and blk.attn.num_heads * 64 == D and blk.attn.qkv.bias is not None and
            model.patch_embed.patch_size in (8, 16) and blk.mlp.fc1.out_features % 64 == 0)


PRECISIONS = {"f16x3": capi.VIT_F16X3, "f16": capi.VIT_F16}


class NativeViT:
    """Packed weights + workspace cache around stego_vit_forward for one frozen ``dino_vit.VisionTransformer``.

    precision "f16x3" (default): split-fp16 operands, three MFMAs per product - the fp32 class of the reference's torch model;
    "f16": plain fp16 operands, 2 - 3 x faster, error at the level of torch's fp16 autocast (include/stego_vit.h)."""

    def __init__(self, model, precision="f16x3"):
        if not supported(model):
            raise RuntimeError("stego_vit: unsupported ViT geometry (need head_dim 64, qkv bias, patch 8/16, D <= 768)")
        if precision not in PRECISIONS:
            raise ValueError("stego_vit: unknown precision %r (f16x3 | f16)" % (precision,))
        self.precision = precision
        self.model = model
        self._packed = {}         # (H, W, device) -> uint8 blob
        self._ws = {}             # (B, H, W, device) -> uint8 workspace

    def _desc(self, B, H, W):
        m = self.model
        return capi.StegoVitDesc(B, H, W, m.patch_embed.patch_size, m.embed_dim, len(m.blocks), m.blocks[0].attn.num_heads,
                                 m.blocks[0].mlp.fc1.out_features, PRECISIONS[self.precision])

    def shape_supported(self, B, H, W):
        """Does this build have kernels for a [B,3,H,W] batch through this backbone?  (include/stego_vit.h: H, W multiples of the patch
        size, B * tokens < 2^20, ...)  A host call: the caller keeps the torch module for what it rejects."""
        if B <= 0 or H <= 0 or W <= 0:
            return False
        return int(capi.load().stego_vit_workspace_bytes(ctypes.byref(self._desc(B, H, W)))) > 0

    def invalidate(self):
        """Call after loading new weights into the torch module."""
        self._packed.clear()

    def _weights(self, desc, dev):
        key = (desc.H, desc.W, dev)
        blob = self._packed.get(key)
        if blob is None:
            lib = capi.load()
            params = [p.to(dev) for p in _params_of(self.model, desc.H, desc.W)]
            n = lib.stego_vit_param_count(ctypes.byref(desc))
            assert n == len(params), (n, len(params))
            arr = (ctypes.c_void_p * n)(*[p.data_ptr() for p in params])
            nbytes = lib.stego_vit_weights_bytes(ctypes.byref(desc))
            blob = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                capi._check(lib.stego_vit_pack_weights(ctypes.byref(desc), arr, n, blob.data_ptr(), nbytes, capi._stream()))
                torch.cuda.current_stream().synchronize()      # `params` may be temporaries: keep them alive until packed
            self._packed[key] = blob
        return blob

    def forward_tokens(self, img):
        """img fp32 [B,3,H,W] on a HIP device -> tokens fp32 [B, 1 + hw, D] (= feat[0] of get_intermediate_feat(n=1))."""
        capi._require_dev(img)
        if img.dtype != torch.float32 or img.dim() != 4 or img.shape[1] != 3:
            raise RuntimeError("stego_vit: expected a float32 [B,3,H,W] image batch, got %s %s" % (img.dtype, tuple(img.shape)))
        img = img.contiguous()
        B, _, H, W = img.shape
        lib = capi.load()
        desc = self._desc(B, H, W)
        dev = img.device
        blob = self._weights(desc, dev)
        # ONE workspace per device, sized for the largest request seen (a smaller batch fits in it): TokenCache partial-miss
        # batches arrive with every B' from 1 to B and must not leave B large buffers behind
        need = int(lib.stego_vit_workspace_bytes(ctypes.byref(desc)))
        if need == 0:
            raise RuntimeError("stego_vit: unsupported shape B=%d %dx%d for this backbone (see include/stego_vit.h)" % (B, H, W))
        ws = self._ws.get(dev)
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
            self._ws[dev] = ws
        ps = self.model.patch_embed.patch_size
        out = torch.empty(B, 1 + (H // ps) * (W // ps), self.model.embed_dim, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            capi._check(lib.stego_vit_forward(ctypes.byref(desc), blob.data_ptr(), img.data_ptr(), out.data_ptr(), ws.data_ptr(),
                                              ws.numel(), capi._stream()))
        return out
