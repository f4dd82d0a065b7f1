#!/usr/bin/env python
"""Where a training step's time goes (BASELINE config 2: B = 32 pairs of 224^2 images, ViT-S/8, dim 70):
frozen backbone on img + img_pos -> segmentation head -> correspondence loss fwd+bwd -> head backward.
Compares the native backbone with the torch backbone around the same (native) loss.  One JSON line."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stego_amd import featurizers, modules


class Cfg:
    dino_patch_size = 8; dino_feat_type = "feat"; model_type = "vit_small"; projection_type = "nonlinear"
    dropout = True; pretrained_weights = None; native_backbone = True
    pointwise = True; zero_clamp = True; stabalize = False; use_salience = False
    feature_samples = 11; neg_samples = 5; pos_intra_shift = .18; pos_inter_shift = .12; neg_inter_shift = .46
    corr_precision = "f16x3"


def timed(fn, iters=10):
    fn(); torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters): fn()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / iters


dev = torch.device("cuda:0")
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = Cfg()
net = featurizers.DinoFeaturizer(70, cfg).to(dev)
loss_fn = modules.ContrastiveCorrelationLoss(cfg)
img = torch.randn(B, 3, 224, 224, device=dev); img_pos = torch.randn(B, 3, 224, 224, device=dev)
both = torch.cat([img, img_pos])
params = [p for p in net.parameters() if p.requires_grad]


def step():
    feats_all, code_all = net(both)                       # one 2B batch through the frozen backbone + head
    feats, feats_pos = feats_all[:B], feats_all[B:]
    code, code_pos = code_all[:B], code_all[B:]
    out = loss_fn(feats, feats_pos, None, None, code, code_pos)
    loss = .67 * out[0] + .25 * out[2] + .63 * out[4].mean()
    for p in params: p.grad = None
    loss.backward()


res = {}
for native, prec in ((True, "f16x3"), (True, "f16"), (False, None)):
    cfg.native_backbone = native
    cfg.backbone_precision = prec or "f16x3"
    net._native = None                                    # (packed for one precision)
    key = ("native_backbone" if prec == "f16x3" else "native_backbone_f16") if native else "torch_fp32_backbone"
    with torch.no_grad():
        bb = timed(lambda: net._tokens(both, 1))
    res[key] = {"step_ms": timed(step), "backbone_ms": bb, "path": net.backbone_path, "precision": prec or "torch fp32"}
cfg.native_backbone = True
cfg.backbone_precision = "f16x3"
net._native = None
feats_all, code_all = net(both)
code = code_all[:B].detach().requires_grad_(True); code_pos = code_all[B:].detach().requires_grad_(True)


def loss_only():
    out = loss_fn(feats_all[:B], feats_all[B:], None, None, code, code_pos)
    (.67 * out[0] + .25 * out[2] + .63 * out[4].mean()).backward()


res["loss_fwd_bwd_eager_ms"] = timed(loss_only, 30)

# the same step with the backbone tokens of the (fixed-crop) dataset served from HBM (featurizers.TokenCache)
cache = net.enable_token_cache(2 * B, (224, 224), dev)
idx = torch.arange(2 * B, device=dev)


def step_cached():
    feats_all, code_all = net(both, cache_index=idx)
    out = loss_fn(feats_all[:B], feats_all[B:], None, None, code_all[:B], code_all[B:])
    loss = .67 * out[0] + .25 * out[2] + .63 * out[4].mean()
    for p in params: p.grad = None
    loss.backward()


cfg.native_head = False                 # the torch head (F.linear / bmm over the token matrix), round 2
res["cached_tokens_torch_head"] = {"step_ms": timed(step_cached, 30)}
cfg.native_head = True                  # the hand-written head (include/stego_head.h), round 3: the default
res["cached_tokens"] = {"step_ms": timed(step_cached, 30), "misses": cache.misses,
                        "table_MB": cache.tokens.numel() * 2 / 1e6}
tok = cache.tokens[idx].float()
image_feat = tok[:, 1:, :].reshape(2 * B, 28, 28, -1).permute(0, 3, 1, 2)
up = torch.randn(2 * B, 70, 28, 28, device=dev) / 784


def head_only(native):
    cfg.native_head = native
    f, c = net._head_native(image_feat) if native else (net.dropout(image_feat), net._head(image_feat))
    for p in params: p.grad = None
    (c * up).sum().backward()


res["head_fwd_bwd_ms"] = {"native": timed(lambda: head_only(True), 30), "torch": timed(lambda: head_only(False), 30)}
cfg.native_head = True
print(json.dumps({"metric": "training step, B=%d pairs, ViT-S/8 224^2 (backbone on 2B images + head + correspondence loss fwd+bwd + head bwd)" % B,
                  "unit": "ms", **res,
                  "pairs_per_s_native": B / res["native_backbone"]["step_ms"] * 1e3,
                  "pairs_per_s_torch_backbone": B / res["torch_fp32_backbone"]["step_ms"] * 1e3,
                  "pairs_per_s_cached_tokens": B / res["cached_tokens"]["step_ms"] * 1e3}))
