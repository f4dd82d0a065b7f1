#!/usr/bin/env python
"""Frozen DINO backbone forward: native kernels vs torch on the same GPU and vs the CPU.
One JSON line.  FLOPs counted: patch GEMM + per block (qkv, proj, fc1, fc2 GEMMs + QK^T + PV)."""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stego_amd import dino_vit, vit_native

ap = argparse.ArgumentParser()
ap.add_argument("--arch", default="vit_small"); ap.add_argument("--patch", type=int, default=8)
ap.add_argument("--size", type=int, default=224); ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--iters", type=int, default=10); ap.add_argument("--no-cpu", action="store_true")
ap.add_argument("--no-torch", action="store_true")
ap.add_argument("--precision", default="f16x3", choices=["f16x3", "f16"])
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = dino_vit.ARCHS[a.arch](patch_size=a.patch).to(dev).eval()
img = torch.randn(a.batch, 3, a.size, a.size, device=dev)
D, depth = model.embed_dim, len(model.blocks)
hw = (a.size // a.patch) ** 2; ntok = hw + 1
flops = a.batch * (2 * hw * D * 3 * a.patch ** 2 + depth * (2 * ntok * D * (3 * D + D + 8 * D) + 4 * ntok * ntok * D))

def timed(fn, iters):
    fn(); torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters): fn()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / iters

nat = vit_native.NativeViT(model, precision=a.precision)
x3 = a.precision == "f16x3"
peak = 2500.0 / 3 if x3 else 2500.0      # f16x3: three MFMAs per product
ms_native = timed(lambda: nat.forward_tokens(img), a.iters)
out = {"metric": "frozen DINO %s/%d forward, %dx%d" % (a.arch, a.patch, a.size, a.size), "batch": a.batch, "unit": "images/s",
       "value": a.batch / ms_native * 1e3, "ms": ms_native, "effective_TFLOPs": flops / ms_native / 1e9,
       "roofline": {"bound": "mfma", "achieved": flops / ms_native / 1e9, "peak": peak, "unit": "TFLOP/s",
                    "frac": flops / ms_native / 1e9 / peak, "algorithmic_flops": flops,
                    "note": "f16x3: dense fp16 peak / 3 (three MFMAs per product)" if x3 else "dense fp16 peak"},
       "precision": a.precision,
       "dtype": ("split-fp16 operands (hi + lo, three MFMAs per product: the fp32 class)" if x3 else "fp16 operands") +
                " on the matrix cores, fp32 accumulate / statistics / residual"}
if not a.no_torch:
    with torch.no_grad():
        ms32 = timed(lambda: model.get_intermediate_feat(img, n=1), max(2, a.iters // 3))
        with torch.autocast("cuda", dtype=torch.float16):
            ms16 = timed(lambda: model.get_intermediate_feat(img, n=1), max(2, a.iters // 3))
        ref = model.get_intermediate_feat(img, n=1)[0][0]
    got = nat.forward_tokens(img)
    out["torch_same_gpu"] = {"fp32_ms": ms32, "fp16_autocast_sdpa_ms": ms16, "speedup_vs_fp32": ms32 / ms_native,
                             "speedup_vs_fp16_autocast": ms16 / ms_native}
    out["rel_l2_vs_torch_fp32"] = float((got - ref).norm() / ref.norm())
    if a.batch <= 16:       # fp64 twin of the model: whose error is whose
        with torch.no_grad():
            m64 = dino_vit.ARCHS[a.arch](patch_size=a.patch).to(dev).double().eval()
            m64.load_state_dict({k: v.double() for k, v in model.state_dict().items()})
            ref64 = m64.get_intermediate_feat(img.double(), n=1)[0][0]
        out["rel_l2_vs_fp64"] = {"native": float((got.double() - ref64).norm() / ref64.norm()),
                                 "torch_fp32": float((ref.double() - ref64).norm() / ref64.norm())}
if not a.no_cpu:
    cm = dino_vit.ARCHS[a.arch](patch_size=a.patch).eval()
    cm.load_state_dict(model.state_dict())
    nb = 2
    ci = img[:nb].cpu()
    torch.set_num_threads(os.cpu_count())
    with torch.no_grad():
        cm.get_intermediate_feat(ci, n=1)
        t = time.perf_counter(); cm.get_intermediate_feat(ci, n=1); dt = time.perf_counter() - t
    out["cpu_baseline"] = {"value": nb / dt, "unit": "images/s", "cores": os.cpu_count(), "kind": "port",
                           "sample": "stego_amd.dino_vit (torch fp32 mirror of dino/vision_transformer.py) on %d images, %.2f s" % (nb, dt)}
print(json.dumps(out))
