"""F16X3 backbone: error vs the fp64 torch model next to torch fp32's own and the F16 mode's; time at B = 64."""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from stego_amd import dino_vit, vit_native

dev = torch.device("cuda:0")
def rel(a, b): return float((a.double() - b.double()).norm() / b.double().norm())

for arch, patch, size, B in (("vit_tiny", 16, 96, 5), ("vit_small", 8, 224, 3), ("vit_small", 16, 224, 2), ("vit_base", 8, 320, 1)):
    torch.manual_seed(5)
    model = dino_vit.ARCHS[arch](patch_size=patch).to(dev).eval()
    with torch.no_grad():
        for name, prm in model.named_parameters():
            if prm.dim() == 1: prm.add_(0.05 * torch.randn_like(prm))
            if "qkv.weight" in name: prm.mul_(4.0)
    img = torch.randn(B, 3, size, size, device=dev)
    with torch.no_grad():
        ref32 = model.get_intermediate_feat(img, n=1)[0][0]
        m64 = dino_vit.ARCHS[arch](patch_size=patch).to(dev).double().eval()
        m64.load_state_dict({k: v.double() for k, v in model.state_dict().items()})
        ref64 = m64.get_intermediate_feat(img.double(), n=1)[0][0]
        del m64
    out = {"arch": arch, "patch": patch, "size": size, "B": B, "torch_fp32_vs_fp64": rel(ref32, ref64)}
    for prec in ("f16x3", "f16"):
        got = vit_native.NativeViT(model, precision=prec).forward_tokens(img)
        out[prec + "_vs_fp64"] = rel(got, ref64)
        out[prec + "_vs_fp32"] = rel(got, ref32)
        out[prec + "_finite"] = bool(torch.isfinite(got).all())
        out[prec + "_maxabs_vs_fp64"] = float((got.double() - ref64).abs().max())
    out["fp32_maxabs_vs_fp64"] = float((ref32.double() - ref64).abs().max())
    print(json.dumps(out), flush=True)

if os.environ.get("TIME", "1") == "1":
    torch.manual_seed(0)
    model = dino_vit.vit_small(patch_size=8).to(dev).eval()
    img = torch.randn(64, 3, 224, 224, device=dev)
    for prec in ("f16x3", "f16"):
        nat = vit_native.NativeViT(model, precision=prec)
        nat.forward_tokens(img); torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(5): nat.forward_tokens(img)
        t1.record(); torch.cuda.synchronize()
        print(json.dumps({"precision": prec, "B": 64, "ms": t0.elapsed_time(t1) / 5}), flush=True)
