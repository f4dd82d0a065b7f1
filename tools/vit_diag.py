import sys, torch
sys.path.insert(0, "/root/repo")
from stego_amd import dino_vit, vit_native
def rel(a, b): return float((a.double() - b.double()).norm() / b.double().norm())
for arch, patch, H, W, B, depth in [("vit_base", 8, 224, 224, 1, 12), ("vit_base", 8, 224, 224, 1, 4), ("vit_small", 8, 224, 224, 2, 12),
                                    ("vit_tiny", 16, 96, 96, 5, 12)]:
    torch.manual_seed(5)
    kw = dict(vit_base=dict(embed_dim=768, num_heads=12), vit_small=dict(embed_dim=384, num_heads=6), vit_tiny=dict(embed_dim=192, num_heads=3))[arch]
    model = dino_vit.VisionTransformer(patch_size=patch, depth=depth, **kw).cuda().eval()
    with torch.no_grad():
        for name, prm in model.named_parameters():
            if prm.dim() == 1: prm.add_(0.05 * torch.randn_like(prm))
            if "qkv.weight" in name: prm.mul_(4.0)
    img = torch.randn(B, 3, H, W, device="cuda")
    with torch.no_grad():
        ref = model.get_intermediate_feat(img, n=1)[0][0]
    got = vit_native.NativeViT(model).forward_tokens(img)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        h16 = model.get_intermediate_feat(img, n=1)[0][0].float()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        b16 = model.get_intermediate_feat(img, n=1)[0][0].float()
    import copy
    m64 = copy.deepcopy(model).double()
    with torch.no_grad():
        r64 = m64.get_intermediate_feat(img.double(), n=1)[0][0]
    print("   vs fp64: native %.2e | torch fp32 %.2e | torch autocast fp16 %.2e | bf16 %.2e" % (rel(got, r64), rel(ref, r64), rel(h16, r64), rel(b16, r64)))
    e = (got - ref).norm(dim=2) / ref.norm(dim=2)
    print("%-9s p%-2d %dx%d B%d depth %2d: rel %.2e  per-token rel: cls %.1e  p50 %.1e  max %.1e (token %d)" % (
        arch, patch, H, W, B, depth, rel(got, ref), float(e[:, 0].max()), float(e.median()), float(e.max()), int(e.max(0)[0].argmax())))
