import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from stego_amd import capi
capi.debug_set("STEGO_DEBUG", 1 << 18)
for B in (2, 8, 9, 32):
    rng = np.random.default_rng(1)
    a = rng.standard_normal((B, 384, 28, 28)).astype(np.float32)
    b = rng.standard_normal((B, 384, 28, 28)).astype(np.float32)
    ta = torch.from_numpy(a).cuda().permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    tb = torch.from_numpy(b).cuda().permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    for rep in range(3):
        out = capi.dense_corr(ta, tb, normalize=True).reshape(B, 784, 784)
        nan = torch.isnan(out)
        an = torch.nn.functional.normalize(ta, dim=1, eps=1e-10); bn = torch.nn.functional.normalize(tb, dim=1, eps=1e-10)
        ref = torch.einsum("nchw,ncij->nhwij", an, bn).reshape(B, 784, 784)
        err = (out - ref).abs()
        err[nan] = 0
        bad = (err > 1e-4)
        print("B", B, "rep", rep, "nan", int(nan.sum()), "bad", int(bad.sum()), "max err", float(err.max()))
        if nan.any() or bad.any():
            m = nan | bad
            idx = m.nonzero()
            print("  images", sorted(set(idx[:, 0].tolist()))[:10], "rows", int(idx[:, 1].min()), int(idx[:, 1].max()), "cols", int(idx[:, 2].min()), int(idx[:, 2].max()),
                  "col blocks", sorted(set((idx[:, 2] // 128).tolist())), "row blocks", sorted(set((idx[:, 1] // 128).tolist())))
B = 2
rng = np.random.default_rng(1)
a = rng.standard_normal((B, 384, 28, 28)).astype(np.float32); b = rng.standard_normal((B, 384, 28, 28)).astype(np.float32)
ta = torch.from_numpy(a).cuda().permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2); tb = torch.from_numpy(b).cuda().permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
out = capi.dense_corr(ta, tb, normalize=True).reshape(B, 784, 784)
an = torch.nn.functional.normalize(ta, dim=1, eps=1e-10); bn = torch.nn.functional.normalize(tb, dim=1, eps=1e-10)
ref = torch.einsum("nchw,ncij->nhwij", an, bn).reshape(B, 784, 784)
bad = ((out - ref).abs() > 1e-4) | torch.isnan(out)
print("bad fraction per (row block, col block), image 0:")
for i in range(7):
    print(" ".join("%.2f" % float(bad[0, 128 * i:128 * (i + 1), 128 * j:128 * (j + 1)].float().mean()) for j in range(7)))
