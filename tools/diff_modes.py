"""Where do the two forward precisions disagree?  Prints the tile / row / column pattern of neg_loss mismatches."""
import sys, os, numpy as np, torch, copy
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from oracle import corr_oracle as O
from stego_amd import modules as M
B, C, H, W, K, S, n_neg = 4, 384, 28, 28, 70, 11, 5
d = O.synth_inputs(B, C, H, W, K, S, n_neg, seed=7, dino_like=False)
cfg = O.CorrCfg()
inputs = {k: d[k] for k in ("feats", "feats_pos", "code", "code_pos", "coords1", "coords2")}
dev = "cuda:0"
def cl(t): return t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
res = {}; cds = {}
for prec in ("f32", "f16x3", "f16x3"):
    capi.debug_set("STEGO_DEBUG", int(os.environ.get("DBG", "0") if prec == "f16x3" else "0"))
    c = copy.copy(cfg); c.corr_precision = prec
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in inputs.items()}
    out = M.ContrastiveCorrelationLoss(c).forward_explicit(cl(t["feats"]), cl(t["feats_pos"]), cl(t["code"]), cl(t["code_pos"]),
                                                            t["coords1"], t["coords2"], torch.from_numpy(d["perms"]).to(dev))
    nl = out[4].cpu().numpy().astype(np.float64)
    cdn = out[5].cpu().numpy().astype(np.float64)
    if prec in res:
        print("second f16x3 run identical to first:", np.array_equal(res[prec], nl), "cd:", np.array_equal(cds[prec], cdn))
    res[prec] = nl; cds[prec] = cdn
P = S * S
diff = np.abs(res["f32"] - res["f16x3"]).reshape(n_neg, B, P, P)
bad = np.argwhere(diff > 1e-4)
print("mismatches:", len(bad), "of", diff.size)
for n in range(n_neg):
    for b in range(B):
        m = diff[n, b] > 1e-4
        if m.any():
            rows = np.unique(np.nonzero(m)[0]); cols = np.unique(np.nonzero(m)[1])
            print("tile n=%d b=%d: %d bad; rows %s cols %s  max %.3e" % (n, b, m.sum(), rows[:12], cols[:12], diff[n, b].max()))

dcd = np.abs(cds["f32"] - cds["f16x3"]).reshape(n_neg, B, P, P)
print("neg_cd mismatches:", int((dcd > 1e-6).sum()), "max", dcd.max())
for n in range(n_neg):
    for b in range(B):
        m = dcd[n, b] > 1e-6
        if m.any():
            print("cd tile n=%d b=%d: %d bad; rows %s cols %s" % (n, b, m.sum(), np.unique(np.nonzero(m)[0])[:8], np.unique(np.nonzero(m)[1])[:8]))


import torch.nn.functional as F
ft = torch.from_numpy(d["feats"]).to(dev); fp = torch.from_numpy(d["feats_pos"]).to(dev)
c1 = torch.from_numpy(d["coords1"]).to(dev); c2 = torch.from_numpy(d["coords2"]).to(dev)
def samp(t, c): return F.grid_sample(t, c.permute(0, 2, 1, 3), padding_mode="border", align_corners=True)
fa = F.normalize(samp(ft, c1), dim=1, eps=1e-10).reshape(B, C, P)
cdv = cds["f32"].reshape(n_neg, B, P, P)
l32 = res["f32"].reshape(n_neg, B, P, P); l16 = res["f16x3"].reshape(n_neg, B, P, P)
shown = 0
for n in range(n_neg):
    perm = torch.from_numpy(d["perms"][n]).to(dev)
    fb = F.normalize(samp(fp[perm], c2), dim=1, eps=1e-10).reshape(B, C, P)
    fd = torch.einsum("bci,bcj->bij", fa, fb).cpu().numpy()
    for b in range(B):
        m = diff[n, b] > 1e-4
        for c in np.unique(np.nonzero(m)[1]):
            cl = np.clip(cdv[n, b, :, c], 0, None); ok = cl > 1e-3
            dl = (l16[n, b, :, c] - l32[n, b, :, c])[ok] / cl[ok]          # = -(fd16 - fd32) + rowmean diff
            f = fd[b][ok, c]; rows = np.nonzero(ok)[0]
            out = []
            for lo, hi in ((0, 64), (64, 128)):
                sel = (rows >= lo) & (rows < hi)
                if sel.sum() < 3: continue
                eps = -(dl[sel] * f[sel]).sum() / (f[sel] ** 2).sum()
                r = dl[sel] + eps * f[sel]
                out.append("rows %d-%d: eps %.4f resid/|d| %.2f" % (lo, hi, eps, np.sqrt((r**2).mean()) / np.sqrt((dl[sel]**2).mean())))
            print("col %3d tile(%d,%d): %s" % (c, n, b, " | ".join(out)))
            shown += 1
    if shown > 14: break
