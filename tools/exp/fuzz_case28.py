"""Fuzz case 28 of `FUZZ_S_MAX=16 tools/fuzz_fused.py 40 7` (B = 3, C = 384, 17 x 13, K = 44, S = 14, no negatives): where do the gradient entries
that miss the bar sit, and how close to the clamp bound is the nearest cd element of the oracle?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import corr_oracle as O
from stego_amd import modules as M
B, C, H, W, K, S, n_neg = 3, 384, 17, 13, 44, 14, 0
case = 28
# replay the fuzz tool's draws up to the case to get the same dino flag: simpler - try both
for dino in (False, True):
    d = O.synth_inputs(B, C, H, W, K, S, n_neg, seed=1000 + case, dino_like=dino)
    cfg = O.CorrCfg(feature_samples=S, neg_samples=n_neg, pointwise=True, zero_clamp=True, stabalize=False)
    cfg.corr_precision = "f32"
    inputs = {k: d[k] for k in ("feats", "feats_pos", "code", "code_pos", "coords1", "coords2")}
    dev = torch.device("cuda:0")
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in inputs.items()}
    for k in ("feats", "feats_pos", "code", "code_pos"):
        t[k] = t[k].permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    code = t["code"].detach().requires_grad_(True); code_pos = t["code_pos"].detach().requires_grad_(True)
    out = M.ContrastiveCorrelationLoss(cfg).forward_explicit(t["feats"], t["feats_pos"], code, code_pos, t["coords1"], t["coords2"], None)
    (0.67 * out[0] + 0.25 * out[2]).backward()
    ref = O.corr_loss_forward(**inputs, perms=d["perms"], cfg=cfg)
    dc, dcp = O.corr_loss_backward(**inputs, perms=d["perms"], cfg=cfg, g_intra=0.67, g_inter=0.25, g_neg_loss=None)
    for got, want, cdref, cdgot, what in ((code.grad.cpu().numpy(), dc, ref.pos_intra_cd, out[1], "d_code"), (code_pos.grad.cpu().numpy(), dcp, ref.pos_inter_cd, out[3], "d_code_pos")):
        want = np.asarray(want, dtype=np.float64)
        tol = 1e-3 * np.abs(want).mean() + 1e-3 * np.abs(want)
        badm = np.abs(got - want) > tol
        pix = np.argwhere(badm.any(axis=1))            # (b, y, x) with any channel off
        print(dino, what, "entries off %.4f %%" % (100 * badm.mean()), "pixels", len(pix), "of", B * H * W, "images", sorted(set(pix[:, 0].tolist())))
    for cdref, cdgot, what in ((ref.pos_intra_cd, out[1], "intra"), (ref.pos_inter_cd, out[3], "inter")):
        g = cdgot.detach().cpu().numpy().astype(np.float64)
        flips = np.argwhere((cdref >= 0) != (g >= 0))
        print(dino, what, "cd sign flips vs the oracle:", len(flips), "closest |cd| of the oracle %.3e" % np.abs(cdref).min(), [ (tuple(f), float(cdref[tuple(f)]), float(g[tuple(f)])) for f in flips[:3]])
