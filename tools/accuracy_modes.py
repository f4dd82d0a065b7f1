import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from oracle import corr_oracle as O
from stego_amd import modules as M
import copy
B, C, H, W, K, S, n_neg = 8, 384, 28, 28, 70, 11, 5
d = O.synth_inputs(B, C, H, W, K, S, n_neg, seed=2024, dino_like=True)
cfg = O.CorrCfg()
inputs = {k: d[k] for k in ("feats", "feats_pos", "code", "code_pos", "coords1", "coords2")}
ref = O.corr_loss_forward(**inputs, perms=d["perms"], cfg=cfg)
dev = "cuda:0"
def cl(t): return t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
for prec in ("f32", "f16x3"):
    c = copy.copy(cfg); c.corr_precision = prec
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in inputs.items()}
    out = M.ContrastiveCorrelationLoss(c).forward_explicit(cl(t["feats"]), cl(t["feats_pos"]), cl(t["code"]), cl(t["code_pos"]),
                                                            t["coords1"], t["coords2"], torch.from_numpy(d["perms"]).to(dev))
    nl = out[4].cpu().numpy().astype(np.float64)
    err = np.abs(nl - ref.neg_inter_loss)
    print(prec, "neg_loss abs err: max %.3e mean %.3e (|loss| mean %.3e)" % (err.max(), err.mean(), np.abs(ref.neg_inter_loss).mean()))
