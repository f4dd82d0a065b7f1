import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_parity_gpu as T
from conftest import assert_close
O, capi = T.O, T.capi
shape = dict(B=8, C=384, H=28, W=28, K=70, S=11, n_neg=5)
d = O.synth_inputs(seed=11, dino_like=True, **shape)
cfg = O.CorrCfg(feature_samples=11, neg_samples=5)
inputs = {k: d[k] for k in ("feats", "feats_pos", "code", "code_pos", "coords1", "coords2")}
ref = O.corr_loss_forward(**inputs, perms=d["perms"], cfg=cfg)
numel = 8 * 11 ** 4
dc, dcp = O.corr_loss_backward(**inputs, perms=d["perms"], cfg=cfg, g_intra=0.67, g_inter=0.25, g_neg_loss=np.full(ref.neg_inter_loss.shape, 0.63 / (5 * numel)))
for dbg in (16384, 0):
    capi.debug_set("STEGO_DEBUG", dbg)
    r = T._run(inputs, d["perms"], cfg, layout="cl", precision="f32")
    capi.debug_set("STEGO_DEBUG", 0)
    e = np.abs(r["d_code"] - dc)
    bad = e > 1e-3 * np.abs(dc).mean() + 1e-3 * np.abs(dc)
    idx = np.argwhere(bad)
    print("debug", dbg, "bad", int(bad.sum()), "max err", float(e.max()), "mean|dc|", float(np.abs(dc).mean()), "images with bad:", sorted(set(idx[:, 0].tolist())), "pixels:", len(set(map(tuple, idx[:, [0, 2, 3]].tolist()))))
    e2 = np.abs(r["d_code_pos"] - dcp)
    print("   d_code_pos max err", float(e2.max()), "mean", float(np.abs(dcp).mean()))
