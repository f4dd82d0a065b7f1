"""Which outputs differ between the default and the STEGO_SHARED_DEVICE launch of the fused forward (same inputs)?"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from stego_amd import capi
dev = torch.device("cuda:0")
cfg = bench.Cfg()
C, H, W, K = bench.WORKLOADS["vits8_224"]
B, S, n_neg = 32, 11, 5
P = S * S
d = bench.make_inputs(B, C, H, W, K, S, n_neg, 1000, dev)
desc = capi.make_desc(B, C, K, H, W, S, n_neg, cfg, (.18, .12, .46), capi.PREC_F16X3)
def run():
    out = capi.corr_fwd(desc, d["feats"], d["feats_pos"], d["code"], d["code_pos"], d["coords1"], d["coords2"], d["perms"], True)
    torch.cuda.synchronize()
    return [o.clone() if torch.is_tensor(o) else o for o in out[:5]] + [out[5][0].clone()]
a = run(); a2 = run()
capi.set_shared_device(True); b = run(); b2 = run(); capi.set_shared_device(False)
names = ["means", "intra_cd", "inter_cd", "neg_loss", "neg_cd", "saved_w"]
for n, x, y, x2, y2 in zip(names, a, b, a2, b2):
    x = x.reshape(-1, P, P) if x.numel() % (P * P) == 0 else x.reshape(1, 1, -1)
    y = y.reshape(x.shape)
    print(n, "default repeatable:", bool((x.reshape(-1) == x2.reshape(-1)).all()), "shared repeatable:", bool((y.reshape(-1) == y2.reshape(-1)).all()))
    df = (x != y)
    if df.any():
        tiles = torch.nonzero(df.reshape(x.shape[0], -1).any(1)).flatten().tolist()
        t = tiles[0]
        rows = torch.nonzero(df[t].any(1)).flatten().tolist()
        cols = torch.nonzero(df[t].any(0)).flatten().tolist()
        print("  ", n, "tiles differing:", len(tiles), tiles[:12], "| tile", t, "rows", len(rows), rows[:10], "cols", len(cols), cols[:10],
              "max abs", float((x - y).abs().max()))
