#!/usr/bin/env python
"""The product-path step (reference draws + reference composition + backward) captured in a HIP graph and replayed 60 times, for
rocprofv3 --kernel-trace (CSV) -> tools/kernel_gaps.py: which kernels a replay holds, how long each takes, the gap in front of each."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from stego_amd.modules import ContrastiveCorrelationLoss
dev = torch.device("cuda:0")
cfg = bench.Cfg()
C, H, W, K = bench.WORKLOADS["vits8_224"]
total = len(sys.argv) > 1 and sys.argv[1] == "total"
d = bench.make_inputs(32, C, H, W, K, 11, 5, 1, dev)
loss_fn = ContrastiveCorrelationLoss(cfg)
c = d["code"].detach().clone().requires_grad_(True); cp = d["code_pos"].detach().clone().requires_grad_(True)
def step():
    c.grad = None; cp.grad = None
    if total:
        loss_fn.total(d["feats"], d["feats_pos"], None, None, c, cp, (cfg.pos_intra_weight, cfg.pos_inter_weight, cfg.neg_inter_weight))[0].backward()
        return
    (pil, _, pel, _, nl, _) = loss_fn(d["feats"], d["feats_pos"], None, None, c, cp)
    (cfg.pos_intra_weight * pil + cfg.pos_inter_weight * pel + cfg.neg_inter_weight * nl.mean()).backward()
for _ in range(3): step()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    step()
torch.cuda.synchronize()
for _ in range(60): g.replay()
torch.cuda.synchronize()
