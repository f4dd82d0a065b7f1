#!/usr/bin/env python
"""Kernel list of ONE product-path step (eager) for rocprofv3 --kernel-trace."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from stego_amd.modules import ContrastiveCorrelationLoss
dev = torch.device("cuda:0")
cfg = bench.Cfg()
C, H, W, K = bench.WORKLOADS["vits8_224"]
d = bench.make_inputs(32, C, H, W, K, 11, 5, 1, dev)
loss_fn = ContrastiveCorrelationLoss(cfg)
c = d["code"].detach().clone().requires_grad_(True); cp = d["code_pos"].detach().clone().requires_grad_(True)
def step():
    c.grad = None; cp.grad = None
    (pil, _, pel, _, nl, _) = loss_fn(d["feats"], d["feats_pos"], None, None, c, cp)
    (cfg.pos_intra_weight * pil + cfg.pos_inter_weight * pel + cfg.neg_inter_weight * nl.mean()).backward()
for _ in range(3): step()
torch.cuda.synchronize()
print("MARK", flush=True)
step()
torch.cuda.synchronize()
