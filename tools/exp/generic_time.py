"""Time of ContrastiveCorrelationLoss fwd + bwd through the product API at feature_samples the fused kernels do not cover."""
import os, sys, json, copy
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from stego_amd.modules import ContrastiveCorrelationLoss
dev = torch.device("cuda:0")
C, H, W, K = bench.WORKLOADS["vits8_224"]
B, n_neg = 32, 5
for S in (11, 12, 16):
    cfg = bench.Cfg(); cfg.feature_samples = S
    d = bench.make_inputs(B, C, H, W, K, S, n_neg, 1000, dev)
    loss_fn = ContrastiveCorrelationLoss(cfg)
    c, cp = d["code"].detach().clone().requires_grad_(True), d["code_pos"].detach().clone().requires_grad_(True)
    def step():
        c.grad = None; cp.grad = None
        (pil, _, pel, _, nl, _) = loss_fn(d["feats"], d["feats_pos"], None, None, c, cp)
        (0.67 * pil + 0.25 * pel + 0.63 * nl.mean()).backward()
    for _ in range(5): step()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20): step()
    t1.record(); torch.cuda.synchronize()
    print(json.dumps({"S": S, "ms_per_step": t0.elapsed_time(t1) / 20, "fused": bool(ContrastiveCorrelationLoss.fused_kernels_cover(B, C, K, H, W, S))}), flush=True)
