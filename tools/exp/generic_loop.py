"""The product API at cfg.feature_samples = S (argv[1], default 16) fwd + bwd, N steps (argv[2]) - for rocprofv3 --kernel-trace --stats."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from stego_amd.modules import ContrastiveCorrelationLoss
dev = torch.device("cuda:0")
C, H, W, K = bench.WORKLOADS["vits8_224"]
B, n_neg = 32, 5
S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
cfg = bench.Cfg(); cfg.feature_samples = S
d = bench.make_inputs(B, C, H, W, K, S, n_neg, 1000, dev)
loss_fn = ContrastiveCorrelationLoss(cfg)
c, cp = d["code"].detach().clone().requires_grad_(True), d["code_pos"].detach().clone().requires_grad_(True)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10):
    c.grad = None; cp.grad = None
    (pil, _, pel, _, nl, _) = loss_fn(d["feats"], d["feats_pos"], None, None, c, cp)
    (0.67 * pil + 0.25 * pel + 0.63 * nl.mean()).backward()
torch.cuda.synchronize()
