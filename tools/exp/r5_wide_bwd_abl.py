"""Ablations of the S = 16 loss backward (csrc/corr_wide.hip): whole stego_corr_bwd time with parts of wide_bwd_kernel removed (STEGO_DEBUG_BWD bits 16..18)."""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from stego_amd import capi
from stego_amd.modules import ContrastiveCorrelationLoss
dev = torch.device("cuda:0")
C, H, W, K = bench.WORKLOADS["vits8_224"]
B, n_neg = 32, 5
S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
cfg = bench.Cfg(); cfg.feature_samples = S
d = bench.make_inputs(B, C, H, W, K, S, n_neg, 1000, dev)
loss_fn = ContrastiveCorrelationLoss(cfg)
c, cp = d["code"].detach().clone().requires_grad_(True), d["code_pos"].detach().clone().requires_grad_(True)
# (the switches of wide_bwd_kernel itself - no matrix products / no w loads / no code-tile copies - were removed again after the measurement:
# profiles/r05g_bwd_abl.txt; they cost the kernel 17 spilled registers)
for dbg, name in ((0, "full"), (8, "scatter: plain stores instead of atomics")):
    capi.debug_set("STEGO_DEBUG_BWD", dbg << 16)
    ts = []
    for it in range(12):
        c.grad = None; cp.grad = None
        (pil, _, pel, _, nl, _) = loss_fn(d["feats"], d["feats_pos"], None, None, c, cp)
        tot = 0.67 * pil + 0.25 * pel + 0.63 * nl.mean()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); tot.backward(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(json.dumps({"S": S, "variant": name, "backward_us_min": round(min(ts[2:]), 1), "median": round(sorted(ts[2:])[5], 1)}), flush=True)
capi.debug_set("STEGO_DEBUG_BWD", 0)
