#!/usr/bin/env python
"""Where the HOST time of the eager product path goes (bench.py product_path: the loss call with its draws, the weighted sum, .backward()):
wall time per section with the device kept busy (no synchronisation inside the loop), then cProfile's top entries."""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
if os.environ.get("PREV"):      # same-box A/B: an earlier stego_amd/modules.py (a file path) in place of the installed one
    import importlib.util
    import stego_amd
    spec = importlib.util.spec_from_file_location("stego_amd.modules", os.environ["PREV"])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["stego_amd.modules"] = mod
    spec.loader.exec_module(mod)
    stego_amd.modules = mod
from stego_amd.modules import ContrastiveCorrelationLoss

dev = torch.device("cuda:0")
cfg = bench.Cfg()
C, H, W, K = bench.WORKLOADS["vits8_224"]
B, S, n_neg = 32, 11, 5
sets = [bench.make_inputs(B, C, H, W, K, S, n_neg, 1000 + i, dev) for i in range(4)]
loss_fn = ContrastiveCorrelationLoss(cfg)
codes = [(d["code"].detach().clone().requires_grad_(True), d["code_pos"].detach().clone().requires_grad_(True)) for d in sets]
N = int(os.environ.get("N", 2000))
sec = {}


def tick(name, t0):
    t = time.perf_counter()
    sec[name] = sec.get(name, 0.0) + t - t0
    return t


def step(i, timed=False):
    d = sets[i]
    c, cp = codes[i]
    t = time.perf_counter()
    c.grad = None
    cp.grad = None
    if timed: t = tick("grads = None", t)
    coords1, coords2, perms = loss_fn.draw(d["feats"], None, None)
    if timed: t = tick("draw", t)
    (pil, _, pel, _, nl, _) = loss_fn.forward_explicit(d["feats"], d["feats_pos"], c, cp, coords1, coords2, perms)
    if timed: t = tick("forward_explicit", t)
    loss = cfg.pos_intra_weight * pil + cfg.pos_inter_weight * pel + cfg.neg_inter_weight * nl.mean()
    if timed: t = tick("weighted sum", t)
    loss.backward()
    if timed: t = tick("backward()", t)


def step_total(i, timed=False):
    d = sets[i]
    c, cp = codes[i]
    t = time.perf_counter()
    c.grad = None
    cp.grad = None
    if timed: t = tick("T grads = None", t)
    out = loss_fn.total(d["feats"], d["feats_pos"], None, None, c, cp, (cfg.pos_intra_weight, cfg.pos_inter_weight, cfg.neg_inter_weight))
    if timed: t = tick("T total()", t)
    out[0].backward()
    if timed: t = tick("T backward()", t)


for fn, label in ((step, "forward() + weighted sum"), (step_total, "total()")):
    for k in range(60):
        fn(k % 4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(N):
        fn(k % 4)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("%s: host enqueue %.1f us / step, incl. final drain %.1f us / step" % (label, 1e6 * t_host / N, 1e6 * t_all / N))
    sec.clear()
    for k in range(N):
        fn(k % 4, True)
    torch.cuda.synchronize()
    for name, v in sec.items():
        print("    %-22s %7.1f us" % (name, 1e6 * v / N))
    pr = cProfile.Profile()
    pr.enable()
    for k in range(N):
        fn(k % 4)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr, stream=sys.stdout)
    st.sort_stats("cumulative").print_stats(28)
    st.sort_stats("tottime").print_stats(18)
