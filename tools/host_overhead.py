#!/usr/bin/env python
"""CPU-side cost of one eager call of the loss (enqueue only, no device sync inside the timed regions)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from stego_amd import modules as M

dev = torch.device("cuda:0")
cfg = bench.Cfg()
C, H, W, K = bench.WORKLOADS["vits8_224"]
B = 32
d = bench.make_inputs(B, C, H, W, K, 11, 5, 1000, dev)
loss = M.ContrastiveCorrelationLoss(cfg)
code = d["code"].clone().requires_grad_(True); code_pos = d["code_pos"].clone().requires_grad_(True)
def t(fn, n=200):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    dt = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return dt
print("draw_coords            %7.1f us" % t(lambda: loss.draw_coords(d["feats"], None, None)))
print("5 x super_perm + stack %7.1f us" % t(lambda: torch.stack([M.super_perm(B, dev) for _ in range(5)])))
c1, c2 = loss.draw_coords(d["feats"], None, None); perms = torch.stack([M.super_perm(B, dev) for _ in range(5)])
with torch.no_grad():
    print("forward_explicit nograd%7.1f us" % t(lambda: loss.forward_explicit(d["feats"], d["feats_pos"], d["code"], d["code_pos"], c1, c2, perms)))
print("forward_explicit grad  %7.1f us" % t(lambda: loss.forward_explicit(d["feats"], d["feats_pos"], code, code_pos, c1, c2, perms)))
def fb():
    out = loss.forward_explicit(d["feats"], d["feats_pos"], code, code_pos, c1, c2, perms)
    (.67 * out[0] + .25 * out[2] + .63 * out[4].mean()).backward()
print("fwd + combine + bwd    %7.1f us" % t(fb))
def full():
    out = loss(d["feats"], d["feats_pos"], None, None, code, code_pos)
    (.67 * out[0] + .25 * out[2] + .63 * out[4].mean()).backward()
print("full forward()+bwd     %7.1f us" % t(full))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): full()
torch.cuda.synchronize()
print("full, wall incl. GPU   %7.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
