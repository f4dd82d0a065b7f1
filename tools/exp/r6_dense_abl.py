#!/usr/bin/env python
"""Timing ablations of dense_stream_kernel (STEGO_DEBUG bits 22..25: no MFMAs / no conversion / fragments read once / no stores); results are wrong by design."""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from stego_amd import capi
dev = torch.device("cuda:0")
B, C, H = 32, 384, 28
g = torch.Generator(device=dev).manual_seed(7)
a = torch.randn(B, H, H, C, device=dev, generator=g).permute(0, 3, 1, 2)
b = torch.randn(B, H, H, C, device=dev, generator=g).permute(0, 3, 1, 2)
def t_us(iters=30):
    capi.dense_corr(a, b, normalize=True); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): capi.dense_corr(a, b, normalize=True)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for bits in [int(x) for x in sys.argv[1:]] or [0, 1, 2, 4, 8, 3, 5, 7, 15]:
    capi.debug_set("STEGO_DEBUG", bits << 22)
    print(json.dumps({"dbg": bits, "us": round(t_us(), 1)}))
capi.debug_set("STEGO_DEBUG", 0)
