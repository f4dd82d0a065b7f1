#!/usr/bin/env python
"""Phase stamps of dense_stream_kernel (STEGO_DEBUG bit 26: every workgroup writes its stamps over the first floats of its first output row)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from stego_amd import capi
dev = torch.device("cuda:0")
B, C, H = int(os.environ.get("B", 32)), int(os.environ.get("C", 384)), int(os.environ.get("H", 28))
g = torch.Generator(device=dev).manual_seed(7)
a = torch.randn(B, H, H, C, device=dev, generator=g).permute(0, 3, 1, 2)
b = torch.randn(B, H, H, C, device=dev, generator=g).permute(0, 3, 1, 2)
capi.debug_set("STEGO_DEBUG", (16 | int(os.environ.get("DBG", 0))) << 22)
for _ in range(3):
    out = capi.dense_corr(a, b, normalize=True)
torch.cuda.synchronize()
capi.debug_set("STEGO_DEBUG", 0)
M = H * H
o = out.reshape(B, M, M).view(torch.int32).cpu().numpy().astype(np.int64) & 0xffffffff
nb = (M + 127) // 128
names = ["A block in registers", "first block parked", "second block parked", "full blocks done", "tail block done", "last stores acknowledged"]
st = np.array([[o[n, 128 * mi, :7] for mi in range(nb)] for n in range(B)]).reshape(-1, 7)
t0 = st[:, 0]
t0 = (t0 - t0.min()) & 0xffffffff
print("B=%d C=%d %dx%d: %d workgroups, 100 MHz ticks -> us; p0 / p50 / p100" % (B, C, H, H, len(st)))
print("  %-28s %7.2f %7.2f %7.2f" % ("start", *(np.percentile(t0, [0, 50, 100]) / 100)))
for k in range(1, 7):
    v = (t0 + st[:, k]) / 100
    print("  %-28s %7.2f %7.2f %7.2f   (since own start: %6.2f %6.2f %6.2f)" % (names[k - 1], *np.percentile(v, [0, 50, 100]), *(np.percentile(st[:, k], [0, 50, 100]) / 100)))
