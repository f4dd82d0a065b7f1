#!/usr/bin/env python
"""Per-stage timeline of the fused kernel's ring loop (STEGO_DEBUG bit 512): for a few workgroups, when wave 0 (MFMA team) and
wave 4 (gather team) reach each point of every stage (us since the workgroup's first ring barrier)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from stego_amd import capi
from ctypes import byref
dev = torch.device("cuda:0")
cfg = bench.Cfg()
C, H, W, K = bench.WORKLOADS["vits8_224"]
B, S, n_neg = 32, 11, 5
sets = [bench.make_inputs(B, C, H, W, K, S, n_neg, 1000 + i, dev) for i in range(4)]
lib = capi.load()
nt = (2 + n_neg) * B
capi.debug_set("STEGO_DEBUG", 512 + int(sys.argv[1]) if len(sys.argv) > 1 else 512)
desc = capi.make_desc(B, C, K, H, W, S, n_neg, cfg, (.18, .12, .46), capi.PREC_F16X3)
f32 = dict(dtype=torch.float32, device=dev)
outs = [torch.empty(3, **f32), torch.empty(B, S**4, **f32), torch.empty(B, S**4, **f32), torch.empty(n_neg * B, S**4, **f32),
        torch.empty(n_neg * B, S**4, **f32), torch.empty(7 * B, S**4, **f32), torch.empty(7, **f32)]
ctx = torch.empty(lib.stego_corr_saved_ctx_bytes(byref(desc)), dtype=torch.uint8, device=dev)
ws = torch.zeros(lib.stego_corr_workspace_bytes(byref(desc)), dtype=torch.uint8, device=dev)
for rep in range(6):
    d = sets[rep % 4]
    maps = [capi._map(d[k]) for k in ("feats", "feats_pos", "code", "code_pos")]
    rc = lib.stego_corr_fwd(byref(desc), *[byref(m) for m in maps], d["coords1"].data_ptr(), d["coords2"].data_ptr(), d["perms"].data_ptr(),
                            *[o.data_ptr() for o in outs], ctx.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
base = nt * 16 + 1024 + nt * 128
tl = ws[base: base + nt * 1024].view(torch.int64).cpu().numpy().reshape(nt, 2, 16, 4)
for wg in (40, 100, 200):
    t0 = tl[wg, 0, 0, 0]
    print("workgroup %d (us since its MFMA wave reached stage 0)" % wg)
    print("  stage | MFMA: wait-start  landed  after-barrier  mfma-done | gather: at-barrier  after-barrier  committed  issued")
    for n in range(15):
        m = (tl[wg, 0, n] - t0) / 100.0
        g = (tl[wg, 1, n] - t0) / 100.0
        print("  %5d | %10.2f %8.2f %10.2f %10.2f | %10.2f %10.2f %10.2f %8.2f" % ((n,) + tuple(m) + tuple(g)))
