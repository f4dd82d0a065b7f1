#!/usr/bin/env python
"""Round 5 diagnostic: the anchor operand images (fs, csf) of the installed library's normal launch against those of the launch in which every
tile samples its anchor itself (STEGO_DEBUG 64: the reference statement of phase 1), byte for byte, by (anchor, stage, plane, row)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from stego_amd import capi
from ctypes import byref

dev = torch.device("cuda:0")
cfg = bench.Cfg()
C, H, W, K = bench.WORKLOADS["vits8_224"]
B, S, n_neg = 32, 11, 5
lib = capi.load()
nt = (2 + n_neg) * B
desc = capi.make_desc(B, C, K, H, W, S, n_neg, cfg, (.18, .12, .46), capi.PREC_F16X3)
f32 = dict(dtype=torch.float32, device=dev)
def ru(x, a=256): return (x + a - 1) // a * a
stats_bytes = ru(nt * 16 + 1024 + nt * (128 + 1024))
sync_bytes = ru(B * 256 + nt * 32) + 256
fs_bytes = max(ru(B * (C // 64) * 2 * 128 * 72 * 2 + 1024), ru(B * 12 * 16384 + 1024))
csf_bytes = ru(B * 3 * 16384 + 1024)
def run(d, dbg):
    capi.debug_set("STEGO_DEBUG", dbg)
    outs = [torch.empty(3, **f32), torch.empty(B, S**4, **f32), torch.empty(B, S**4, **f32), torch.empty(n_neg * B, S**4, **f32),
            torch.empty(n_neg * B, S**4, **f32), torch.empty(7 * B, S**4, **f32), torch.empty(7, **f32)]
    nctx = lib.stego_corr_saved_ctx_bytes(byref(desc))
    ctx = torch.zeros(nctx, dtype=torch.uint8, device=dev)
    nws = lib.stego_corr_workspace_bytes(byref(desc))
    ws = torch.zeros(nws, dtype=torch.uint8, device=dev)
    maps = [capi._map(d[k]) for k in ("feats", "feats_pos", "code", "code_pos")]
    rc = lib.stego_corr_fwd(byref(desc), *[byref(m) for m in maps], d["coords1"].data_ptr(), d["coords2"].data_ptr(), d["perms"].data_ptr(),
                            *[o.data_ptr() for o in outs], ctx.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    torch.cuda.synchronize()
    capi.debug_set("STEGO_DEBUG", 0)
    o0 = stats_bytes + sync_bytes
    fs = ws[o0: o0 + B * 12 * 16384].cpu().numpy().reshape(B, 12, 2, 128, 64)
    csf = ws[o0 + fs_bytes: o0 + fs_bytes + B * 3 * 16384].cpu().numpy().reshape(B, 3, 128, 128)
    return fs, csf, ctx.cpu().numpy(), [o.cpu().numpy() for o in outs]
for seed in (4101, 4102):
    d = bench.make_inputs(B, C, H, W, K, S, n_neg, seed, dev)
    a = run(d, 0); b = run(d, 64)
    df = (a[0] != b[0]); dc = (a[1] != b[1])
    print("seed", seed, "fs bytes differing", int(df.sum()), "csf", int(dc.sum()), "ctx", int((a[2] != b[2]).sum()),
          "outs", [int((x != y).sum()) for x, y in zip(a[3], b[3])])
    if df.any():
        idx = np.argwhere(df.any(axis=4))
        print("  fs (anchor, stage, plane, row) differing: %d;" % len(idx), "rows:", sorted(set(idx[:, 3].tolist()))[:40], "anchors:", sorted(set(idx[:, 0].tolist())),
              "stages:", sorted(set(idx[:, 1].tolist())), "planes:", sorted(set(idx[:, 2].tolist())))
        r = idx[0]
        print("  first:", r, a[0][tuple(r)][:32], b[0][tuple(r)][:32])
    if dc.any():
        idx = np.argwhere(dc.any(axis=3))
        print("  csf (anchor, chunk, row) differing: %d;" % len(idx), "rows:", sorted(set(idx[:, 2].tolist()))[:40], "anchors:", sorted(set(idx[:, 0].tolist())))
        r = idx[0]
        print("  first:", r, a[1][tuple(r)].view(np.float32)[:32], b[1][tuple(r)].view(np.float32)[:32])
