#!/usr/bin/env python
"""Per-workgroup phase stamps of corr_fused_half_kernel (the column-half launch of small batches; STEGO_DEBUG bit 256, 100 MHz ticks).
env: WL (vits8_224), B (16)."""
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
wl = os.environ.get("WL", "vits8_224")
C, H, W, K = bench.WORKLOADS[wl]
B, S, n_neg = int(os.environ.get("B", 16)), 11, 5
sets = [bench.make_inputs(B, C, H, W, K, S, n_neg, 1000 + i, dev) for i in range(4)]
lib = capi.load()
nt = (2 + n_neg) * B
cus = torch.cuda.get_device_properties(0).multi_processor_count & ~7
na = (cus - 2 * nt) & ~7
grid = na + 2 * nt
names = {12: "cd swept", 13: "w swept", 14: "old_mean known", 15: "loss swept", 0: "start", 1: "phase 1 done (samplers) / skipped", 6: "item known", 11: "gather head done", 2: "anchor ready", 7: "loop end", 3: "E0",
         4: "row sums + sum fd published", 8: "row means (partner read)", 5: "sums published + ticket"}
DBG = 256 | int(os.environ.get("DBG", 0))
capi.debug_set("STEGO_DEBUG", DBG)
print("STEGO_DEBUG", DBG)
desc = capi.make_desc(B, C, K, H, W, S, n_neg, cfg, (.18, .12, .46), capi.PREC_F16X3)
f32 = dict(dtype=torch.float32, device=dev)
outs = [torch.empty(3, **f32), torch.empty(B, S**4, **f32), torch.empty(B, S**4, **f32), torch.empty(n_neg * B, S**4, **f32),
        torch.empty(n_neg * B, S**4, **f32), torch.empty(7 * B, S**4, **f32), torch.empty(7, **f32)]
ctx = torch.empty(lib.stego_corr_saved_ctx_bytes(byref(desc)), dtype=torch.uint8, device=dev)
ws = torch.zeros(lib.stego_corr_workspace_bytes(byref(desc)), dtype=torch.uint8, device=dev)
for rep in range(6):
    d = sets[rep % len(sets)]
    maps = [capi._map(d[k]) for k in ("feats", "feats_pos", "code", "code_pos")]
    rc = lib.stego_corr_fwd(byref(desc), *[byref(m) for m in maps], d["coords1"].data_ptr(), d["coords2"].data_ptr(), d["perms"].data_ptr(),
                            *[o.data_ptr() for o in outs], ctx.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    torch.cuda.synchronize()
ts = ws[nt * 16 + 1024: nt * 16 + 1024 + grid * 128].view(torch.int64).cpu().numpy().reshape(grid, 16).copy()
t0 = ts[:, 0].min()
rel = (ts - t0) / 100.0
anchor = np.arange(grid) < na
slot = (np.arange(grid) - na) >> 3
nb = (B + 7) // 8
intra = (~anchor) & (slot < 2 * nb)
heavy = (~anchor) & ~intra
print("B=%d %s: grid %d = %d anchor workgroups + %d items (us since the first workgroup started; p0 / p50 / p100)" % (B, wl, grid, na, 2 * nt))
for cname, sel in (("anchor workgroups", anchor), ("self-correlation halves (samplers)", intra), ("gathered items", heavy)):
    print(" %s (%d)" % (cname, int(sel.sum())))
    for k in (0, 1, 6, 11, 2, 7, 3, 4, 12, 8, 13, 14, 15, 5):
        if cname.startswith("anchor") and k not in (0, 1):
            continue
        if cname.startswith("self") and k in (14, 15):
            continue
        if cname.startswith("self") and k == 11:
            continue
        v = rel[sel, k]
        print("   %-38s %7.2f %7.2f %7.2f" % ((names[k],) + tuple(np.percentile(v, [0, 50, 100]))))
last = int(np.argmax(ts[:, 10] * (ts[:, 10] > ts[:, 0])))
print(" last workgroup %d: tail start %.2f, tail end %.2f (us since the first workgroup started)" % (last, rel[last, 9], rel[last, 10]))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
tt = []
for rep in range(12):
    d = sets[rep % len(sets)]
    maps = [capi._map(d[k]) for k in ("feats", "feats_pos", "code", "code_pos")]
    e0.record()
    lib.stego_corr_fwd_prepared(byref(desc), *[byref(m) for m in maps], d["coords1"].data_ptr(), d["coords2"].data_ptr(), d["perms"].data_ptr(),
                                *[o.data_ptr() for o in outs], ctx.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
    e1.record()
    torch.cuda.synchronize()
    tt.append(e0.elapsed_time(e1) * 1e3)
print(" the same launch by torch events around stego_corr_fwd_prepared: %s us" % np.round(sorted(tt)[2:-2], 1))
capi.debug_set("STEGO_DEBUG", 0)
