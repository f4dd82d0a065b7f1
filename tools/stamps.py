#!/usr/bin/env python
"""Per-workgroup phase stamps of corr_tile_kernel (STEGO_DEBUG bit 256, s_memrealtime, 100 MHz)."""
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
d = bench.make_inputs(B, C, H, W, K, S, n_neg, 1000, dev)
lib = capi.load()
nt = (2 + n_neg) * B
for prec in (capi.PREC_F32, capi.PREC_BF16X3):
    for dbg in [int(x) for x in (sys.argv[1:] or ["256"])]:
        capi.debug_set("STEGO_DEBUG", int(dbg))
        desc = capi.make_desc(B, C, K, H, W, S, n_neg, cfg, (.18, .12, .46), prec)
        f32 = dict(dtype=torch.float32, device=dev)
        outs = [torch.empty(3, **f32), torch.empty(B, S**4, **f32), torch.empty(B, S**4, **f32),
                torch.empty(n_neg * B, S**4, **f32), torch.empty(n_neg * B, S**4, **f32),
                torch.empty(7 * B, S**4, **f32), torch.empty(7, **f32)]
        nctx = lib.stego_corr_saved_ctx_bytes(byref(desc))
        ctx = torch.empty(nctx, dtype=torch.uint8, device=dev)
        nws = lib.stego_corr_workspace_bytes(byref(desc))
        ws = torch.zeros(nws, dtype=torch.uint8, device=dev)
        maps = [capi._map(d[k]) for k in ("feats", "feats_pos", "code", "code_pos")]
        for rep in range(3):
            rc = lib.stego_corr_fwd(byref(desc), *[byref(m) for m in maps], d["coords1"].data_ptr(), d["coords2"].data_ptr(),
                                    d["perms"].data_ptr(), *[o.data_ptr() for o in outs], ctx.data_ptr(), ws.data_ptr(), ws.numel(),
                                    torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
            torch.cuda.synchronize()
        ts = ws[nt * 16 + 1024: nt * 16 + 1024 + nt * 64].view(torch.int64).cpu().numpy().reshape(nt, 8)
        t0 = ts[:, 0].min()
        rel = (ts[:, [0, 1, 2, 4]] - t0) / 100.0
        names = ["start", "mainloop end", "parked", "end"]
        print("prec=%d debug=%d" % (prec, dbg))
        for k, n in enumerate(names):
            print("   %-13s p0/p50/p100 %.2f %.2f %.2f us" % ((n,) + tuple(np.percentile(rel[:, k], [0, 50, 100]))))
        capi.debug_set("STEGO_DEBUG", 0)
