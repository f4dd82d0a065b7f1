#!/usr/bin/env python
"""Dump the in-kernel s_memtime stamps of block 0 (STEGO_DEBUG bit 256) for a few configurations."""
import os, sys
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
for prec in (capi.PREC_F32, capi.PREC_BF16X3):
    for dbg in (256, 256 + 64):
        os.environ["STEGO_DEBUG"] = str(dbg)
        desc = capi.make_desc(B, C, K, H, W, S, n_neg, cfg, (.18, .12, .46), prec)
        f32 = dict(dtype=torch.float32, device=dev)
        outs = [torch.empty(2, **f32), torch.empty(B, S**4, **f32), torch.empty(B, S**4, **f32),
                torch.empty(n_neg * B, S**4, **f32), torch.empty(n_neg * B, S**4, **f32),
                torch.empty(7 * B, S**4, **f32), torch.empty(7, **f32)]
        nws = lib.stego_corr_workspace_bytes(byref(desc))
        ws = torch.zeros(nws, dtype=torch.uint8, device=dev)
        maps = [capi._map(d[k]) for k in ("feats", "feats_pos", "code", "code_pos")]
        for rep in range(3):
            ws.zero_()
            rc = lib.stego_corr_fwd(byref(desc), *[byref(m) for m in maps], d["coords1"].data_ptr(), d["coords2"].data_ptr(),
                                    d["perms"].data_ptr(), *[o.data_ptr() for o in outs], ws.data_ptr(), ws.numel(),
                                    torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
            torch.cuda.synchronize()
        ts = ws[7 * B * 16: 7 * B * 16 + 512].view(torch.int64).cpu().tolist()
        cons, prod = ts[:32], ts[32:64]
        t0 = min(x for x in cons + prod if x > 0)
        def rel(a):
            return [(x - t0) if x > 0 else None for x in a]
        print("prec=%d debug=%d (cycles since first stamp; ~2.1 cycles/ns)" % (prec, dbg))
        c, p = rel(cons), rel(prod)
        print("  consumer: entry %s taps %s | per-iter (work_done, barrier_passed): %s | parked %s after-bar %s epilogue-end %s" %
              (c[0], c[1], [(c[2 + 2 * i], c[3 + 2 * i]) for i in range(9)], c[24], c[25], c[26]))
        print("  producer: entry %s taps %s | per-iter: %s" % (p[0], p[1], [(p[2 + 2 * i], p[3 + 2 * i]) for i in range(9)]))
os.environ["STEGO_DEBUG"] = "0"
