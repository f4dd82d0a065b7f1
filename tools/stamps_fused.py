#!/usr/bin/env python
"""Per-workgroup phase stamps of corr_fused_kernel (STEGO_DEBUG bit 256, s_memrealtime, 100 MHz ticks)."""
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
B, S, n_neg = int(os.environ.get("B", 32)), 11, 5
sets = [bench.make_inputs(B, C, H, W, K, S, n_neg, 1000 + i, dev) for i in range(int(os.environ.get("SETS", 4)))]
lib = capi.load()
nt = (2 + n_neg) * B
names = ["start", "phase1 done", "anchor ready", "main loop end (E0)", "parked+rowmean+om (E2)", "end", "tile assigned", "own codes done",
         "p1: taps done", "p1: gathers landed", "p1: stores issued", "gather head done", "epi: sum fd done", "epi: row sums done",
         "epi: fd parked", "gather head start"]
for prec in (capi.PREC_F16X3,):
    for dbg in [int(x) for x in (sys.argv[1:] or ["256"])]:
        capi.debug_set("STEGO_DEBUG", dbg)
        desc = capi.make_desc(B, C, K, H, W, S, n_neg, cfg, (.18, .12, .46), prec)
        f32 = dict(dtype=torch.float32, device=dev)
        outs = [torch.empty(3, **f32), torch.empty(B, S**4, **f32), torch.empty(B, S**4, **f32),
                torch.empty(n_neg * B, S**4, **f32), torch.empty(n_neg * B, S**4, **f32),
                torch.empty(7 * B, S**4, **f32), torch.empty(7, **f32)]
        nctx = lib.stego_corr_saved_ctx_bytes(byref(desc))
        ctx = torch.empty(nctx, dtype=torch.uint8, device=dev)
        nws = lib.stego_corr_workspace_bytes(byref(desc))
        ws = torch.zeros(nws, dtype=torch.uint8, device=dev)
        for rep in range(6):
            d = sets[rep % len(sets)]
            maps = [capi._map(d[k]) for k in ("feats", "feats_pos", "code", "code_pos")]
            rc = lib.stego_corr_fwd(byref(desc), *[byref(m) for m in maps], d["coords1"].data_ptr(), d["coords2"].data_ptr(),
                                    d["perms"].data_ptr(), *[o.data_ptr() for o in outs], ctx.data_ptr(), ws.data_ptr(), ws.numel(),
                                    torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
            torch.cuda.synchronize()
        ts = ws[nt * 16 + 1024: nt * 16 + 1024 + nt * 128].view(torch.int64).cpu().numpy().reshape(nt, 16)
        t0 = ts[:, 0].min()
        rel = (ts[:, :16] - t0) / 100.0
        print("prec=%d debug=%d  (us since the first workgroup started; p0 / p50 / p100 over %d workgroups)" % (prec, dbg, nt))
        for k in (0, 8, 9, 10, 1, 6, 15, 11, 7, 2, 3, 12, 13, 14, 4, 5):
            print("   %-26s %7.2f %7.2f %7.2f" % ((names[k],) + tuple(np.percentile(rel[:, k], [0, 50, 100]))))
        loop = rel[:, 3] - rel[:, 2]
        print("   ring loop (anchor ready -> E0) percentiles 0/10/25/50/75/90/100:", np.round(np.percentile(loop, [0, 10, 25, 50, 75, 90, 100]), 1))
        print("   ring loop mean by XCD (workgroup %% 8):", [round(float(loop[x::8].mean()), 1) for x in range(8)])
        print("   ring loop mean by slot (workgroup // 8) quartiles:", [round(float(loop[8 * a: 8 * a + 56].mean()), 1) for a in (0, 7, 14, 21)])
        slot = np.arange(nt) >> 3
        for cname, sel in (("intra slots 0-3 (no gather)", slot < 4), ("gathered slots 4-27", slot >= 4)):
            if sel.any():
                print("   %-28s" % cname + "  ".join("%s %.2f/%.2f" % (n, *np.percentile(rel[sel, k], [50, 100])) for n, k in
                      (("p1 landed", 9), ("p1 stores", 10), ("p1 done", 1), ("anchor", 2), ("E0", 3), ("end", 5))))
        if dbg & 1024:      # diagnostic: the gathered tiles' phase 1 in detail (stamps 12-14 are phase-1 stamps in this mode)
            sel = slot >= 4
            print("   gathered slots, phase 1: " + "  ".join("%s %.2f/%.2f" % (n, *np.percentile(rel[sel, k], [50, 100])) for n, k in
                  (("taps", 8), ("landed", 9), ("computed", 12), ("team barrier", 13), ("stores issued", 10), ("stores landed", 14), ("done", 1), ("tile assigned", 6), ("head start", 15))))
        last = int(np.argmax(rel[:, 5]))
        print("   last workgroup %d: end %.2f | tail start %.2f  granules in %.2f  sums done %.2f  tail end %.2f" %
              ((last, rel[last, 5]) + tuple(rel[last, 12:16])))
        if os.environ.get("DUMP"):
            np.save(os.path.join(ROOT, "gpurun_out", "stamps_fused.npy"), rel)
        capi.debug_set("STEGO_DEBUG", 0)
