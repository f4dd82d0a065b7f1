#!/usr/bin/env python
"""Same-process A/B of STEGO_DEBUG values on the forward (HIP events around single launches, rotating inputs, alternating rounds).
usage: r6_ab_debug.py <workload> <B> <debugA> <debugB> [...]"""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from stego_amd import capi

dev = torch.device("cuda:0")
cfg = bench.Cfg()
wl, B = sys.argv[1], int(sys.argv[2])
dbgs = [int(x) for x in sys.argv[3:]]
C, H, W, K = bench.WORKLOADS[wl]
S, n_neg = 11, 5
sets = [bench.make_inputs(B, C, H, W, K, S, n_neg, 1000 + i, dev) for i in range(4)]
desc = capi.make_desc(B, C, K, H, W, S, n_neg, cfg, (.18, .12, .46), capi.PREC_F16X3)
acc = {d: [] for d in dbgs}
for r in range(14):
    for dbg in dbgs:
        capi.debug_set("STEGO_DEBUG", dbg)
        for d in sets:
            k = capi.corr_fwd_profile(desc, d["feats"], d["feats_pos"], d["code"], d["code_pos"], d["coords1"], d["coords2"], d["perms"], True, 1)
            if r > 1:
                acc[dbg].append(k[1] * 1e3)
capi.debug_set("STEGO_DEBUG", 0)
for dbg in dbgs:
    v = sorted(acc[dbg])
    print(json.dumps(dict(workload=wl, B=B, debug=dbg, forward_us_mean=round(sum(v) / len(v), 2), p10=round(v[len(v) // 10], 2), p50=round(v[len(v) // 2], 2),
                          p90=round(v[len(v) * 9 // 10], 2), n=len(v))))
