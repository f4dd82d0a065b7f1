#!/usr/bin/env python
"""Round-2 experiment A: per-kernel forward times of the existing kernels with STEGO_DEBUG knobs (HIP events in the C ABI)."""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from stego_amd import capi

def main():
    dev = torch.device("cuda:0")
    cfg = bench.Cfg()
    C, H, W, K = bench.WORKLOADS["vits8_224"]
    B, S, n_neg = 32, 11, 5
    sets = [bench.make_inputs(B, C, H, W, K, S, n_neg, 1000 + i, dev) for i in range(4)]
    desc = capi.make_desc(B, C, K, H, W, S, n_neg, cfg, (.18, .12, .46), capi.PREC_F16X3)
    ref = None
    for debug in [int(x) for x in sys.argv[1:]] or [0, 16, 0, 16]:
        capi.debug_set("STEGO_DEBUG", debug)
        ts = [0.0, 0.0, 0.0]; n = 0
        for r in range(6):
            for d in sets:
                k = capi.corr_fwd_profile(desc, d["feats"], d["feats_pos"], d["code"], d["code_pos"], d["coords1"], d["coords2"], d["perms"], True, 1)
                if r > 0:
                    for i in range(3): ts[i] += k[i]
                    n += 1
        d = sets[0]
        out = capi.corr_fwd(desc, d["feats"], d["feats_pos"], d["code"], d["code_pos"], d["coords1"], d["coords2"], d["perms"], True)
        torch.cuda.synchronize()
        if ref is None: ref = out
        diff = max(float((a - b).abs().max()) for a, b in zip(out[:5], ref[:5]))
        print(json.dumps(dict(debug=debug, sample_us=round(ts[0] / n * 1e3, 2), tile_us=round(ts[1] / n * 1e3, 2), finalize_us=round(ts[2] / n * 1e3, 2), maxdiff_vs_first=diff)), flush=True)

main()
