#!/usr/bin/env python
"""Kernel-level timing matrix on the GPU box (measurement only, not a test):
forward tile kernel by variant x precision x ablation flags (HIP events inside the C ABI),
backward kernel by ablation flags (torch events around the C-ABI call)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from stego_amd import capi  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    cfg = bench.Cfg()
    wl = sys.argv[1] if len(sys.argv) > 1 else "vits8_224"
    C, H, W, K = bench.WORKLOADS[wl]
    B, S, n_neg = 32, 11, 5
    sets = [bench.make_inputs(B, C, H, W, K, S, n_neg, 1000 + i, dev) for i in range(4)]
    out = {"workload": wl, "fwd": [], "bwd": []}

    def fwd_time(prec, variant, debug):
        capi.debug_set("STEGO_FWD_VARIANT", int(variant))
        capi.debug_set("STEGO_DEBUG", int(debug))
        desc = capi.make_desc(B, C, K, H, W, S, n_neg, cfg, (.18, .12, .46), prec)
        ts = tm = tf = 0.0
        n = 0
        for r in range(4):
            for d in sets:
                k0, m, f = capi.corr_fwd_profile(desc, d["feats"], d["feats_pos"], d["code"], d["code_pos"], d["coords1"],
                                                 d["coords2"], d["perms"], True, 1)
                if r > 0:
                    ts += k0; tm += m; tf += f; n += 1
        return ts / n * 1e3, tm / n * 1e3, tf / n * 1e3

    for prec, pname in ((capi.PREC_F32, "f32"), (capi.PREC_BF16X3, "f16x3")):
        for variant in (0, 1):
            if variant == 0 and prec != capi.PREC_F32:
                continue
            abl = ((0, "full"), (1, "no-mfma"), (2, "no-loads"), (3, "epilogue-only"))
            if variant == 1:
                abl = abl + ((4, "full-no-stores"), (8, "no elementwise pass"), (3 + 8, "park+rowmean only"))
            for debug, dname in abl:
                try:
                    k0, m, f = fwd_time(prec, variant, debug)
                    rec = dict(prec=pname, variant=variant, ablation=dname, sample_us=round(k0, 2), main_us=round(m, 2),
                               finalize_us=round(f, 2))
                except Exception as e:  # noqa: BLE001
                    rec = dict(prec=pname, variant=variant, ablation=dname, error=str(e))
                out["fwd"].append(rec)
                print(rec, flush=True)
    capi.debug_set("STEGO_DEBUG", 0)
    capi.debug_set("STEGO_FWD_VARIANT", 1)

    # backward
    desc = capi.make_desc(B, C, K, H, W, S, n_neg, cfg, (.18, .12, .46), capi.PREC_F32)
    g_intra = torch.tensor(0.67, device=dev)
    g_inter = torch.tensor(0.25, device=dev)
    g_neg = torch.full((1,), 0.63 / (n_neg * B * S ** 4), device=dev).expand(n_neg * B, S, S, S, S)
    fw = []
    for d in sets:
        fw.append(capi.corr_fwd(desc, d["feats"], d["feats_pos"], d["code"], d["code_pos"], d["coords1"], d["coords2"],
                                d["perms"], True))
    for debug, dname in ((0, "full"), (1, "no-mfma"), (2, "no-scatter"), (4, "no-gfill-loads"), (7, "gather+norm only")):
        capi.debug_set("STEGO_DEBUG_BWD", int(debug))
        ts = []
        for r in range(4):
            for d, o in zip(sets, fw):
                lm, icd, ecd, nl, ncd, saved = o
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                capi.corr_bwd(desc, d["code"], d["code_pos"], d["coords1"], d["coords2"], d["perms"], saved, icd, ecd,
                              ncd, g_intra, g_inter, g_neg, None, None, None)
                e1.record()
                torch.cuda.synchronize()
                if r > 0:
                    ts.append(e0.elapsed_time(e1) * 1e3)
        rec = dict(ablation=dname, bwd_call_us=round(sum(ts) / len(ts), 2), note="2 memsets + kernel")
        out["bwd"].append(rec)
        print(rec, flush=True)
    capi.debug_set("STEGO_DEBUG_BWD", 0)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
