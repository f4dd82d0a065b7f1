#!/usr/bin/env python
"""Round 6: the column-half launch (default for 16 B <= CUs) against the full-tile launch of the same library (STEGO_DEBUG bit 16384
keeps the full-tile kernel): every output compared, forward times by HIP events around single launches on rotating inputs.
usage: r6_half_check.py [workload] [B ...]"""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from stego_amd import capi


def run(desc, d, need_grad=True):
    out = capi.corr_fwd(desc, d["feats"], d["feats_pos"], d["code"], d["code_pos"], d["coords1"], d["coords2"], d["perms"], need_grad)
    torch.cuda.synchronize()
    return out


def main():
    dev = torch.device("cuda:0")
    cfg = bench.Cfg()
    wl = sys.argv[1] if len(sys.argv) > 1 else "vits8_224"
    Bs = [int(x) for x in sys.argv[2:]] or [16]
    C, H, W, K = bench.WORKLOADS[wl]
    S, n_neg = 11, 5
    for B in Bs:
        sets = [bench.make_inputs(B, C, H, W, K, S, n_neg, 1000 + i, dev) for i in range(4)]
        for prec, pname in ((capi.PREC_F16X3, "f16x3"), (capi.PREC_F32, "f32")):
            desc = capi.make_desc(B, C, K, H, W, S, n_neg, cfg, (.18, .12, .46), prec)
            capi.debug_set("STEGO_DEBUG", 16384)
            ref = run(desc, sets[0])
            capi.debug_set("STEGO_DEBUG", 0)
            for rep in range(2):
                out = run(desc, sets[0])
                rel = []
                for a, b in zip(out[:5], ref[:5]):
                    rel.append(float((a - b).abs().max() / (b.abs().mean() + 1e-30)))
                sw = float((out[5][0] - ref[5][0]).abs().max() / ref[5][0].abs().mean())
                sm = float((out[5][1] - ref[5][1]).abs().max())
                na = out[5][2].numel() // 4
                ctx = float((out[5][2].view(torch.float32)[:na] - ref[5][2].view(torch.float32)[:na]).abs().nan_to_num(0).max())
                print(json.dumps(dict(B=B, prec=pname, rep=rep, maxdiff_over_mean=[round(x, 9) for x in rel], saved_w=sw, saved_mean=sm, ctx=ctx,
                                      nan=bool(any(torch.isnan(o).any() for o in out[:5])), means=[float(x) for x in out[0]],
                                      events=capi.event_counters_total() if hasattr(capi, "event_counters_total") else None)), flush=True)
            for debug, name in ((16384, "full tiles"), (0, "column halves")):
                capi.debug_set("STEGO_DEBUG", debug)
                ts, n = 0.0, 0
                for r in range(8):
                    for d in sets:
                        k = capi.corr_fwd_profile(desc, d["feats"], d["feats_pos"], d["code"], d["code_pos"], d["coords1"], d["coords2"], d["perms"], True, 1)
                        if r > 1:
                            ts += k[1]
                            n += 1
                print(json.dumps(dict(B=B, prec=pname, launch=name, forward_us=round(ts / n * 1e3, 2))), flush=True)
            capi.debug_set("STEGO_DEBUG", 0)


main()
