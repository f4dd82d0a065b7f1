#!/usr/bin/env python
"""Round-2 experiment B: fused forward vs the three-launch forward (same inputs): equality of every output, fallback
paths (debug 32 = repair kernel applies old_mean, 64 = every tile samples its anchor itself), kernel times."""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    C, H, W, K = bench.WORKLOADS[wl]
    S, n_neg = 11, 5
    sets = [bench.make_inputs(B, C, H, W, K, S, n_neg, 1000 + i, dev) for i in range(4)]
    for prec, pname in ((capi.PREC_F16X3, "f16x3"), (capi.PREC_F32, "f32")):
        desc = capi.make_desc(B, C, K, H, W, S, n_neg, cfg, (.18, .12, .46), prec)
        capi.debug_set("STEGO_FWD_VARIANT", 1); capi.debug_set("STEGO_DEBUG", 0)
        ref = run(desc, sets[0])
        for variant, debug, name in ((2, 0, "fused"), (2, 32, "fused, repair path"), (2, 64, "fused, help path"), (2, 0, "fused again")):
            capi.debug_set("STEGO_FWD_VARIANT", variant); capi.debug_set("STEGO_DEBUG", debug)
            out = run(desc, sets[0])
            diffs = [float((a - b).abs().max()) for a, b in zip(out[:5], ref[:5])]
            sv = [float((a.float() - b.float()).abs().max()) if a.dtype.is_floating_point else -1 for a, b in zip(out[5][:2], ref[5][:2])]
            ctx = float((out[5][2].view(torch.float32)[: out[5][2].numel() // 4] - ref[5][2].view(torch.float32)[: ref[5][2].numel() // 4]).abs().nan_to_num(0).max())
            print(json.dumps(dict(prec=pname, variant=name, maxdiff=diffs, saved_w_mean=sv, ctx=ctx, nan=bool(any(torch.isnan(o).any() for o in out[:5])))), flush=True)
        for variant in (1, 2):
            capi.debug_set("STEGO_FWD_VARIANT", variant); capi.debug_set("STEGO_DEBUG", 0)
            ts = [0.0, 0.0, 0.0]; n = 0
            for r in range(6):
                for d in sets:
                    k = capi.corr_fwd_profile(desc, d["feats"], d["feats_pos"], d["code"], d["code_pos"], d["coords1"], d["coords2"], d["perms"], True, 1)
                    if r > 0:
                        for i in range(3): ts[i] += k[i]
                        n += 1
            print(json.dumps(dict(prec=pname, variant=variant, k0_us=round(ts[0] / n * 1e3, 2), k1_us=round(ts[1] / n * 1e3, 2), k2_us=round(ts[2] / n * 1e3, 2),
                                  total_us=round(sum(ts) / n * 1e3, 2))), flush=True)
    capi.debug_set("STEGO_FWD_VARIANT", -1)

main()
