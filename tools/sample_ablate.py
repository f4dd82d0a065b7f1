import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from stego_amd import capi
dev = torch.device("cuda:0")
cfg = bench.Cfg()
C, H, W, K = bench.WORKLOADS["vits8_224"]
B, S, n_neg = 32, 11, 5
sets = [bench.make_inputs(B, C, H, W, K, S, n_neg, 1000 + i, dev) for i in range(4)]
for prec in (capi.PREC_F32, capi.PREC_BF16X3):
    for dbg in (0, 1, 2, 3, 4, 7):
        capi.debug_set("STEGO_DEBUG_SAMPLE", int(dbg))
        desc = capi.make_desc(B, C, K, H, W, S, n_neg, cfg, (.18, .12, .46), prec)
        tot = 0.0; n = 0
        for r in range(4):
            for d in sets:
                k0, m, f = capi.corr_fwd_profile(desc, d["feats"], d["feats_pos"], d["code"], d["code_pos"], d["coords1"], d["coords2"], d["perms"], True, 1)
                if r > 0: tot += k0; n += 1
        print("prec", prec, "debug(1=no stores,2=no feat loads,4=no code)", dbg, "sample_us %.2f" % (tot / n * 1e3), flush=True)
capi.debug_set("STEGO_DEBUG_SAMPLE", 0)
