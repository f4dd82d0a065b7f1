"""sample_panels_kernel at the loss's S = 16 shapes (features C = 384 + codes K = 70 in one launch, the three sources): time with and without the row stores."""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from stego_amd import capi
from stego_amd.modules import ContrastiveCorrelationLoss
dev = torch.device("cuda:0")
C, H, W, K = bench.WORKLOADS["vits8_224"]
B, n_neg, S = 32, 5, int(sys.argv[1]) if len(sys.argv) > 1 else 16
cfg = bench.Cfg(); cfg.feature_samples = S
d = bench.make_inputs(B, C, H, W, K, S, n_neg, 1000, dev)
loss_fn = ContrastiveCorrelationLoss(cfg)
for dbg, name in ((0, "full"), (1, "no row stores")):
    capi.debug_set("STEGO_DEBUG_SAMPLE", dbg << 16)
    ts = []
    for it in range(12):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        with torch.no_grad():
            loss_fn(d["feats"], d["feats_pos"], None, None, d["code"], d["code_pos"])
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(json.dumps({"S": S, "variant": name, "forward_us_min": round(min(ts[2:]), 1), "median": round(sorted(ts[2:])[5], 1)}), flush=True)
capi.debug_set("STEGO_DEBUG_SAMPLE", 0)
