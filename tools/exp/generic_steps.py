"""Per-step wall times (host clock, synchronised) of the generic loss step at S = argv[1]: is the step time stable?"""
import os, sys, json, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from stego_amd.modules import ContrastiveCorrelationLoss
dev = torch.device("cuda:0")
C, H, W, K = bench.WORKLOADS["vits8_224"]
B, n_neg = 32, 5
for S in [int(a) for a in sys.argv[1:]] or [12, 16]:
    cfg = bench.Cfg(); cfg.feature_samples = S
    d = bench.make_inputs(B, C, H, W, K, S, n_neg, 1000, dev)
    loss_fn = ContrastiveCorrelationLoss(cfg)
    c, cp = d["code"].detach().clone().requires_grad_(True), d["code_pos"].detach().clone().requires_grad_(True)
    def step():
        c.grad = None; cp.grad = None
        (pil, _, pel, _, nl, _) = loss_fn(d["feats"], d["feats_pos"], None, None, c, cp)
        (0.67 * pil + 0.25 * pel + 0.63 * nl.mean()).backward()
    ts = []
    for i in range(40):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        step()
        t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append((round((t1 - t0) * 1e3, 3), round((t2 - t0) * 1e3, 3)))
    print(json.dumps({"S": S, "host_ms, total_ms per step": ts[:4] + ts[20:30]}), flush=True)
    print(json.dumps({"S": S, "reserved_MB": torch.cuda.memory_reserved() >> 20, "allocated_MB": torch.cuda.memory_allocated() >> 20,
                      "num_alloc_retries": torch.cuda.memory_stats().get("num_alloc_retries"), "segments": torch.cuda.memory_stats().get("segment.all.current")}), flush=True)
