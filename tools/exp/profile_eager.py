"""Host-side profile of the eager product path: ContrastiveCorrelationLoss.forward (reference draws) + weighted sum + backward."""
import cProfile, pstats, io, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from stego_amd.modules import ContrastiveCorrelationLoss
dev = torch.device("cuda:0")
cfg = bench.Cfg()
C, H, W, K = bench.WORKLOADS["vits8_224"]
B, S, n_neg = 32, 11, 5
sets = [bench.make_inputs(B, C, H, W, K, S, n_neg, 1000 + i, dev) for i in range(4)]
loss_fn = ContrastiveCorrelationLoss(cfg)
codes = [(d["code"].detach().clone().requires_grad_(True), d["code_pos"].detach().clone().requires_grad_(True)) for d in sets]
def step(i):
    d = sets[i]; c, cp = codes[i]; c.grad = None; cp.grad = None
    (pil, _, pel, _, nl, _) = loss_fn(d["feats"], d["feats_pos"], None, None, c, cp)
    (cfg.pos_intra_weight * pil + cfg.pos_inter_weight * pel + cfg.neg_inter_weight * nl.mean()).backward()
for i in range(8): step(i % 4)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    for k in range(200): step(k % 4)
    torch.cuda.synchronize()
    print("eager us/step", (time.perf_counter() - t0) / 200 * 1e6)
pr = cProfile.Profile(); pr.enable()
for k in range(200): step(k % 4)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(35); print(s.getvalue()[:6000])
