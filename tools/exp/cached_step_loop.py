"""The cached-token training step (tools/bench_step.py::step_cached) alone, N times: for rocprofv3 --kernel-trace --stats (GPU time per
step = sum of kernel time / N) and for a host-side timing without device work (enqueue rate)."""
import os, sys, time, torch, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
warnings.simplefilter("ignore")
import bench
from stego_amd import featurizers, modules
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
B = 32
class Cfg(bench.Cfg):
    pass
cfg = Cfg()
for k, v in dict(model_type="vit_small", dino_patch_size=8, dino_feat_type="feat", projection_type="nonlinear", dropout=True,
                 pretrained_weights=None, native_backbone=True, native_head=True).items():
    setattr(cfg, k, v)
torch.manual_seed(0)
net = featurizers.DinoFeaturizer(70, cfg).to(dev)
loss_fn = modules.ContrastiveCorrelationLoss(cfg)
both = torch.randn(2 * B, 3, 224, 224, device=dev)
params = [p for p in net.parameters() if p.requires_grad]
cache = net.enable_token_cache(2 * B, (224, 224), dev)
idx = torch.arange(2 * B, device=dev)
def step_cached():
    feats_all, code_all = net(both, cache_index=idx)
    out = loss_fn(feats_all[:B], feats_all[B:], None, None, code_all[:B], code_all[B:])
    loss = .67 * out[0] + .25 * out[2] + .63 * out[4].mean()
    for p in params: p.grad = None
    loss.backward()
for _ in range(5): step_cached()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N): step_cached()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue ms/step %.3f   wall ms/step %.3f" % ((t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
