"""Wall time per training step of stego_amd.train_segmentation (synthetic data, token cache on): the trainer a user runs, not a kernel loop."""
import os, sys, time, torch, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
warnings.simplefilter("ignore")
from stego_amd.train_segmentation import LitUnsupervisedSegmenter, SyntheticContrastiveDataset, Trainer, load_config
ov = ["batch_size=32", "cache_backbone_tokens=True", "native_backbone=True"] + sys.argv[1:]
cfg = load_config(overrides=ov)
torch.manual_seed(0)
model = LitUnsupervisedSegmenter(27, cfg)
ds = SyntheticContrastiveDataset(256, cfg.res, 27)
loader = torch.utils.data.DataLoader(ds, batch_size=cfg.batch_size, shuffle=False, drop_last=True)
tr = Trainer(max_steps=8, log_every=1000)
tr.fit(model, loader)                      # builds optimizers / the token cache and fills it for the 8 batches below
dev = tr.device
batches = []
for b in loader:
    batches.append({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()})
    if len(batches) == 8:
        break
for i in range(16):
    model.training_step(batches[i % 8], i)
torch.cuda.synchronize()
N = 64
t0 = time.perf_counter()
for i in range(N):
    model.training_step(batches[i % 8], 100 + i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("training_step (cached tokens, B = %d pairs, batches resident): enqueue %.3f ms, wall %.3f ms per step" %
      (cfg.batch_size, (t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
if os.environ.get("PROFILE"):
    import cProfile, pstats, io
    pr = cProfile.Profile(); pr.enable()
    for i in range(32):
        model.training_step(batches[i % 8], 200 + i)
    torch.cuda.synchronize(); pr.disable()
    st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(45); print(st.getvalue()[:9000])
