"""Single-GPU dress rehearsal of the data-parallel path (run as a subprocess by tests/test_parity_gpu.py): backend "nccl" (= RCCL) with a
process group of ONE; three training steps through LitUnsupervisedSegmenter.manual_backward / wait_gradients (flat bucket, asynchronous
ReduceOp.AVG) against the same three steps without any collective.  Prints one JSON line."""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stego_amd.train_segmentation import LitUnsupervisedSegmenter, SyntheticContrastiveDataset, load_config  # noqa: E402

OVERRIDES = ["model_type=vit_tiny", "dino_patch_size=16", "res=64", "batch_size=4", "feature_samples=5", "neg_samples=2", "dim=16",
             "dropout=False", "native_backbone=False"]


def steps(collective, n=3):
    cfg = load_config(overrides=OVERRIDES)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = LitUnsupervisedSegmenter(27, cfg).to(dev)
    if collective:
        model.setup_distributed()
        model.force_collective = True
    ds = SyntheticContrastiveDataset(8, cfg.res, 27, seed=0)
    loader = torch.utils.data.DataLoader(ds, cfg.batch_size, shuffle=False, drop_last=True)
    losses = []
    torch.manual_seed(1)
    t0 = None
    it = 0
    for epoch in range(8):
        for batch in loader:
            batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
            if it == n:                         # timed tail: a few more steps for the record
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            loss = model.training_step(batch, it)
            if it < n:
                losses.append(float(loss))
            it += 1
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / (it - n)
    head = model.net.cluster1[0].weight.detach().flatten()[:64].cpu().tolist()
    return losses, head, ms


def main():
    os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29541"))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    with_c = steps(True)
    without = steps(False)
    dist.destroy_process_group()
    print(json.dumps({"backend": "nccl", "world": 1, "loss_collective": with_c[0], "loss_plain": without[0],
                      "head_equal": with_c[1] == without[1], "ms_per_step_collective": with_c[2], "ms_per_step_plain": without[2]}))


if __name__ == "__main__":
    main()
