"""KNN precompute on the MI355X: the role of the reference's ``src/precompute_knns.py``.

  get_feats(model, loader)                      reference :15-21 (mean-pooled, L2-normalised features)
  compute_nearest_neighbors(normed_feats, k)    reference :86-96 (16 row blocks of einsum + topk) -> int64 [N, k];
                                                here ONE fused call, the [N, N] matrix is never materialised
  sharded_nearest_neighbors(local_feats, k)     SURVEY.md 8(e): rows sharded over the ranks of one node, one RCCL
                                                all-gather of the feature matrix, every rank answers its own rows
  save_nns(path, nns)                           the reference's file format: np.savez_compressed(..., nns=...)

The kernels live behind the C ABI (include/stego_corr.h: stego_knn_topk); there is no CPU path.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import capi

_backend = capi            # tests swap in an oracle-backed double (the product never does)


def get_feats(model, loader, device=None):
    """precompute_knns.py:15-21.  Stays on the device (the reference bounces every batch to the CPU)."""
    all_feats = []
    device = device or next(model.parameters()).device
    with torch.no_grad():
        for pack in loader:
            img = pack["img"]
            feats = F.normalize(model.forward(img.to(device)).mean([2, 3]), dim=1)
            all_feats.append(feats)
    return torch.cat(all_feats, dim=0).contiguous()


def compute_nearest_neighbors(normed_feats, k=30):
    """int64 [N, k], neighbours by descending cosine similarity (rank 0 is the row itself; data.py:524 skips it)."""
    return _backend.knn_topk(normed_feats.float(), k=k)


def shard_rows(n, world, rank):
    """Contiguous row shards whose starts are multiples of 128 (the kernel's query-block size)."""
    blocks = (n + 127) // 128
    per = (blocks + world - 1) // world
    b0 = min(blocks, rank * per)
    b1 = min(blocks, b0 + per)
    return min(n, b0 * 128), min(n, b1 * 128)


def sharded_nearest_neighbors(local_feats, k=30, group=None):
    """Every rank holds ``local_feats`` = its shard (``shard_rows`` of the global row order) of the normalised
    feature matrix.  One all-gather (RCCL over xGMI) rebuilds X on every rank, each rank answers its own rows and
    rank 0 receives the full [N, k] table (other ranks get None)."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = local_feats.device
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    counts[rank] = local_feats.shape[0]
    dist.all_reduce(counts, group=group)
    counts = [int(c) for c in counts.tolist()]
    n, d = sum(counts), local_feats.shape[1]
    bounds = [shard_rows(n, world, r) for r in range(world)]
    if [b1 - b0 for b0, b1 in bounds] != counts:
        raise ValueError("local shards must follow shard_rows(): expected %s rows per rank, got %s" %
                         ([b1 - b0 for b0, b1 in bounds], counts))
    pad = max(counts)
    buf = torch.zeros(pad, d, dtype=torch.float32, device=dev)
    buf[:counts[rank]] = local_feats
    gathered = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(gathered, buf, group=group)
    x = torch.cat([g[:c] for g, c in zip(gathered, counts)], dim=0)
    q0, q1 = bounds[rank]
    mine = _backend.knn_topk(x, k=k, q_begin=q0, q_count=q1 - q0) if q1 > q0 else \
        torch.empty(0, k, dtype=torch.int64, device=dev)
    out = torch.full((pad, k), -1, dtype=torch.int64, device=dev)
    out[:mine.shape[0]] = mine
    parts = [torch.empty_like(out) for _ in range(world)] if rank == 0 else None
    dist.gather(out, parts, dst=0, group=group)
    if rank != 0:
        return None
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)


def nns_filename(model_type, dataset_name, image_set, crop_type, res):
    """The cache file name data.py:503-511 looks up (precompute_knns.py:66-67)."""
    return "nns_{}_{}_{}_{}_{}.npz".format(model_type, dataset_name, image_set, crop_type, res)


def load_nns(path):
    """int64 [N, k] neighbour table (data.py:511: np.load(...)["nns"])."""
    return torch.from_numpy(np.load(path)["nns"])


def save_nns(path, nns):
    """Same on-disk format as the reference (:96): a compressed npz with the single key ``nns``."""
    np.savez_compressed(path, nns=nns.cpu().numpy() if torch.is_tensor(nns) else np.asarray(nns))
