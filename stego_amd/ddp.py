"""Data-parallel gradient exchange for the STEGO training loop on one MI355X node.

The reference trains under Lightning ``accelerator='ddp'`` (train_segmentation.py:476): one process
per GPU, NCCL all-reduce of gradients inside ``manual_backward`` (:227).  The backbone is frozen
(modules.py:30-31), so the only traffic is the segmentation head + probes: ~205 k fp32 (0.82 MB) for
ViT-S, ~0.70 M (2.8 MB) for ViT-B - latency-bound, not bandwidth-bound.  Nothing inside the
correspondence loss is synchronised (every reduction there is over the rank-local batch,
SURVEY.md 8(e)), so the batch is simply sharded and ONE collective per step averages the trainable
gradients.

MI355X mapping: ``torch.distributed`` backend ``"nccl"`` is RCCL on ROCm; xGMI is a point-to-point
mesh, so instead of DDP's per-bucket hooks (several small launches, each paying ring latency over
single links) all trainable gradients live in ONE flat fp32 buffer whose views are the
``.grad`` tensors, reduced with a single ``all_reduce`` that RCCL's tree/direct algorithms handle
in one hop at this size.  With backend ``"gloo"`` the same code runs on CPU (used by the tests).
"""
import os

import torch
import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def init_from_env(backend=None):
    """Initialise the default process group from torchrun-style env vars (RANK, WORLD_SIZE,
    LOCAL_RANK, MASTER_ADDR, MASTER_PORT).  Returns (rank, world_size, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return rank, world, local_rank


class FlatGradReducer:
    """Owns one flat fp32 gradient buffer for a list of parameters; ``p.grad`` of every parameter is
    a view into it, so the backward writes straight into the bucket and ``allreduce_mean()`` is a
    single collective with no packing copies."""

    def __init__(self, params, process_group=None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatGradReducer: no trainable parameters")
        dev = self.params[0].device
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.group = process_group
        off = 0
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("FlatGradReducer expects fp32 parameters on one device")
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n

    def zero_grad(self):
        """Replaces optimizer.zero_grad(): keeps the .grad views alive (set_to_none would detach them)."""
        self.flat.zero_()

    def reattach(self):
        """Re-point .grad at the bucket if something replaced it (e.g. optimizer.zero_grad(set_to_none=True));
        a gradient accumulated into a fresh tensor is copied into its slot."""
        off = 0
        for p in self.params:
            n = p.numel()
            view = self.flat[off:off + n].view_as(p)
            if p.grad is None:
                view.zero_()
                p.grad = view
            elif p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
                p.grad = view
            off += n

    def allreduce_mean(self, async_op=False, single_rank_ok=False):
        """One all-reduce of the whole bucket, averaged over the ranks; no-op (returns None) for a single process.

        RCCL (backend 'nccl'): ``ReduceOp.AVG`` - the 1/world is folded into the collective, no extra kernel - launched
        asynchronously on RCCL's stream.  With ``async_op=True`` the work handle is returned: ``handle.wait()`` is a
        stream dependency (the host does not block), so the caller can enqueue whatever does not need the gradients
        (logging reductions, the next batch's copies) before it waits.  gloo (CPU tests) has no AVG: SUM + divide,
        synchronous."""
        if not is_distributed() and not (single_rank_ok and dist.is_available() and dist.is_initialized()):
            return None                  # (single_rank_ok: run the collective on a group of one - bench.py --force-collective)
        if dist.get_backend(self.group) == "nccl":
            # (A caller that really enqueues the next step's loss before it waits for this collective sets cfg.shared_device /
            # capi.set_shared_device so that the fused forward leaves the compute units beyond its tiles alone; the trainer waits
            # right after the backward, so it keeps the faster one-workgroup-per-CU launch.  Nothing global is flipped here.)
            work = dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
            if async_op:
                return work
            work.wait()
            return None
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.flat.div_(dist.get_world_size(self.group))
        return None

    def broadcast_params(self, src=0):
        """Startup synchronisation of the trainable parameters (DDP does the same at construction)."""
        if not is_distributed():
            return
        for p in self.params:
            dist.broadcast(p.data, src=src, group=self.group)
