"""The loss kernels in shared-device mode (STEGO_FLAG_SHARED_DEVICE) while an RCCL kernel occupies part of the device: run as a subprocess
by tests/test_parity_gpu.py on ONE GPU.  A process group of one rank on backend "nccl": ReduceOp.AVG is the collective that launches a
device kernel at this world size (RCCL's oneRankReduce; the ring kernels need a second rank - this image's boxes have one GPU).  The
forward (one launch, workgroups that wait for each other) and the backward of BASELINE config 2 must give the SAME BITS as on a quiet
device, launched eagerly beside the collective's stream and replayed from a graph that holds the collective.  Prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from stego_amd import capi  # noqa: E402


def main():
    os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29563"))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=dev)
    B, C, H, W, K, S, n_neg = 32, 384, 28, 28, 70, 11, 5
    cfg = bench.Cfg()
    d = bench.make_inputs(B, C, H, W, K, S, n_neg, 777, dev)
    desc = capi.make_desc(B, C, K, H, W, S, n_neg, cfg, (.18, .12, .46), capi.PREC_F16X3, shared_device=True)
    gi, ge = torch.tensor(0.67, device=dev), torch.tensor(0.25, device=dev)
    gn = torch.full((1,), 0.63, device=dev)
    bucket = torch.ones(24 << 20, device=dev)          # 96 MB: the collective's kernel outlasts the step beside it

    def step():
        o = capi.corr_fwd(desc, d["feats"], d["feats_pos"], d["code"], d["code_pos"], d["coords1"], d["coords2"], d["perms"], True)
        dc, dcp = capi.corr_bwd(desc, d["code"], d["code_pos"], d["coords1"], d["coords2"], d["perms"], o[5], o[1], o[2], o[4],
                                gi, ge, gn, None, None, None, neg_is_mean=True)
        return [o[0], o[1], o[2], o[3], o[4], dc, dcp]

    assert capi.corr_fwd_launches(desc, d["feats"], d["feats_pos"], d["code"], d["code_pos"]) == 1
    quiet = [t.clone() for t in step()]
    torch.cuda.synchronize()
    rec = {"backend": "nccl", "world": 1, "eager_rounds": 0, "graph_rounds": 0, "eager_equal": True, "graph_equal": True, "diffs": []}
    names = ("means", "intra_cd", "inter_cd", "neg_loss", "neg_cd", "d_code", "d_code_pos")

    def same(got, mode, rnd):
        ok = True
        for n, a, b in zip(names, got, quiet):
            if not torch.equal(a, b):       # (for the record of a failure: which output, by how much, in which round)
                ok = False
                rec["diffs"].append([mode, rnd, n, float((a.float() - b.float()).abs().max()), int((a != b).sum())])
        return ok
    side = torch.cuda.Stream()
    for _ in range(12):                                 # eager: the collective on its own stream, the step beside it
        with torch.cuda.stream(side):
            w1 = dist.all_reduce(bucket, op=dist.ReduceOp.AVG, async_op=True)
            w2 = dist.all_reduce(bucket, op=dist.ReduceOp.AVG, async_op=True)
        got = step()
        w1.wait(); w2.wait()
        torch.cuda.synchronize()
        rec["eager_equal"] = same(got, "eager", rec["eager_rounds"]) and rec["eager_equal"]
        rec["eager_rounds"] += 1
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):                              # (RCCL sets up its capture-time resources on the first calls)
            dist.all_reduce(bucket, op=dist.ReduceOp.AVG)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
            fork = torch.cuda.Stream()
            fork.wait_stream(s)
            with torch.cuda.stream(fork):               # a parallel branch of the graph: the collective beside the step
                dist.all_reduce(bucket, op=dist.ReduceOp.AVG)
            held = step()
            s.wait_stream(fork)
    torch.cuda.synchronize()
    for _ in range(12):
        g.replay()
        torch.cuda.synchronize()
        rec["graph_equal"] = same(held, "graph", rec["graph_rounds"]) and rec["graph_equal"]
        rec["graph_rounds"] += 1
    rec["bucket_is_ones"] = bool((bucket == 1).all())   # (the mean over one rank)
    dist.destroy_process_group()
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
