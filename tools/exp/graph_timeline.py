#!/usr/bin/env python
"""What one replay of the captured product step (loss call with its draws + weighted sum + backward) puts on the device.
  run:    rocprofv3 --kernel-trace -d DIR -o gt -- python tools/exp/graph_timeline.py run
  report: python tools/exp/graph_timeline.py report DIR/gt_results.db
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run():
    import torch
    import bench
    from stego_amd.modules import ContrastiveCorrelationLoss
    dev = torch.device("cuda:0")
    cfg = bench.Cfg()
    C, H, W, K = bench.WORKLOADS["vits8_224"]
    B, S, n_neg = 32, 11, 5
    d = bench.make_inputs(B, C, H, W, K, S, n_neg, 1000, dev)
    loss_fn = ContrastiveCorrelationLoss(cfg)
    c, cp = d["code"].detach().clone().requires_grad_(True), d["code_pos"].detach().clone().requires_grad_(True)
    total = os.environ.get("TOTAL") == "1"

    def step():
        c.grad = None
        cp.grad = None
        if total:
            loss_fn.total(d["feats"], d["feats_pos"], None, None, c, cp, (cfg.pos_intra_weight, cfg.pos_inter_weight, cfg.neg_inter_weight))[0].backward()
            return
        (pil, _, pel, _, nl, _) = loss_fn(d["feats"], d["feats_pos"], None, None, c, cp)
        (cfg.pos_intra_weight * pil + cfg.pos_inter_weight * pel + cfg.neg_inter_weight * nl.mean()).backward()

    for _ in range(30):
        step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        step()
    torch.cuda.synchronize()
    for _ in range(40):
        g.replay()
    torch.cuda.synchronize()


def report(path):
    import sqlite3
    db = sqlite3.connect(path)
    cur = db.cursor()
    names = [n for (n,) in cur.execute("select name from sqlite_master where type='table'")]
    kd = next(n for n in names if n.startswith("rocpd_kernel_dispatch"))
    ks = next(n for n in names if n.startswith("rocpd_info_kernel_symbol"))
    rows = list(cur.execute("select d.start, d.end, s.kernel_name from %s d join %s s on d.kernel_id=s.id order by d.start" % (kd, ks)))
    # the replays are the tail of the trace: find the period by the last forward kernels
    fw = [i for i, r in enumerate(rows) if "corr_fused_kernel" in r[2]]
    per = fw[-1] - fw[-2]
    print("kernels per replay:", per)
    for rep in (3, 2):
        i0 = fw[-rep]
        # start of this replay = first kernel after the previous replay's last kernel
        lo = i0
        while lo > 0 and lo > fw[-rep - 1] and "corr_unsample" not in rows[lo - 1][2] and "corr_bwd" not in rows[lo - 1][2]:
            lo -= 1
        seq = rows[lo: lo + per]
        t0 = seq[0][0]
        print("replay -%d: span %.2f us (first kernel start -> last kernel end); period to the next replay's first kernel %.2f us" %
              (rep, (seq[-1][1] - t0) / 1e3, (rows[lo + per][0] - t0) / 1e3 if lo + per < len(rows) else float("nan")))
        prev_end = None
        for s, e, n in seq:
            gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
            print("   +%7.2f us  gap %5.2f  dur %6.2f  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, n.replace("_ZN5stego", "")[:90]))
            prev_end = e


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        report(sys.argv[2])
