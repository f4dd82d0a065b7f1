#!/usr/bin/env python
"""Benchmark of the correspondence-loss hot path on MI355X.

Metric (BASELINE.json): image-pairs/sec through the correspondence loss, B=32 per GPU,
224^2 crops, ViT-S/8 (C=384, 28x28 feature map, K=70 code channels, S=11, 5 negatives).
One "step" = forward + backward of the fused loss over one batch (per rank), through the
C ABI of libstego_corr.so; at N>1 each step is followed by the DDP collective of the
reference's training loop (one flat RCCL all-reduce of the segmentation-head gradients,
~0.82 MB for ViT-S).  Inputs are synthetic, resident in HBM before the timed region; a ring
of input sets larger than the 256 MB Infinity Cache is rotated so steps read from HBM.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # B/s, MI355X_MICROARCH.md chip-level table
HBM_ACHIEVABLE = 6.3e12    # B/s a streaming kernel reaches on this part (same guide: copy / read micro-benchmarks); reported beside the peak fraction
MFMA_F32_PEAK = 157.3e12   # FLOP/s, f32-input MFMA (same table)
MFMA_F16_PEAK = 2.5e15    # FLOP/s dense bf16 MFMA


class Cfg:
    """Shipped defaults of the reference (configs/train_config.yml:41-64)."""
    pointwise = True
    zero_clamp = True
    stabalize = False
    use_salience = False
    feature_samples = 11
    neg_samples = 5
    pos_intra_shift = 0.18
    pos_inter_shift = 0.12
    neg_inter_shift = 0.46
    pos_intra_weight = 0.67
    pos_inter_weight = 0.25
    neg_inter_weight = 0.63
    corr_precision = "f16x3"


WORKLOADS = {
    # name: (C, H, W, K)   (BASELINE.json configs[1] and configs[3])
    "vits8_224": (384, 28, 28, 70),
    "vitb8_320": (768, 40, 40, 70),
    "vitt16_224": (192, 14, 14, 70),       # vit_tiny/16 (src/dino/vision_transformer.py:259-263): not a BASELINE config, the third backbone width
}


def algorithmic_bytes_fwd(B, C, H, W, K, S, n_neg):
    """every distinct tensor once, fp32."""
    P = 2 + n_neg
    return 4 * (2 * B * C * H * W + 2 * B * K * H * W + 2 * B * S * S * 2) + 8 * n_neg * B + \
        4 * (P * B * S ** 4 + n_neg * B * S ** 4) + 12


def algorithmic_flops_fwd(B, C, K, S, n_neg):
    return 2 * (2 + n_neg) * B * S ** 4 * (C + K)


def algorithmic_bytes_bwd(B, K, H, W, S, n_neg):
    """Backward of the code side (the feature side is no_grad, modules.py:326): reads saved_w and cd of every pair-set, the
    sampled-code context of every set and the three upstream gradients (two scalars + one broadcast tensor: stride 0);
    writes d_code and d_code_pos.  Every distinct tensor once, fp32."""
    P = 2 + n_neg
    LDK = (K + 7) // 8 * 8
    nset = 2 + n_neg                       # anchor set is shared: (1 + 1 + n_neg) sampled sets of 128 x LDK rows
    return 4 * (2 * P * B * S ** 4 + nset * B * 128 * LDK + 2 * B * K * H * W)


def head_grad_numel(C, K, n_classes=27):
    """Trainable parameters whose gradients DDP all-reduces:
    cluster1 (C*K+K), cluster2 (C*C+C + C*K+K), linear_probe (K*n+n), cluster_probe (n*K)."""
    return (C * K + K) + (C * C + C + C * K + K) + (K * n_classes + n_classes) + n_classes * K


def make_inputs(B, C, H, W, K, S, n_neg, seed, dev, layout="cl"):
    """DINO-like synthetic features (SURVEY.md 8(d) distribution ii), channels-last strided views
    exactly as DinoFeaturizer hands them over (modules.py:97), plus the RNG draws of one step."""
    g = torch.Generator(device=dev).manual_seed(seed)
    R = 8
    proto = torch.randn(R, C, device=dev, generator=g)
    head = torch.randn(K, C, device=dev, generator=g) / C ** 0.5

    def feat():
        z = torch.randn(B, H // 4 + 2, W // 4 + 2, R, device=dev, generator=g)
        z = z.repeat_interleave(4, 1).repeat_interleave(4, 2)[:, :H, :W]
        x = z @ proto + 0.3 * torch.randn(B, H, W, C, device=dev, generator=g)
        keep = (torch.rand(B, 1, 1, C, device=dev, generator=g) > 0.1).float() / 0.9
        return (x * keep).contiguous()            # [B,H,W,C] memory

    f, fp = feat(), feat()
    c = (f @ head.t()).contiguous()
    cp = (fp @ head.t()).contiguous()
    coords1 = torch.rand(B, S, S, 2, device=dev, generator=g) * 2 - 1
    coords2 = torch.rand(B, S, S, 2, device=dev, generator=g) * 2 - 1
    perms = []
    for _ in range(n_neg):
        p = torch.randperm(B, device=dev, generator=g)
        p = torch.where(p == torch.arange(B, device=dev), p + 1, p) % B
        perms.append(p)
    perms = torch.stack(perms) if perms else torch.zeros(0, B, dtype=torch.long, device=dev)
    maps = [t.permute(0, 3, 1, 2) for t in (f, fp, c, cp)]        # channels-last strided views (modules.py:97)
    if layout == "nchw":
        maps = [t.contiguous() for t in maps]                      # NCHW-contiguous: the generic (scalar gather) paths
    return dict(feats=maps[0], feats_pos=maps[1], code=maps[2], code_pos=maps[3],
                coords1=coords1, coords2=coords2, perms=perms)


def product_path(sets, cfg, args, kernel_step_s):
    """Default (the reference's RNG draws call for call) plus the opt-in cfg.fast_draws variant (same distributions from one
    kernel: not the reference's random stream)."""
    import copy
    out = _product_path(sets, cfg, args, kernel_step_s)
    fast = copy.copy(cfg)
    fast.fast_draws = True
    out["fast_draws"] = _product_path(sets, fast, args, kernel_step_s)
    out["fast_draws"]["what"] = "the same call with cfg.fast_draws=True (opt-in): draws of the reference's distributions from one Philox kernel"
    # what stego_amd's own training step calls: ContrastiveCorrelationLoss.total() - the three means come out of the forward launch,
    # one dot product combines them, the backward takes three device scalars
    out["total_api"] = _product_path(sets, cfg, args, kernel_step_s, total=True)
    out["total_api"]["what"] = "loss_fn.total(feats, feats_pos, None, None, code, code_pos, (0.67, 0.25, 0.63))[0].backward(), the reference's draws"
    out["total_api_fast_draws"] = _product_path(sets, fast, args, kernel_step_s, total=True)
    out["total_api_fast_draws"]["what"] = "the same with cfg.fast_draws=True"
    return out


def _product_path(sets, cfg, args, kernel_step_s, total=False):
    """Times the op as a training step calls it (train_segmentation.py:163-181): ContrastiveCorrelationLoss.forward with its own
    RNG draws (coords1, coords2, one randperm per negative), the weighted sum and .backward() into the code maps -
    eagerly, and replayed from a HIP graph (the draws are captured with the generator's graph-safe offsets)."""
    from stego_amd.modules import ContrastiveCorrelationLoss
    dev = sets[0]["feats"].device
    loss_fn = ContrastiveCorrelationLoss(cfg)
    codes = [(d["code"].detach().clone().requires_grad_(True), d["code_pos"].detach().clone().requires_grad_(True)) for d in sets]

    def step(i):
        d = sets[i]
        c, cp = codes[i]
        c.grad = None
        cp.grad = None
        if total:
            loss_fn.total(d["feats"], d["feats_pos"], None, None, c, cp,
                          (cfg.pos_intra_weight, cfg.pos_inter_weight, cfg.neg_inter_weight))[0].backward()
            return
        (pil, _, pel, _, nl, _) = loss_fn(d["feats"], d["feats_pos"], None, None, c, cp)
        (cfg.pos_intra_weight * pil + cfg.pos_inter_weight * pel + cfg.neg_inter_weight * nl.mean()).backward()

    n = len(sets)
    for k in range(3 * n + 40):            # warm-up: allocator pools, the draws' self-check, the autograd engine's device thread
        step(k % n)
    torch.cuda.synchronize()
    steps = max(200, args.steps // 2)
    t0 = time.perf_counter()
    for k in range(steps):
        step(k % n)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / steps
    out = {"eager_ms_per_step": eager * 1e3, "eager_over_kernel_step": eager / kernel_step_s, "steps": steps,
           "what": "ContrastiveCorrelationLoss(cfg)(feats, feats_pos, None, None, code, code_pos) incl. RNG draws; "
                   "0.67 intra + 0.25 inter + 0.63 neg.mean(); .backward() into code / code_pos"}
    try:
        graphs = []
        for i in range(n):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):      # (RCCL's watchdog thread must not break a capture)
                step(i)
            graphs.append(g)
        torch.cuda.synchronize()
        for k in range(8):
            graphs[k % n].replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            graphs[k % n].replay()
        torch.cuda.synchronize()
        gt = (time.perf_counter() - t0) / steps
        out.update(graph_ms_per_step=gt * 1e3, graph_over_kernel_step=gt / kernel_step_s)
    except Exception as e:      # noqa: BLE001
        out.update(graph_ms_per_step=None, graph_error="%s: %s" % (type(e).__name__, str(e)[:200]))
        torch.cuda.synchronize()
    return out


def measured_traffic(workload, B, precision, timeout_s=90):
    """HBM / fabric bytes of ONE forward launch, measured by THIS run (missing 4): two rocprofv3 counter passes
    (--kernel-trace --pmc FETCH_SIZE, then WRITE_SIZE: one counter group per pass, no other tracing - MI355X_MICROARCH.md) over a child
    process that launches the same forward 12 times on rotating inputs (tools/exp/fwd_loop.py); 2 x FETCH_SIZE + WRITE_SIZE, KB -> bytes
    (gfx950 tallies 128-byte requests at 64 B: the guide's correction).  None when rocprofv3 is missing or fails - the caller then
    falls back on the builder's constant and says so."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    vals = {}
    env = dict(os.environ, B=str(B), WL=workload, TMPDIR="/tmp")
    if precision == "f32":
        env["PREC"] = "f32"
    for cnt in ("FETCH_SIZE", "WRITE_SIZE"):
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            try:
                subprocess.run([exe, "--kernel-trace", "--pmc", cnt, "-d", td, "-o", "pmc", "--", sys.executable,
                                os.path.join(ROOT, "tools", "exp", "fwd_loop.py"), "12"], env=env, cwd="/tmp", timeout=timeout_s,
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
                dbs = [os.path.join(r, f) for r, _, fs in os.walk(td) for f in fs if f.endswith("_results.db")]
                db = sqlite3.connect(dbs[0])
                cur = db.cursor()
                tabs = [n for (n,) in cur.execute("select name from sqlite_master where type='table'")]
                pick = lambda pre: next(n for n in tabs if n.startswith(pre))      # noqa: E731
                q = ("select s.kernel_name, count(*), avg(e.value) from %s e join %s p on e.pmc_id=p.id join %s d on e.event_id=d.event_id "
                     "join %s s on d.kernel_id=s.id where p.name='%s' group by s.kernel_name"
                     % (pick("rocpd_pmc_event"), pick("rocpd_info_pmc"), pick("rocpd_kernel_dispatch"), pick("rocpd_info_kernel_symbol"), cnt))
                rows = [r for r in cur.execute(q) if "corr_fused" in r[0]]
                db.close()
                if not rows:
                    return None
                vals[cnt] = (rows[0][2], rows[0][1], rows[0][0])
            except Exception:       # noqa: BLE001
                return None
    kb = 2.0 * vals["FETCH_SIZE"][0] + vals["WRITE_SIZE"][0]
    return {"bytes": int(kb * 1024), "fetch_kb": vals["FETCH_SIZE"][0], "write_kb": vals["WRITE_SIZE"][0], "samples": vals["FETCH_SIZE"][1],
            "kernel": vals["FETCH_SIZE"][2].replace("_ZN5stego", "")[:48]}


def cpu_baseline(B, C, H, W, K, S, n_neg, cfg, budget_s=20.0):
    """The reference's CPU path, forward+backward, on the host cores of this box; a bounded sample.  BASELINE.md 2: "imported
    unmodified" - when /root/reference is present (the build container) the timed callable IS the reference's own
    ContrastiveCorrelationLoss.forward (src/modules.py:349-398, through oracle/ref_shim.py; kind "reference", its own RNG draws);
    on the GPU box the reference does not exist and the torch CPU port of the same lines is timed (oracle/torch_cpu_port.py, checked
    against the reference-generated goldens; kind "port")."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    d = make_inputs(B, C, H, W, K, S, n_neg, 4321, torch.device("cpu"))
    code = d["code"].clone().requires_grad_(True)
    code_pos = d["code_pos"].clone().requires_grad_(True)
    kind, what = "port", "port = oracle/torch_cpu_port.py (the ATen CPU kernels the reference calls)"
    ref_loss = None
    try:
        from oracle import ref_shim
        if ref_shim.available():
            ref_loss = ref_shim.load_reference_modules().ContrastiveCorrelationLoss(cfg)
            kind, what = "reference", "reference = the unmodified /root/reference/src/modules.py:349-398 through oracle/ref_shim.py (its own torch.rand / randperm draws)"
    except Exception:       # noqa: BLE001 - the port is the fallback, and says so
        ref_loss = None
    if ref_loss is None:
        from oracle.torch_cpu_port import corr_loss_torch_cpu

    def step():
        code.grad = None
        code_pos.grad = None
        if ref_loss is not None:
            out = ref_loss(d["feats"], d["feats_pos"], None, None, code, code_pos)
        else:
            out = corr_loss_torch_cpu(d["feats"], d["feats_pos"], code, code_pos, d["coords1"], d["coords2"],
                                      list(d["perms"]), cfg)
        (cfg.pos_intra_weight * out[0] + cfg.pos_inter_weight * out[2] + cfg.neg_inter_weight * out[4].mean()).backward()

    # intra-op thread count: the ATen CPU kernels stop scaling (and then collapse) well below the 256
    # hardware threads of the bench box, so take the fastest of a few candidates, one trial step each
    t_lim = time.perf_counter() + budget_s
    best, cores = None, 1
    for n in sorted({c for c in (8, 16, 32, 64) if c <= avail} or {avail}):
        torch.set_num_threads(n)
        step()
        t0 = time.perf_counter()
        step()
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, cores = dt, n
        if time.perf_counter() > t_lim:
            break
    torch.set_num_threads(cores)
    step()
    times = []
    t_end = time.perf_counter() + budget_s
    while len(times) < 12 and (time.perf_counter() < t_end or len(times) < 3):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return dict(value=B / med, unit="image-pairs/s", cores=cores, kind=kind,
                sample="%d fwd+bwd steps of the same B=%d workload (median %.1f ms), torch %s CPU, %d threads "
                       "(fastest of 8/16/32/64 tried; %d hardware threads visible); %s"
                       % (len(times), B, med * 1e3, torch.__version__, cores, avail, what))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (B)")
    ap.add_argument("--traffic", choices=["measure", "static", "none"], default="measure",
                    help="roofline.traffic: two rocprofv3 counter passes over a child process in this run (default; falls back on the "
                         "builder's constant of profiles/traffic.json when rocprofv3 is not usable), that constant, or nothing")
    ap.add_argument("--workload", default="vits8_224", choices=sorted(WORKLOADS))
    ap.add_argument("--feature-samples", type=int, default=0,
                    help="cfg.feature_samples (default: the reference's 11); 12 .. 16 run the multi-launch path of csrc/corr_wide.hip")
    ap.add_argument("--precision", default="f16x3", choices=["f32", "f16x3"],
                    help="contraction arithmetic of the feature correlation: f16x3 = fp16 hi+lo split products on the matrix "
                         "cores with f32 accumulation (22-bit products; measured error equals the f32 path); f32 = "
                         "v_mfma_f32 (runs at the VALU rate on gfx950)")
    ap.add_argument("--no-alt", action="store_true", help="skip the short run in the other precision mode")
    ap.add_argument("--sets", type=int, default=4, help="input sets rotated (4 x 91 MB > 256 MB Infinity Cache)")
    ap.add_argument("--no-graph", action="store_true", help="same as --launch eager")
    ap.add_argument("--launch", default="auto", choices=["auto", "eager", "graph"],
                    help="eager launches through the C ABI, HIP-graph replays, or (auto) whichever a short trial of both finds faster")
    ap.add_argument("--steps-per-graph", type=int, default=0,
                    help="steps captured per HIP graph (0 = one graph holding one step of every rotated input set)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--shared-device", choices=["auto", "0", "1"], default="auto",
                    help="STEGO_SHARED_DEVICE for the fused forward (one workgroup per tile instead of one per CU); auto = when a collective runs")
    ap.add_argument("--force-collective", action="store_true",
                    help="initialise RCCL and run the per-step all-reduce even with one rank (exercises the N > 1 path)")
    ap.add_argument("--fwd-only", action="store_true", help="time the forward alone (reported in config)")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="CI only: run the multi-rank protocol of this file (rendezvous, per-step gradient all-reduce through "
                         "FlatGradReducer, barriers, MAX-over-ranks timing, rank-0 JSON) on CPU with gloo and NO kernels; "
                         "the record says so and carries no throughput claim")
    ap.add_argument("--layout", default="cl", choices=["cl", "nchw"],
                    help="cl = channels-last strided views as DinoFeaturizer emits (default); nchw = contiguous NCHW")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dry = args.dry_run_cpu
    if dry:
        dev = torch.device("cpu")
    else:
        assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU path exists)"
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.force_collective:
        import torch.distributed as dist
        if args.force_collective and "RANK" not in os.environ:      # single-process self-test of the N > 1 code path
            os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
        if dry:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)       # backend "nccl" is RCCL on ROCm
    assert args.gpus == world, "--gpus must equal WORLD_SIZE (launch with torch.distributed.run)"

    cfg = Cfg()
    cfg.corr_precision = args.precision
    if args.feature_samples:
        cfg.feature_samples = args.feature_samples
    C, H, W, K = WORKLOADS[args.workload]
    B, S, n_neg = args.batch, cfg.feature_samples, cfg.neg_samples
    # the DDP exchange of the training loop: the trainer's own FlatGradReducer over a bucket of the head's size
    # (stego_amd/ddp.py; LitUnsupervisedSegmenter.manual_backward calls exactly this)
    from stego_amd.ddp import FlatGradReducer
    head = torch.nn.Parameter(torch.zeros(head_grad_numel(C, K), device=dev))
    reducer = FlatGradReducer([head])
    grad_buf = reducer.flat
    if dry:
        capi = None
        sets = [None] * args.sets
        args.no_alt = True
        args.no_cpu_baseline = True
        args.launch = "eager"
    else:
        from stego_amd import capi
        if args.shared_device == "1" or (args.shared_device == "auto" and dist is not None):
            # the all-reduce of step t overlaps the kernels of step t + 1: the fused forward leaves the CUs beyond its tiles alone
            capi.set_shared_device(True)
        sets = [make_inputs(B, C, H, W, K, S, n_neg, 1234 + 97 * rank + i, dev, args.layout) for i in range(args.sets)]
        # upstream gradients exactly as train_segmentation.py:169-181 produces them
        g_intra = torch.tensor(cfg.pos_intra_weight, device=dev)
        g_inter = torch.tensor(cfg.pos_inter_weight, device=dev)
        g_neg = torch.full((1,), cfg.neg_inter_weight / (n_neg * B * S ** 4), device=dev).expand(n_neg * B, S, S, S, S)
        from stego_amd.modules import as_channels_last      # the host-side layout policy of the op (no-op for cl views)

    def barrier():
        if dist is not None:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    last_local = [0.0]                       # this rank's own seconds of the last timed_run (before the MAX over ranks)

    def timed_run(precision, steps, warmup, fwd_only=args.fwd_only, collective=True, use_graph=True):
        """W untimed + K timed steps of the whole job in one precision mode; returns (seconds, launch mode, desc)."""
        collective_on = collective
        keep = [None] * args.sets
        if dry:
            desc = None
            counter = [0]

            def step_compute(i):                    # no kernels: each step leaves "its gradient" rank + 1 in the bucket
                counter[0] += 1
                grad_buf.fill_(float(rank + 1))
        else:
            prec = capi.PREC_F32 if precision == "f32" else capi.PREC_F16X3
            desc = capi.make_desc(B, C, K, H, W, S, n_neg, cfg, (cfg.pos_intra_shift, cfg.pos_inter_shift, cfg.neg_inter_shift),
                                  prec)

        def step_compute_gpu(i):
            d = sets[i]
            need_grad = not fwd_only
            out = capi.corr_fwd(desc, as_channels_last(d["feats"]), as_channels_last(d["feats_pos"]),
                                as_channels_last(d["code"]), as_channels_last(d["code_pos"]), d["coords1"], d["coords2"],
                                d["perms"], need_grad)
            if need_grad:
                lm, icd, ecd, nl, ncd, saved = out
                grads = capi.corr_bwd(desc, d["code"], d["code_pos"], d["coords1"], d["coords2"], d["perms"], saved,
                                      icd, ecd, ncd, g_intra, g_inter, g_neg, None, None, None)
                keep[i] = (out, grads)
            else:
                keep[i] = (out,)

        if not dry:
            step_compute = step_compute_gpu
        for i in range(args.sets):          # eager warm-up (also sets kernel attributes before any capture)
            step_compute(i)
        if not dry:
            torch.cuda.synchronize()
        graphs = None
        launch = "eager"
        # a torch.distributed call costs ~40 us of host time, which eager stepping at ~100 us/step cannot hide: in graph mode
        # the all-reduce of every step is captured with the step (RCCL collectives are graph-capturable) and replays for free
        coll_in_graph = use_graph and dist is not None and collective_on and not dry
        if use_graph and not dry:
            try:
                # One replay runs `spg` consecutive steps (rotating through the input sets): a hipGraphLaunch costs ~18 us on
                # top of its kernels whatever it holds (measured: one 58 us kernel per graph = 76 us per replay), which a
                # real training step amortises over its backbone forward; K steps are always exactly K steps of work.
                spg = args.steps_per_graph or args.sets
                graphs = []
                for g0 in range(0, args.sets * spg, spg):
                    gr = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gr, capture_error_mode="thread_local"):     # RCCL's watchdog thread must not break a capture
                        prev = None
                        for j in range(spg):
                            step_compute((g0 + j) % args.sets)
                            if coll_in_graph:
                                # step j's all-reduce runs on RCCL's stream under step j + 1's kernels (as it runs under the next
                                # backbone forward in training); the graph joins the last one before it ends
                                if prev is not None:
                                    prev.wait()
                                prev = reducer.allreduce_mean(async_op=True, single_rank_ok=True)
                        if prev is not None:
                            prev.wait()
                    graphs.append(gr)
                    if spg % args.sets == 0:
                        break                       # every replay covers whole rotations: one graph is enough
                single = []
                for i in range(args.sets):          # single-step graphs for the remainder of K
                    gr = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gr, capture_error_mode="thread_local"):
                        step_compute(i)
                        if coll_in_graph:
                            reducer.allreduce_mean(async_op=False, single_rank_ok=True)
                    single.append(gr)
                launch = "hipgraph (%d steps per replay%s)" % (spg, ", the gradient all-reduce captured in the graph" if coll_in_graph else "")
            except Exception as e:      # noqa: BLE001 - fall back to eager launches, say so in the output
                graphs = None
                launch = "eager (graph capture failed: %s)" % type(e).__name__
                torch.cuda.synchronize()
            if dist is not None and world > 1:
                # every rank must step the same way: a rank that replays graphs holding the all-reduce and a rank that launches eagerly
                # with its own all-reduce calls would wait for each other forever - if ANY rank's capture failed, all go eager
                ok = torch.tensor([1 if graphs is not None else 0], device=dev, dtype=torch.int32)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()) == 0 and graphs is not None:
                    graphs = None
                    launch = "eager (graph capture failed on another rank)"

        pending = [None]

        def step(k):
            i = k % args.sets
            if graphs is not None:
                single[i].replay()
            else:
                step_compute(i)
            do_collective()

        def do_collective():
            if dist is not None and collective_on and not (graphs is not None and coll_in_graph):
                # gradients of the segmentation head only (backbone frozen): one flat bucket per step.  async_op: RCCL's
                # stream first waits for this step's kernels, then the all-reduce runs while the next step computes - in
                # training it overlaps the next step's backbone forward the same way; every all-reduce is complete
                # before the clock stops (drain()).
                pending[0] = reducer.allreduce_mean(async_op=True, single_rank_ok=True)

        def run(n):
            """exactly n steps: whole multi-step replays first, single-step graphs for the rest"""
            k = 0
            if graphs is not None:
                spg = args.steps_per_graph or args.sets
                r = 0
                while n - k >= spg:
                    graphs[r % len(graphs)].replay()
                    r += 1
                    for _ in range(spg):
                        do_collective()
                    k += spg
            while k < n:
                step(k)
                k += 1

        def drain():
            if pending[0] is not None:
                pending[0].wait()
                pending[0] = None

        run(warmup)
        drain()
        barrier()
        t0 = time.perf_counter()
        run(steps)
        drain()
        barrier()
        dt = time.perf_counter() - t0
        last_local[0] = dt
        if dist is not None:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        if dry:
            launch = "dry run on CPU: no kernels (%d no-op steps per rank)" % counter[0]
        return dt, launch, desc

    # ---- launch mode: the C ABI is asynchronous either way.  With three kernels per step, eager launches keep the queue full
    # (measured 121.7 us/step); a HIP graph replay adds ~5 us per replay on this stack (129.3 with one step per replay, 124.0 with
    # eight).  auto = a short trial of both, the faster one runs the K timed steps (all ranks agree through a MAX-reduce).
    mode = "eager" if args.no_graph else args.launch
    trial = None
    if mode == "auto":
        trial = {}
        for m in ("eager", "graph"):
            dt_m, _, _ = timed_run(args.precision, 48, 16, use_graph=(m == "graph"))
            trial[m] = dt_m / 48 * 1e3
        if dist is not None:                # every rank takes the same decision: the slowest rank's trial times
            tt = torch.tensor([trial["eager"], trial["graph"]], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            trial = {"eager": float(tt[0]), "graph": float(tt[1])}
        mode = "graph" if trial["graph"] < trial["eager"] else "eager"
    dt, launch, desc = timed_run(args.precision, args.steps, args.warmup, use_graph=(mode == "graph"))
    dt_local = last_local[0]
    use_graph = mode == "graph"
    alt = None
    if not args.no_alt:                     # the other arithmetic mode, shorter, for the record (all ranks take part)
        other = "f32" if args.precision == "f16x3" else "f16x3"
        steps_alt = max(20, args.steps // 4)
        dt_alt, _, desc_alt = timed_run(other, steps_alt, max(4, args.warmup // 4), use_graph=use_graph)
        alt = {"precision": other, "value": world * B * steps_alt / dt_alt, "unit": "image-pairs/s",
               "ms_per_step": dt_alt / steps_alt * 1e3, "steps": steps_alt, "desc": desc_alt}

    # ---- the forward alone (same graphs without the backward): the backward's share is the difference
    split = None
    if not args.fwd_only and not args.no_alt and not dry:
        steps_f = max(20, args.steps // 4)
        dt_f, _, _ = timed_run(args.precision, steps_f, max(4, args.warmup // 4), fwd_only=True, collective=False, use_graph=use_graph)
        split = {"forward_ms": dt_f / steps_f * 1e3, "backward_ms": dt / args.steps * 1e3 - dt_f / steps_f * 1e3, "steps": steps_f}

    # ---- the product path: ContrastiveCorrelationLoss(cfg)(...) + .backward() exactly as a training step calls it
    # (train_segmentation.py:163-181): RNG draws, autograd.Function, weighted sum, backward - eager and graph-replayed
    product = None
    if rank == 0 and not args.no_alt and not dry:
        product = product_path(sets, cfg, args, dt / args.steps)

    # ---- the other native path, for the record: cfg.feature_samples = 16 (256 points per image: 8 + 4 launches of csrc/corr_wide.hip behind the
    # same two C-ABI calls), one input set, eager launches (the host needs ~0.16 ms per step, the kernels ~0.36)
    wide = None
    if rank == 0 and not args.no_alt and not dry and S <= 11 and not args.fwd_only:
        try:
            S16 = 16
            cfg16 = Cfg()
            cfg16.corr_precision = args.precision
            cfg16.feature_samples = S16
            d16 = make_inputs(B, C, H, W, K, S16, n_neg, 4242, dev, args.layout)
            desc16 = capi.make_desc(B, C, K, H, W, S16, n_neg, cfg16, (cfg16.pos_intra_shift, cfg16.pos_inter_shift, cfg16.neg_inter_shift),
                                    capi.PREC_F32 if args.precision == "f32" else capi.PREC_F16X3)
            gi, ge = (torch.full((1,), w_, device=dev) for w_ in (cfg16.pos_intra_weight, cfg16.pos_inter_weight))
            gn = torch.full((1,), cfg16.neg_inter_weight / (n_neg * B * S16 ** 4), device=dev).expand(n_neg * B, S16, S16, S16, S16)
            maps16 = [as_channels_last(d16[k]) for k in ("feats", "feats_pos", "code", "code_pos")]

            def step16():
                out16 = capi.corr_fwd(desc16, *maps16, d16["coords1"], d16["coords2"], d16["perms"], True)
                lm, icd, ecd, nl, ncd, saved = out16
                return capi.corr_bwd(desc16, d16["code"], d16["code_pos"], d16["coords1"], d16["coords2"], d16["perms"], saved, icd, ecd, ncd,
                                     gi, ge, gn, None, None, None)

            for _ in range(4):
                step16()
            torch.cuda.synchronize()
            n16 = 30
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n16):
                step16()
            e1.record()
            torch.cuda.synchronize()
            ms16 = e0.elapsed_time(e1) / n16
            wide = {"feature_samples": S16, "ms_per_step": ms16, "value": B / (ms16 * 1e-3), "unit": "image-pairs/s", "steps": n16,
                    "forward_launches": capi.corr_fwd_launches(desc16, *maps16), "launch": "eager",
                    "path": "csrc/corr_wide.hip: the multi-launch kernels behind stego_corr_fwd / stego_corr_bwd (4.5 x the pair products of feature_samples = 11)"}
            del d16, maps16
        except Exception as e:       # noqa: BLE001 - a record for the reader, never the reason a bench run fails
            wide = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- the reference's SHIPPED batch size, for the record (src/configs/train_config.yml:11: batch_size 16; BASELINE config 4 names B = 16
    # too): the same workload at B = 16 - since round 6 the column-half launch (csrc/corr_fused_half.hip) -, four rotating input sets, eager
    # launches; the forward by HIP events around single launches like the headline's, the full-tile launch of the same library beside it
    ref16 = None
    if rank == 0 and not args.no_alt and not dry and S == 11 and B == 32 and not args.fwd_only:
        try:
            B16 = 16
            sets16 = [make_inputs(B16, C, H, W, K, S, n_neg, 5000 + i, dev, args.layout) for i in range(4)]
            d16b = capi.make_desc(B16, C, K, H, W, S, n_neg, cfg, (cfg.pos_intra_shift, cfg.pos_inter_shift, cfg.neg_inter_shift),
                                  capi.PREC_F32 if args.precision == "f32" else capi.PREC_F16X3)
            gi16, ge16 = torch.tensor(cfg.pos_intra_weight, device=dev), torch.tensor(cfg.pos_inter_weight, device=dev)
            gn16 = torch.full((1,), cfg.neg_inter_weight / (n_neg * B16 * S ** 4), device=dev).expand(n_neg * B16, S, S, S, S)

            def step_b16(i):
                d = sets16[i % 4]
                o = capi.corr_fwd(d16b, as_channels_last(d["feats"]), as_channels_last(d["feats_pos"]), as_channels_last(d["code"]),
                                  as_channels_last(d["code_pos"]), d["coords1"], d["coords2"], d["perms"], True)
                lm, icd, ecd, nl, ncd, saved = o
                capi.corr_bwd(d16b, d["code"], d["code_pos"], d["coords1"], d["coords2"], d["perms"], saved, icd, ecd, ncd, gi16, ge16, gn16, None, None, None)

            def fwd_b16():
                acc, n = 0.0, 0
                for r in range(6):
                    for d in sets16:
                        k = capi.corr_fwd_profile(d16b, as_channels_last(d["feats"]), as_channels_last(d["feats_pos"]), as_channels_last(d["code"]),
                                                  as_channels_last(d["code_pos"]), d["coords1"], d["coords2"], d["perms"], True, 1)
                        if r > 0:
                            acc += k[1]
                            n += 1
                return acc / n

            for k in range(8):
                step_b16(k)
            torch.cuda.synchronize()
            nb16 = 100
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for k in range(nb16):
                step_b16(k)
            e1.record()
            torch.cuda.synchronize()
            ms_b16 = e0.elapsed_time(e1) / nb16
            f_half = fwd_b16()
            capi.debug_set("STEGO_DEBUG", 16384)            # the full-tile launch of the same library (tools only; reset right below)
            try:
                f_full = fwd_b16()
            finally:
                capi.debug_set("STEGO_DEBUG", 0)
            ab16 = algorithmic_bytes_fwd(B16, C, H, W, K, S, n_neg)
            ref16 = {"batch": B16, "ms_per_step": ms_b16, "value": B16 / (ms_b16 * 1e-3), "unit": "image-pairs/s", "steps": nb16, "launch": "eager",
                     "forward_us": f_half * 1e3, "forward_frac_of_hbm_peak": ab16 / (f_half * 1e-3) / HBM_PEAK,
                     "forward_us_full_tile_launch": f_full * 1e3,
                     "what": "src/configs/train_config.yml:11 ships batch_size 16: the same workload at B = 16; forward = corr_fused_half_kernel (one "
                             "workgroup per 128 x 64 column half, every compute unit busy) - next to it the full-tile launch of the same library "
                             "(STEGO_DEBUG bit 16384)"}
            del sets16
        except Exception as e:       # noqa: BLE001 - a record for the reader, never the reason a bench run fails
            ref16 = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- dominant kernel: HIP-event duration per launch, rotating input sets (HBM-cold like the timed loop)
    roof = roof_mfma = roof_bwd = None
    fin_us = None

    fwd_samples = {}                          # id(desc) -> every single-launch duration of the main forward kernel (ms)

    def forward_launch_ms(desc_):
        """(ms in front of, of, behind the main forward kernel) per launch: HIP events on the launch stream around single launches."""
        ms = [0.0, 0.0, 0.0]
        rounds = 5
        samples = fwd_samples.setdefault(id(desc_), [])
        for r in range(rounds + 1):
            acc = [0.0, 0.0, 0.0]
            for i in range(args.sets):
                d = sets[i]
                # the same host-side layout policy as the timed loop (a no-op for channels-last views): the kernels profiled
                # are the kernels the timed steps ran
                k = capi.corr_fwd_profile(desc_, as_channels_last(d["feats"]), as_channels_last(d["feats_pos"]),
                                          as_channels_last(d["code"]), as_channels_last(d["code_pos"]),
                                          d["coords1"], d["coords2"], d["perms"], not args.fwd_only, 1)
                acc = [x + y for x, y in zip(acc, k)]
                if r > 0:
                    samples.append(k[1])
            if r > 0:
                ms = [x + y / args.sets for x, y in zip(ms, acc)]
        return [x / rounds for x in ms]

    def pctl(xs, scale=1.0):
        """p10 / p50 / p90 of a list of durations (median and p10 / p90, not only a mean)."""
        if not xs:
            return None
        v = sorted(xs)
        at = lambda q: v[min(len(v) - 1, max(0, int(round(q * (len(v) - 1)))))] * scale      # noqa: E731
        return {"p10": at(0.10), "p50": at(0.50), "p90": at(0.90), "min": v[0] * scale, "max": v[-1] * scale, "n": len(v)}

    # ---- per-step distribution (outside the timed region): one HIP event behind every eagerly launched step on torch's current
    # stream - the stream capi launches on -, consecutive differences = what each step took incl. its launch gaps
    step_dist = None
    if rank == 0 and not dry:
        try:
            prec_ = capi.PREC_F32 if args.precision == "f32" else capi.PREC_F16X3
            desc_d = capi.make_desc(B, C, K, H, W, S, n_neg, cfg, (cfg.pos_intra_shift, cfg.pos_inter_shift, cfg.neg_inter_shift), prec_)

            def one_step(i):
                d = sets[i]
                out = capi.corr_fwd(desc_d, as_channels_last(d["feats"]), as_channels_last(d["feats_pos"]), as_channels_last(d["code"]),
                                    as_channels_last(d["code_pos"]), d["coords1"], d["coords2"], d["perms"], not args.fwd_only)
                if not args.fwd_only:
                    lm, icd, ecd, nl, ncd, saved = out
                    capi.corr_bwd(desc_d, d["code"], d["code_pos"], d["coords1"], d["coords2"], d["perms"], saved, icd, ecd, ncd,
                                  g_intra, g_inter, g_neg, None, None, None)

            nd = 200
            for k in range(20):
                one_step(k % args.sets)
            torch.cuda.synchronize()
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(nd + 1)]
            evs[0].record()
            for k in range(nd):
                one_step(k % args.sets)
                evs[k + 1].record()
            torch.cuda.synchronize()
            step_dist = pctl([evs[k].elapsed_time(evs[k + 1]) for k in range(nd)], 1e3)
            step_dist["unit"] = "us"
            step_dist["what"] = "%d eager steps (%s), one HIP event behind each on the launch stream: consecutive differences" % (
                nd, "forward only" if args.fwd_only else "forward+backward")
        except Exception as e:       # noqa: BLE001 - a record for the reader, never the reason a bench run fails
            step_dist = {"error": "%s: %s" % (type(e).__name__, e)}

    if rank == 0 and not dry:
        ms_samp, ms_main, ms_fin = forward_launch_ms(desc)
        ab = algorithmic_bytes_fwd(B, C, H, W, K, S, n_neg)
        fl = algorithmic_flops_fwd(B, C, K, S, n_neg)
        # HBM / fabric bytes per launch: NOT measured by this run (rocprofv3 counter passes cannot run inside the timed process) - the
        # constant that the builder's counter passes of this kernel produced, with where it comes from
        traffic = traffic_source = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if world == 1 and S == 11 and args.traffic == "measure":
            mt = measured_traffic(args.workload, B, args.precision)
            if mt is not None:
                traffic = mt["bytes"]
                traffic_source = ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE passes over a child process "
                                  "(tools/exp/fwd_loop.py, 12 launches, rotating inputs): 2 x %.1f + %.1f KB per launch of %s"
                                  % (mt["fetch_kb"], mt["write_kb"], mt["kernel"]))
        if traffic is None and args.traffic != "none" and os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                # (the constants are counter passes of the feature_samples = 11 single-launch kernel: any other S is a different set of
                # kernels with different traffic - no constant rather than the wrong one)
                traffic = tj.get("%s_%s_B%d" % (args.workload, args.precision, B)) if S == 11 else tj.get("%s_%s_B%d_S%d" % (args.workload, args.precision, B, S))
                if traffic is not None:
                    traffic_source = "static: profiles/traffic.json (%s) - builder-run rocprofv3 --pmc passes, 2 x FETCH_SIZE + WRITE_SIZE per launch; not measured in this run" % tj.get("_source", "see its _note")
            except Exception:       # noqa: BLE001
                traffic = None
        d0 = sets[0]
        n_launch = capi.corr_fwd_launches(desc, as_channels_last(d0["feats"]), as_channels_last(d0["feats_pos"]),
                                          as_channels_last(d0["code"]), as_channels_last(d0["code_pos"]))
        fused = n_launch == 1
        peak = MFMA_F32_PEAK if args.precision == "f32" else MFMA_F16_PEAK / 3.0
        # ---- the traffic skeleton of the same launch: tools/ubench/fused_skeleton.hip, built by
        # __graft_entry__.build() beside the library - the forward's grid, placement, gather addresses, LDS-DMA copies and output / context
        # stores with no normalisation and no rendezvous, on rotating inputs, timed by HIP events in its own process right here.
        # frac_of_skeleton = skeleton time / kernel time: how much of the launch is the memory system under this access pattern.
        skeleton = None
        if n_launch == 1 and K == 70 and S == 11 and n_neg == 5 and C in (384, 768) and H == W and B % 8 == 0 and world == 1:
            try:
                import subprocess
                from stego_amd import _build as _b
                if os.path.exists(_b.SKELETON_PATH):
                    o = subprocess.run([_b.SKELETON_PATH, str(B), str(C), str(H), "30", "--json"], stdout=subprocess.PIPE,
                                       stderr=subprocess.DEVNULL, timeout=120, check=True).stdout.decode().strip().splitlines()[-1]
                    sk = json.loads(o)
                    cus = torch.cuda.get_device_properties(dev).multi_processor_count & ~7
                    layout = "half" if (2 * (2 + n_neg) * B + 8 <= cus and "half" in sk) else "full"
                    skeleton = {"us": sk[layout]["us"], "p10": sk[layout]["p10"], "p90": sk[layout]["p90"],
                                "layout": {"full": "one workgroup per 128 x 128 tile (corr_fused_kernel)",
                                           "half": "one workgroup per 128 x 64 column half (corr_fused_half_kernel)"}[layout],
                                "other_layout_us": sk.get("half" if layout == "full" else "full", {}).get("us"),
                                "what": "tools/ubench/fused_skeleton.hip: same grid / placement / gather addresses / LDS-DMA copies / stores, "
                                        "MFMAs on whatever the ring holds, no normalisation, no rendezvous; %d launches on %d rotating input sets"
                                        % (sk.get("launches", 0), sk.get("input_sets", 0))}
            except Exception as e:       # noqa: BLE001 - a record for the reader, never the reason a bench run fails
                skeleton = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        if fused:
            # ONE launch does the whole forward (every distinct tensor of SURVEY.md 8(d) once): its duration prices all
            # algorithmic bytes.  ms_samp / ms_fin are the empty event intervals in front of / behind it.
            t_fwd = ms_main * 1e-3
            ach = ab / t_fwd
            cus_ = torch.cuda.get_device_properties(dev).multi_processor_count & ~7
            half_ = 2 * (2 + n_neg) * B + 8 <= cus_ and C in (384, 768) and K % 2 == 0 and S * S > 64 and not (args.shared_device == "1")
            kname = "corr_fused_half_kernel" if half_ else "corr_fused_kernel"
            roof = dict(bound="hbm", kernel="%s (the whole forward, one launch)" % kname, dominant_kernel=kname,
                        achieved=ach / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s", frac=ach / HBM_PEAK,
                        frac_of_achievable=ach / HBM_ACHIEVABLE, achievable_peak=HBM_ACHIEVABLE / 1e9,
                        traffic=traffic, traffic_source=traffic_source,
                        algorithmic_bytes=ab, us_per_launch={kname: ms_main * 1e3},
                        us_per_launch_dist=pctl(fwd_samples.get(id(desc), []), 1e3),
                        skeleton=skeleton,
                        frac_of_skeleton=(skeleton["us"] / (ms_main * 1e3)) if skeleton and "us" in skeleton and ms_main > 0 else None,
                        timing="HIP events on the launch stream around single launches, input sets rotated")
            roof_mfma = dict(bound="mfma", kernel="corr_fused_kernel", achieved=fl / t_fwd / 1e12, peak=peak / 1e12,
                             unit="TFLOP/s", frac=fl / t_fwd / peak, algorithmic_flops=fl,
                             note="f32: v_mfma_f32_32x32x2_f32 peak; f16x3: dense fp16 peak / 3 (three MFMAs per product)")
        elif n_launch > 3:
            # feature_samples 12 .. 16 (csrc/corr_wide.hip): the whole multi-launch forward between two events
            t_fwd = ms_main * 1e-3
            ach = ab / t_fwd
            what = "forward = %d launches (csrc/corr_wide.hip: samplers, two correlations, three elementwise)" % n_launch
            roof = dict(bound="hbm", kernel=what, dominant_kernel="dense_rowblock_kernel", achieved=ach / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s",
                        frac=ach / HBM_PEAK, frac_of_achievable=ach / HBM_ACHIEVABLE, achievable_peak=HBM_ACHIEVABLE / 1e9,
                        traffic=traffic, traffic_source=traffic_source, algorithmic_bytes=ab, us_per_launch={"forward (%d launches)" % n_launch: ms_main * 1e3},
                        timing="HIP events on the launch stream around the forward's launches, input sets rotated")
            roof_mfma = dict(bound="mfma", kernel=what, achieved=fl / t_fwd / 1e12, peak=peak / 1e12, unit="TFLOP/s", frac=fl / t_fwd / peak,
                             algorithmic_flops=fl, note="dense fp16 peak / 3 (three MFMAs per product)")
        else:
            fin_us = ms_fin * 1e3
            kernels = {"sample_norm_kernel": ms_samp * 1e3, "corr_tile_kernel": ms_main * 1e3,
                       "corr_finalize_kernel": ms_fin * 1e3}
            ab_in = 4 * (B * C * H * W + 2 * B * K * H * W + 2 * B * S * S * 2) + 8 * n_neg * B
            ab_out = ab - ab_in
            dom = "sample_norm_kernel" if ms_samp >= ms_main else "corr_tile_kernel"
            t_fwd = (ms_samp + ms_main + ms_fin) * 1e-3
            ach = ab / t_fwd
            roof = dict(bound="hbm", kernel="forward = sample_norm_kernel + corr_tile_kernel + corr_finalize_kernel",
                        dominant_kernel=dom, achieved=ach / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s",
                        frac=ach / HBM_PEAK, frac_of_achievable=ach / HBM_ACHIEVABLE, achievable_peak=HBM_ACHIEVABLE / 1e9,
                        traffic=traffic, traffic_source=traffic_source, algorithmic_bytes=ab, us_per_launch=kernels,
                        per_kernel={"sample_norm_kernel": dict(algorithmic_bytes=ab_in, achieved_GBps=ab_in / (ms_samp * 1e-3) / 1e9,
                                                               frac=ab_in / (ms_samp * 1e-3) / HBM_PEAK),
                                    "corr_tile_kernel": dict(algorithmic_bytes=ab_out, achieved_GBps=ab_out / (ms_main * 1e-3) / 1e9,
                                                             frac=ab_out / (ms_main * 1e-3) / HBM_PEAK)})
            roof_mfma = dict(bound="mfma", kernel="corr_tile_kernel", achieved=fl / (ms_main * 1e-3) / 1e12,
                             peak=peak / 1e12, unit="TFLOP/s", frac=fl / (ms_main * 1e-3) / peak, algorithmic_flops=fl,
                             note="f32: v_mfma_f32_32x32x2_f32 peak; f16x3: dense fp16 peak / 3 (three MFMAs per product)")
        if alt is not None and alt.get("desc") is not None:
            # the other arithmetic mode's forward, measured the same way (the strict like-for-like number when the headline is f16x3)
            d0a = alt.pop("desc")
            _, ms_alt, _ = forward_launch_ms(d0a)
            if fused and ms_alt > 0:
                ach_alt = ab / (ms_alt * 1e-3)
                alt["roofline"] = dict(bound="hbm", kernel="corr_fused_kernel", achieved=ach_alt / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s",
                                       frac=ach_alt / HBM_PEAK, frac_of_achievable=ach_alt / HBM_ACHIEVABLE, algorithmic_bytes=ab,
                                       us_per_launch={"corr_fused_kernel": ms_alt * 1e3})
        if split is not None:
            abb = algorithmic_bytes_bwd(B, K, H, W, S, n_neg)
            tb = split["backward_ms"] * 1e-3
            roof_bwd = dict(bound="hbm", kernel="backward = corr_bwd_tile_build_kernel (corr_bwd_tile32_build_kernel in F32 mode) + corr_unsample_list_kernel (step time minus the forward-only step time)",
                            achieved=abb / tb / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s", frac=abb / tb / HBM_PEAK,
                            algorithmic_bytes=abb, us=tb * 1e6)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(B, C, H, W, K, S, n_neg, cfg)

    check = None
    if dry:
        # after the last step's averaged all-reduce every rank holds mean(rank + 1) = (world + 1) / 2
        check = {"grad_mean_after_allreduce": float(grad_buf[0]), "expected": (world + 1) / 2.0,
                 "bucket_numel": int(grad_buf.numel())}
    if alt is not None:
        alt.pop("desc", None)
    # ---- N > 1 (and --force-collective): what each rank saw, and the gradient all-reduce on its own, so that a scaling run explains itself
    coll_stats = None
    if dist is not None:
        per_rank = torch.zeros(world, device=dev, dtype=torch.float64)
        per_rank[rank] = dt_local / args.steps * 1e3
        dist.all_reduce(per_rank)
        for _ in range(5):
            reducer.allreduce_mean(async_op=False, single_rank_ok=True)
        barrier()
        n_ar = 50
        if dry:                                  # (CPU / gloo: the protocol only - wall clock)
            t_ar = time.perf_counter()
        else:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        for _ in range(n_ar):
            reducer.allreduce_mean(async_op=False, single_rank_ok=True)      # (the stream waits for RCCL's stream: back-to-back collectives)
        if dry:
            ar_us = (time.perf_counter() - t_ar) / n_ar * 1e6
        else:
            ev1.record()
            torch.cuda.synchronize()
            ar_us = ev0.elapsed_time(ev1) / n_ar * 1e3
        ar = torch.tensor([ar_us], device=dev, dtype=torch.float64)
        dist.all_reduce(ar, op=dist.ReduceOp.MAX)
        coll_stats = {"ms_per_step_by_rank": [float(x) for x in per_rank.tolist()], "ms_per_step_rank_min": float(per_rank.min()),
                      "ms_per_step_rank_max": float(per_rank.max()),
                      "allreduce_us": float(ar.item()), "allreduce_what": "%d back-to-back FlatGradReducer.allreduce_mean calls of the %d-float bucket, HIP events on the compute stream (which waits for RCCL's), max over ranks" % (n_ar, grad_buf.numel())}
    if rank == 0:
        value = world * B * args.steps / dt
        rec = {
            "metric": "image-pairs/sec through correspondence loss, B=32 224^2, ViT-S/8",
            "value": value, "unit": "image-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.precision == "f32" else "f16x3 (f32 inputs / outputs / accumulation; products of the feature and "
                                                                   "code correlations and of the backward's code GEMMs as fp16 hi+lo splits, 22-bit, on the "
                                                                   "matrix cores - error vs fp64 in the fp32 class, bounded by tests on adversarial inputs)",
            "data": "synthetic" if not dry else "none (dry run of the multi-rank protocol on CPU: no kernels ran, value is not a measurement)",
            "config": {"workload": "%s: B=%d/GPU, C=%d, %dx%d map, K=%d, S=%d, %d negatives, self+KNN+random "
                                   "correlation loss, %s" % (args.workload, B, C, H, W, K, S, n_neg,
                                                             "forward only" if args.fwd_only else "forward+backward"),
                       "global_batch": world * B, "parallelism": "dp%d" % world, "launch": launch,
                       "launch_trial_ms_per_step": trial,
                       "input_sets_rotated": args.sets, "layout": "channels-last strided views (as DinoFeaturizer)" if args.layout == "cl" else "NCHW contiguous",
                       "collective": ({"what": "all_reduce(%d f32 head grads)/step, FlatGradReducer.allreduce_mean(async)" % grad_buf.numel(),
                                       **(coll_stats or {})})
                       if dist is not None else None,
                       "shared_device": (args.shared_device == "1" or (args.shared_device == "auto" and dist is not None)) and not dry},
            "roofline": roof, "roofline_mfma": roof_mfma, "roofline_bwd": roof_bwd, "forward_backward_split": split,
            "step_us_dist": step_dist,
            "product_path": product, "finalize_kernel_us": fin_us, "other_precision": alt, "feature_samples_16": wide, "reference_batch_16": ref16,
            "cpu_baseline": cpu,
        }
        if dry:
            rec["dry_run"] = True
            rec["value"] = None
            rec["collective_check"] = check
    # RCCL (NCCL_DEBUG=VERSION on the bench boxes) writes its banner through C stdio, which would otherwise be flushed at
    # exit, after the result: every rank pushes it out before the last barrier, rank 0 prints the record after it, so
    # the JSON line is the last line of the job's stdout
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:       # noqa: BLE001
        pass
    sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:       # noqa: BLE001
            pass
    if rank == 0:
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
