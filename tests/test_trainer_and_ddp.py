"""CPU tests of the training-loop surface (reference train_segmentation.py:53-245) and of the data-parallel
gradient exchange (gloo, world_size 2 - the no-GPU stand-in for RCCL).  The loss inside training_step runs on
the oracle-backed double of the C-ABI backend (tests/oracle_backend.py); the product has no CPU path."""
import os
import socket
import warnings

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_backend
from stego_amd import ddp
from stego_amd import modules as M
from stego_amd.train_segmentation import (LitUnsupervisedSegmenter, SyntheticContrastiveDataset, Trainer,
                                          get_class_labels, load_config)
from stego_amd.utils import UnsupervisedMetrics, prep_args

warnings.filterwarnings("ignore", message="DinoFeaturizer")

TINY = ["model_type=vit_tiny", "dino_patch_size=16", "res=32", "batch_size=3", "feature_samples=3", "neg_samples=2",
        "dim=6", "max_steps=2"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_config_keys_and_prep_args():
    cfg = load_config()
    for k in ("feature_samples", "neg_samples", "pointwise", "zero_clamp", "stabalize", "use_salience",
              "pos_intra_shift", "pos_inter_shift", "neg_inter_shift", "pos_intra_weight", "dim", "model_type"):
        assert hasattr(cfg, k), k
    assert (cfg.feature_samples, cfg.neg_samples, cfg.dim, cfg.batch_size) == (11, 5, 70, 16)
    assert (cfg.pos_intra_shift, cfg.pos_inter_shift, cfg.neg_inter_shift) == (0.18, 0.12, 0.46)
    assert prep_args(["prog", "--batch_size", "8", "res=64"]) == ["prog", "batch_size=8", "res=64"]
    assert load_config(overrides=["batch_size=8", "pretrained_weights=~"]).batch_size == 8
    assert get_class_labels("cocostuff27") == 27


def test_module_surface_and_state_dict_keys():
    cfg = load_config(overrides=TINY)
    m = LitUnsupervisedSegmenter(27, cfg)
    keys = set(m.state_dict().keys())
    for k in ("net.cluster1.0.weight", "net.cluster1.0.bias", "net.cluster2.0.weight", "net.cluster2.2.bias",
              "train_cluster_probe.clusters", "cluster_probe.clusters", "linear_probe.weight", "decoder.bias",
              "net.model.cls_token", "net.model.pos_embed", "net.model.patch_embed.proj.weight",
              "net.model.blocks.0.attn.qkv.weight", "net.model.blocks.11.mlp.fc2.bias", "net.model.norm.weight"):
        assert k in keys, k
    assert not any(p.requires_grad for p in m.net.model.parameters())           # frozen backbone (modules.py:30-31)
    x = torch.randn(2, 3, 32, 32)
    feats, code = m.net(x)
    assert tuple(feats.shape) == (2, 192, 2, 2) and tuple(code.shape) == (2, 6, 2, 2)
    assert feats.stride(1) == 1                                                  # channels-last view, as the kernels like it
    assert tuple(m(x).shape) == (2, 6, 2, 2)
    # every name the reference scripts import from modules
    for name in ("LambdaLayer", "DinoFeaturizer", "ResizeAndClassify", "ClusterLookup", "FeaturePyramidNet", "DoubleConv",
                 "norm", "average_norm", "tensor_correlation", "sample", "super_perm", "sample_nonzero_locations",
                 "ContrastiveCorrelationLoss", "Decoder", "NetWithActivations", "ContrastiveCRFLoss"):
        assert hasattr(M, name), name


def test_unsupervised_metrics_hungarian():
    m = UnsupervisedMetrics("t/", 3, 0, True)
    target = torch.tensor([0, 0, 1, 1, 2, 2, 2])
    preds = torch.tensor([2, 2, 0, 0, 1, 1, 0])        # a relabelling of target with one error
    m.update(preds, target)
    out = m.compute()
    assert abs(out["t/Accuracy"] - 100 * 6 / 7) < 1e-6


def _one_step(monkeypatch_backend=True, overrides=TINY, seed=0):
    if monkeypatch_backend:
        M._backend = oracle_backend
    cfg = load_config(overrides=overrides)
    torch.manual_seed(seed)
    m = LitUnsupervisedSegmenter(27, cfg)
    ds = SyntheticContrastiveDataset(6, cfg.res, 27, seed=seed)
    batch = torch.utils.data.default_collate([ds[i] for i in range(cfg.batch_size)])
    return m, batch


def test_training_step_runs_and_updates_only_trainables():
    try:
        m, batch = _one_step()
        before = {k: v.clone() for k, v in m.state_dict().items()}
        loss = m.training_step(batch, 0)
        assert torch.isfinite(loss)
        for k in ("loss/pos_intra", "loss/pos_inter", "loss/neg_inter", "cd/pos_intra", "loss/linear", "loss/cluster",
                  "loss/total"):
            assert k in m.logged
        after = m.state_dict()
        assert not torch.equal(before["net.cluster1.0.weight"], after["net.cluster1.0.weight"])
        assert not torch.equal(before["linear_probe.weight"], after["linear_probe.weight"])
        assert torch.equal(before["net.model.blocks.0.attn.qkv.weight"], after["net.model.blocks.0.attn.qkv.weight"])
        assert m.global_step == 1
    finally:
        from stego_amd import capi
        M._backend = capi


def test_flat_grad_reducer_single_process():
    lin = torch.nn.Linear(4, 3)
    red = ddp.FlatGradReducer(lin.parameters())
    assert red.numel == 15 and lin.weight.grad.data_ptr() == red.flat.data_ptr()
    lin(torch.ones(2, 4)).sum().backward()
    assert torch.allclose(red.flat[:12], torch.full((12,), 2.0))
    red.allreduce_mean()                     # no process group: no-op
    opt = torch.optim.SGD(lin.parameters(), lr=0.1)
    opt.zero_grad(set_to_none=True)
    red.reattach()
    assert lin.weight.grad is not None and lin.weight.grad.data_ptr() == red.flat.data_ptr()
    assert float(red.flat.abs().sum()) == 0.0


def _ddp_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    warnings.filterwarnings("ignore", message="DinoFeaturizer")
    r, w, _ = ddp.init_from_env("gloo")
    assert (r, w) == (rank, world)
    M._backend = oracle_backend
    cfg = load_config(overrides=TINY)
    torch.manual_seed(123 + rank)            # ranks start from DIFFERENT heads: broadcast must fix that
    model = LitUnsupervisedSegmenter(27, cfg)
    reducer = model.setup_distributed()
    ds = SyntheticContrastiveDataset(6, cfg.res, 27, seed=rank)          # per-rank shard
    batch = torch.utils.data.default_collate([ds[i] for i in range(cfg.batch_size)])

    # local gradient of this rank (no collective) for the reference value
    torch.manual_seed(7 + rank)
    rng = torch.get_rng_state()
    model._reducer = None
    for p in model.parameters():
        p.grad = None
    opt_state = [o.state_dict() for o in model.optimizers()]
    params0 = [p.detach().clone() for p in model.parameters()]
    model.training_step(batch, 0)
    local = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
                       for p in model.parameters() if p.requires_grad]).clone()
    with torch.no_grad():                    # rewind parameters / optimizers / RNG
        for p, p0 in zip(model.parameters(), params0):
            p.copy_(p0)
    model._optims = None
    model.global_step = 0
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    expect = torch.stack(gathered).mean(0)

    # the real path: flat bucket + one all-reduce inside manual_backward
    model._reducer = reducer
    for p in reducer.params:
        p.grad = None
    reducer.reattach()
    torch.set_rng_state(rng)
    model.training_step(batch, 0)
    np.save(os.path.join(out_dir, "grad_%d.npy" % rank), reducer.flat.numpy())
    np.save(os.path.join(out_dir, "expect_%d.npy" % rank), expect.numpy())
    np.save(os.path.join(out_dir, "head_%d.npy" % rank), model.net.cluster1[0].weight.detach().numpy())
    dist.destroy_process_group()


def test_two_rank_gloo_gradient_allreduce(tmp_path):
    """world_size 2 on CPU: after one step every rank holds the MEAN of the per-rank gradients (what Lightning DDP
    gives the reference, per-rank loss statistics included - SURVEY.md 8(e)) and identical parameters."""
    port = _free_port()
    mp.spawn(_ddp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g0, g1 = np.load(tmp_path / "grad_0.npy"), np.load(tmp_path / "grad_1.npy")
    e0 = np.load(tmp_path / "expect_0.npy")
    np.testing.assert_allclose(g0, g1, rtol=0, atol=0)
    np.testing.assert_allclose(g0, e0, rtol=1e-5, atol=1e-8)
    assert np.abs(e0).sum() > 0
    np.testing.assert_array_equal(np.load(tmp_path / "head_0.npy"), np.load(tmp_path / "head_1.npy"))


def test_trainer_fit_history_cpu():
    try:
        M._backend = oracle_backend
        cfg = load_config(overrides=TINY)
        torch.manual_seed(0)
        model = LitUnsupervisedSegmenter(27, cfg)
        ds = SyntheticContrastiveDataset(6, cfg.res, 27)
        loader = torch.utils.data.DataLoader(ds, cfg.batch_size, shuffle=False, drop_last=True)
        hist = Trainer(cfg.max_steps, device=torch.device("cpu"), log_every=100).fit(model, loader)
        assert len(hist) == cfg.max_steps and all(np.isfinite(hist))
    finally:
        from stego_amd import capi
        M._backend = capi


def test_checkpoint_roundtrip_in_the_lightning_layout(tmp_path):
    """`.ckpt` = {state_dict, hyper_parameters: {n_classes, cfg}, ...} as written by Lightning for the
    reference (train_segmentation.py:106,:487) and read by eval_segmentation.py:67."""
    cfg = load_config(overrides=["model_type=vit_tiny", "dino_patch_size=16", "res=32", "batch_size=2", "dim=12"])
    torch.manual_seed(1)
    m = LitUnsupervisedSegmenter(5, cfg).cpu()
    m.optimizers()
    m.global_step = 7
    path = tmp_path / "model.ckpt"
    m.save_checkpoint(str(path), epoch=3)
    raw = torch.load(str(path), map_location="cpu", weights_only=False)
    assert {"state_dict", "hyper_parameters", "epoch", "global_step", "optimizer_states"} <= set(raw)
    assert raw["hyper_parameters"]["n_classes"] == 5 and raw["hyper_parameters"]["cfg"]["dim"] == 12
    assert "net.cluster1.0.weight" in raw["state_dict"] and "cluster_probe.clusters" in raw["state_dict"]
    m2 = LitUnsupervisedSegmenter.load_from_checkpoint(str(path))
    assert m2.n_classes == 5 and m2.cfg.dim == 12 and m2.global_step == 7
    for (k1, v1), (k2, v2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1.cpu(), v2.cpu()), k1
    # a reference-style checkpoint: cfg as a mapping object, extra torchmetrics-like buffers in the state dict
    raw["state_dict"]["cluster_metrics.stats"] = torch.zeros(3)
    torch.save(raw, str(path))
    m3 = LitUnsupervisedSegmenter.load_from_checkpoint(str(path), strict=False, native_backbone=False)
    assert m3.cfg.native_backbone is False and "cluster_metrics.stats" in m3.load_result.unexpected_keys


def test_neighbour_file_name_and_format(tmp_path):
    from stego_amd import precompute_knns as P
    name = P.nns_filename("vit_small", "cocostuff27", "train", "five", 224)
    assert name == "nns_vit_small_cocostuff27_train_five_224.npz"          # precompute_knns.py:66-67, data.py:503-511
    nns = torch.arange(12, dtype=torch.int64).reshape(4, 3)
    P.save_nns(str(tmp_path / name), nns)
    assert torch.equal(P.load_nns(str(tmp_path / name)), nns)
    import numpy as np
    assert list(np.load(str(tmp_path / name)).keys()) == ["nns"]
    # both directions against the reference's own lines: its reader (data.py:509-510: ``loaded = np.load(feature_cache_file);
    # self.nns = loaded["nns"]``) on our file, our reader on what its writer leaves (precompute_knns.py:96:
    # ``np.savez_compressed(feature_cache_file, nns=nearest_neighbors.numpy())``), and its file-name expression (data.py:503-504)
    loaded = np.load(str(tmp_path / name))
    assert np.array_equal(loaded["nns"], nns.numpy()) and loaded["nns"].dtype == np.int64
    ref_name = "nns_{}_{}_{}_{}_{}.npz".format("vit_small", "cocostuff27", "train", "five", 224)
    assert ref_name == name
    np.savez_compressed(str(tmp_path / "ref_written.npz"), nns=nns.numpy())
    assert torch.equal(P.load_nns(str(tmp_path / "ref_written.npz")), nns)
    assert open(str(tmp_path / "ref_written.npz"), "rb").read() == open(str(tmp_path / name), "rb").read()     # byte-identical archives


def test_token_cache_serves_the_frozen_backbone_from_memory():
    """featurizers.TokenCache: the second visit of a dataset index does not run the backbone; values = the direct
    forward up to the fp16 storage; the trainer wires it through batch['ind'] / batch['ind_pos']."""
    cfg = load_config(overrides=["model_type=vit_tiny", "dino_patch_size=16", "res=32", "batch_size=4", "dim=12",
                                 "feature_samples=3", "neg_samples=1", "cache_backbone_tokens=True", "max_steps=4", "dropout=False"])
    torch.manual_seed(2)
    m = LitUnsupervisedSegmenter(5, cfg).cpu()
    net = m.net
    calls = []
    orig = net._tokens
    net._tokens = lambda img, n: (calls.append(img.shape[0]), orig(img, n))[1]
    cache = net.enable_token_cache(8, (32, 32), torch.device("cpu"))
    net.eval()
    img = torch.randn(8, 3, 32, 32)
    idx = torch.arange(8)
    f_direct, _ = net(img)
    assert calls == [8]
    f1, _ = net(img[:5], cache_index=idx[:5])               # 5 misses
    f2, _ = net(img[2:], cache_index=idx[2:])               # rows 2..4 hit, 5..7 miss
    assert calls == [8, 5, 3] and cache.misses == 8 and cache.complete
    f3, _ = net(img, cache_index=idx)                       # all hits: the backbone is not called
    assert calls == [8, 5, 3]
    assert torch.allclose(f3, f_direct, rtol=2e-3, atol=2e-3) and torch.allclose(f1, f_direct[:5], rtol=2e-3, atol=2e-3)
    assert f3.stride(1) == 1                                # still the channels-last view the loss kernels want


def test_token_cache_positive_has_its_own_rows_and_is_dropped_with_the_weights():
    """Trainer-level: with cfg.cache_backbone_tokens the positive's cached features equal its uncached features (it must not
    be served the anchor's row), random-crop loaders are refused, and loading new weights empties the cache."""
    from stego_amd.train_segmentation import SyntheticContrastiveDataset, Trainer
    cfg = load_config(overrides=["model_type=vit_tiny", "dino_patch_size=16", "res=32", "batch_size=4", "dim=12",
                                 "feature_samples=3", "neg_samples=1", "cache_backbone_tokens=True", "max_steps=3", "dropout=False"])
    torch.manual_seed(3)
    m = LitUnsupervisedSegmenter(5, cfg).cpu()
    ds = SyntheticContrastiveDataset(8, 32, 5, seed=1)
    item = ds[3]
    assert item["ind_pos"] == 8 + 3 and ds.n_cache_items == 16
    loader = torch.utils.data.DataLoader(ds, 4, shuffle=False, drop_last=True)
    tr = Trainer(3, device=torch.device("cpu"))
    M._backend = oracle_backend                 # the loss itself has no CPU path: the oracle-backed double stands in
    try:
        tr.fit(m, loader)
    finally:
        from stego_amd import capi
        M._backend = capi
    cache = m.net.token_cache
    assert cache is not None and cache.tokens.shape[0] == 16
    m.net.eval()
    batch = next(iter(loader))
    with torch.no_grad():
        cached_pos, _ = m.net(batch["img_pos"], cache_index=batch["ind_pos"])
        direct_pos, _ = m.net(batch["img_pos"])
        direct_anchor, _ = m.net(batch["img"])
    assert torch.allclose(cached_pos, direct_pos, rtol=2e-3, atol=2e-3)
    assert not torch.allclose(cached_pos, direct_anchor, rtol=2e-3, atol=2e-3)
    # new weights -> stale tokens are dropped
    assert bool(cache.filled.any())
    m.net.load_state_dict(m.net.state_dict())
    assert not bool(cache.filled.any()) and not cache.complete

    class Recrop(SyntheticContrastiveDataset):
        deterministic_items = False
    with pytest.raises(ValueError, match="fixed per index"):
        Trainer(1, device=torch.device("cpu")).fit(LitUnsupervisedSegmenter(5, cfg).cpu(),
                                                    torch.utils.data.DataLoader(Recrop(8, 32, 5), 4, drop_last=True))
    with pytest.raises(ValueError, match="empty loader"):
        Trainer(1, device=torch.device("cpu")).fit(LitUnsupervisedSegmenter(5, cfg).cpu(),
                                                    torch.utils.data.DataLoader(SyntheticContrastiveDataset(2, 32, 5), 4, drop_last=True))


def test_bench_multi_rank_protocol_two_processes_gloo():
    """bench.py's own N > 1 path under torch.distributed.run (the way the driver launches it): rendezvous on 127.0.0.1, the
    per-step gradient exchange through the trainer's FlatGradReducer, barriers, MAX-over-ranks timing, ONE JSON line from
    rank 0 as the last line of stdout.  --dry-run-cpu replaces the kernels by no-ops (no GPU in CI); the record says so."""
    import json, subprocess, sys, socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--dry-run-cpu"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    last = [ln for ln in out.stdout.strip().splitlines() if ln.strip()][-1]
    rec = json.loads(last)
    assert rec["n_gpus"] == 2 and rec["steps"] == 6 and rec["warmup"] == 2 and rec["dry_run"] is True and rec["value"] is None
    assert rec["scaling"] == "weak" and rec["config"]["global_batch"] == 64 and rec["config"]["parallelism"] == "dp2"
    chk = rec["collective_check"]
    assert chk["grad_mean_after_allreduce"] == chk["expected"] == 1.5
    assert chk["bucket_numel"] == 384 * 70 + 70 + 384 * 384 + 384 + 384 * 70 + 70 + 70 * 27 + 27 + 27 * 70
    coll = rec["config"]["collective"]
    assert coll["what"].startswith("all_reduce(")
    # (round 5) a scaling run explains itself: every rank's own ms per step and the all-reduce timed on its own
    assert len(coll["ms_per_step_by_rank"]) == 2 and coll["ms_per_step_rank_min"] <= coll["ms_per_step_rank_max"] and coll["allreduce_us"] > 0


def test_shapes_outside_the_fused_kernels_are_announced_at_construction_with_the_cfg_key_named():
    """cfg.dim > 128 / cfg.feature_samples > 16 are valid in the reference (train_config.yml:39,51 are free): they run
    on the generic path (ContrastiveCorrelationLoss.generic_forward) - the constructor says so, naming the key; so does 72 < dim <= 128 on
    feature maps the single-launch kernel does not take (the feature-pyramid arch: 2048 channels): valid in the reference, it runs on the
    generic path too.  (Odd code dimensions and vit_tiny's 192 channels are served by the single-launch kernel since round 4,
    feature_samples 12 .. 16 by the multi-launch kernels of csrc/corr_wide.hip since round 5: no announcement.)"""
    import warnings
    cfg = load_config(overrides=["model_type=vit_tiny", "dino_patch_size=16", "res=32", "feature_samples=12"])
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        LitUnsupervisedSegmenter(5, cfg)
    assert not [w for w in rec if "feature_samples" in str(w.message)]
    for ov, key in ((["dim=130"], "cfg.dim=130"), (["feature_samples=17"], "cfg.feature_samples=17")):
        cfg = load_config(overrides=["model_type=vit_tiny", "dino_patch_size=16", "res=32"] + ov)
        with pytest.warns(UserWarning, match=key):
            LitUnsupervisedSegmenter(5, cfg)
    cfg = load_config(overrides=["model_type=vit_tiny", "dino_patch_size=16", "res=32", "dim=99"])
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        LitUnsupervisedSegmenter(5, cfg)
    assert not [w for w in rec if "cfg.dim=99" in str(w.message)]
    cfg = load_config(overrides=["arch=feature-pyramid", "model_type=resnet50", "granularity=2", "res=64", "allow_random_trunk=True", "dim=99"])
    with pytest.warns(UserWarning, match="cfg.dim=99"):
        LitUnsupervisedSegmenter(5, cfg)


def test_feature_pyramid_arch_builds_through_the_segmenter():
    """cfg.arch == 'feature-pyramid' (train_segmentation.py:65-67): load_model's ResNet-50 trunk (torchvision-compatible
    parameter names; random init when allowed, FileNotFoundError naming the checkpoint otherwise) cut by FeaturePyramidNet."""
    from stego_amd import trunks
    ov = ["arch=feature-pyramid", "model_type=resnet50", "granularity=2", "dim=16", "res=64", "allow_random_trunk=True"]
    m = LitUnsupervisedSegmenter(5, load_config(overrides=ov)).cpu().eval()
    with torch.no_grad():
        feats, code = m.net(torch.randn(2, 3, 224, 224))
    assert feats.shape == (2, 2048, 7, 7) and code.shape == (2, 16, 56, 56)
    keys = trunks.ResNet50().state_dict().keys()
    assert "layer3.5.conv3.weight" in keys and "layer1.0.downsample.1.running_var" in keys and "fc.bias" in keys
    assert sum(p.numel() for p in trunks.ResNet50().parameters()) == 25557032          # torchvision's resnet50
    with pytest.raises(FileNotFoundError, match="resnet50-0676ba61.pth"):
        LitUnsupervisedSegmenter(5, load_config(overrides=ov[:-1]))
    with pytest.raises(ValueError, match="No model"):
        trunks.load_model("vgg11", ".")


def test_trainer_runs_validation_and_writes_a_loadable_checkpoint(tmp_path):
    """Trainer.fit: validation every val_check_interval steps (train_segmentation.py:247-330, :489) and a Lightning-layout
    checkpoint that LitUnsupervisedSegmenter.load_from_checkpoint reads back (the ModelCheckpoint callback of :482-486)."""
    from stego_amd.train_segmentation import SyntheticContrastiveDataset, Trainer
    cfg = load_config(overrides=["model_type=vit_tiny", "dino_patch_size=16", "res=32", "batch_size=4", "dim=12",
                                 "feature_samples=3", "neg_samples=1", "max_steps=4", "dropout=False", "n_images=2"])
    torch.manual_seed(4)
    m = LitUnsupervisedSegmenter(5, cfg).cpu()
    ds = SyntheticContrastiveDataset(8, 32, 5, seed=2)
    loader = torch.utils.data.DataLoader(ds, 4, shuffle=False, drop_last=True)
    ck = str(tmp_path / "last.ckpt")
    tr = Trainer(4, device=torch.device("cpu"), val_loader=torch.utils.data.DataLoader(ds, 4), val_check_interval=2, checkpoint_path=ck)
    M._backend = oracle_backend
    try:
        hist = tr.fit(m, loader)
    finally:
        from stego_amd import capi
        M._backend = capi
    assert len(hist) == 4 and len(tr.val_history) == 2
    assert any(k.endswith("mIoU") or "Accuracy" in k for k in tr.val_history[-1]), tr.val_history[-1].keys()
    assert os.path.exists(ck)
    m2 = LitUnsupervisedSegmenter.load_from_checkpoint(ck)
    for (k1, v1), (k2, v2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2), k1


def test_cropped_dataset_tree_round_trip(tmp_path):
    """crop_datasets.py:76-123 / data.py:370-400: five crops per image under cropped/{ds}_{type}_crop_{ratio}/{img,label}/{split}/
    {5 i + c}.{jpg,png}; labels stored + 1 (PNG, lossless), unlabelled -> mask."""
    from stego_amd import data as D
    g = torch.Generator().manual_seed(3)
    items = []
    for _ in range(3):
        yy, xx = torch.meshgrid(torch.linspace(0, 1, 40), torch.linspace(0, 1, 60), indexing="ij")
        ph = torch.rand(3, 1, 1, generator=g)
        img = (0.5 + 0.4 * torch.sin(6.0 * (xx + ph)) * torch.cos(4.0 * (yy + ph))).clamp(0, 1)      # smooth: JPEG keeps it
        label = torch.randint(-1, 27, (40, 60), generator=g)
        items.append((img, label))
    n = D.write_cropped(str(tmp_path), "cocostuff27", "five", 0.5, "val", items)
    assert n == 15
    base = tmp_path / "cropped" / "cocostuff27_five_crop_0.5"
    assert sorted(os.listdir(base / "img" / "val"), key=lambda s: int(s.split(".")[0])) == ["%d.jpg" % i for i in range(15)]
    assert sorted(os.listdir(base / "label" / "val"), key=lambda s: int(s.split(".")[0])) == ["%d.png" % i for i in range(15)]
    ds = D.CroppedDataset(str(tmp_path), "cocostuff27", "five", 0.5, "val")
    assert len(ds) == 15
    boxes = D.five_crop_boxes(40, 60, 20, 30)
    assert boxes == [(0, 0), (0, 30), (20, 0), (20, 30), (10, 15)]
    for idx in (0, 4, 7, 14):
        image, target, mask = ds[idx]
        src_img, src_label = items[idx // 5]
        t, l = boxes[idx % 5]
        assert image.shape == (3, 20, 30) and target.shape == (20, 30) and mask.shape == (1, 20, 30)
        assert torch.equal(target, src_label[t:t + 20, l:l + 30])                       # PNG: exact
        assert torch.equal(mask.squeeze(0), src_label[t:t + 20, l:l + 30] == -1)
        assert float((image - src_img[:, t:t + 20, l:l + 30]).abs().mean()) < 0.02     # JPEG: lossy
    D.write_cropped(str(tmp_path), "cocostuff27", "random", 0.5, "train", items[:1])
    assert len(D.CroppedDataset(str(tmp_path), "cocostuff27", "random", 0.5, "train")) == 5
    with pytest.raises(ValueError, match="Unknown crop type"):
        D.write_cropped(str(tmp_path), "x", "center", 0.5, "val", items[:1])


def test_trainer_shards_a_plain_loader_across_ranks(monkeypatch):
    """a real dataset without a DistributedSampler gave every rank the same batches.  Trainer.fit rebuilds such a loader with
    one (as Lightning does for the reference); a dataset that is already per-rank, a single process and a loader that has one stay."""
    from stego_amd import train_segmentation as T

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 20

        def __getitem__(self, i):
            return {"ind": i}

    seen = {}
    for rank in (0, 1):
        tr = T.Trainer(max_steps=1)
        tr.rank, tr.world = rank, 2
        ld = tr._shard_loader(torch.utils.data.DataLoader(DS(), batch_size=5, shuffle=True, drop_last=True))
        assert isinstance(ld.sampler, torch.utils.data.distributed.DistributedSampler)
        ld.sampler.set_epoch(0)
        seen[rank] = sorted(int(i) for b in ld for i in b["ind"])
        assert len(seen[rank]) == 10
    assert not set(seen[0]) & set(seen[1]) and sorted(seen[0] + seen[1]) == list(range(20))
    tr = T.Trainer(max_steps=1)
    tr.rank, tr.world = 0, 1
    plain = torch.utils.data.DataLoader(DS(), batch_size=5)
    assert tr._shard_loader(plain) is plain
    tr.world = 2
    syn = torch.utils.data.DataLoader(T.SyntheticContrastiveDataset(8, 32, 5, seed=0), batch_size=4)
    assert tr._shard_loader(syn) is syn



# ------------------------------------------------------------------ on-disk formats against artefacts the REFERENCE's code produced
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("crop_type", ["five", "random"])
def test_cropped_tree_written_by_the_reference_is_read_and_reproduced(crop_type, tmp_path):
    """tests/golden/cropped_ref/ was written by the unmodified RandomCropComputer.__getitem__ / random_crops / five_crops of the
    reference (oracle/make_format_golden.py drives /root/reference/src/crop_datasets.py:14-123).  stego_amd.data.CroppedDataset reads
    it with the reference reader's semantics (data.py:370-400: target - 1, mask = target == -1), and stego_amd.data.write_cropped
    writes the same files from the same source items (same boxes, same uint8 conversions; labels bit for bit, JPEGs pixel for pixel)."""
    from PIL import Image
    from stego_amd import data as D
    root = os.path.join(GOLD, "cropped_ref")
    src = np.load(os.path.join(root, "source_items.npz"))
    items = [(torch.from_numpy(src["img%d" % i]), torch.from_numpy(src["label%d" % i])) for i in range(2)]
    ds = D.CroppedDataset(root, "toyset", crop_type, 0.5, "train")
    assert len(ds) == 10
    n = D.write_cropped(str(tmp_path), "toyset", crop_type, 0.5, "train", items)
    assert n == 10
    mine = D.crop_dir(str(tmp_path), "toyset", crop_type, 0.5)
    for idx in range(10):
        image, target, mask = ds[idx]
        item, crop_num = divmod(idx, 5)
        img, label = items[item]
        H, W = label.shape
        ch, cw = int(H * 0.5), int(W * 0.5)
        boxes = D.five_crop_boxes(H, W, ch, cw) if crop_type == "five" else D.random_crop_boxes(H, W, ch, cw, item)
        t, l = boxes[crop_num]
        assert torch.equal(target, label[t:t + ch, l:l + cw])                       # labels survive the + 1 / - 1 round trip exactly
        assert torch.equal(mask.squeeze(0), label[t:t + ch, l:l + cw] == -1)
        assert image.shape == (3, ch, cw) and float((image - img[:, t:t + ch, l:l + cw]).abs().mean()) < 0.3    # (JPEG of white noise)
        for sub, ext in (("img", "jpg"), ("label", "png")):
            a = np.asarray(Image.open(os.path.join(ds.root, sub, "train", "%d.%s" % (idx, ext))))
            b = np.asarray(Image.open(os.path.join(mine, sub, "train", "%d.%s" % (idx, ext))))
            np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("arch", ["vit_small", "vit_base"])
def test_state_dict_is_key_for_key_the_reference_lightning_checkpoint(arch, tmp_path):
    """tests/golden/ref_ckpt_manifest.json lists the state_dict of the reference's LitUnsupervisedSegmenter (keys, shapes, dtypes taken
    from the reference's own DinoFeaturizer / ClusterLookup classes, train_segmentation.py:53-106).  This build's module tree must have
    exactly those entries, and a checkpoint in Lightning 1.2's layout holding them loads with strict=True."""
    import json
    with open(os.path.join(GOLD, "ref_ckpt_manifest.json")) as f:
        man = json.load(f)
    m = man[arch]
    cfg = load_config(overrides=["model_type=%s" % arch, "dino_patch_size=8", "dim=%d" % m["dim"], "extra_clusters=%d" % m["extra_clusters"],
                                 "native_backbone=False"])
    model = LitUnsupervisedSegmenter(m["n_classes"], cfg).cpu()
    mine = {k: [list(v.shape), str(v.dtype)] for k, v in model.state_dict().items()}
    assert mine == m["state_dict"]
    g = torch.Generator().manual_seed(3)
    sd = {k: torch.randn(shape, generator=g).to(getattr(torch, dt.split(".")[1])) for k, (shape, dt) in m["state_dict"].items()}
    ck = {k: None for k in man["lightning_1_2_top_level_keys"]}
    ck.update({"epoch": 2, "global_step": 1234, "pytorch-lightning_version": "1.2.10", "callbacks": {}, "optimizer_states": [],
               "lr_schedulers": [], "state_dict": sd, "hparams_name": "kwargs",
               "hyper_parameters": {"n_classes": m["n_classes"], "cfg": {k: v for k, v in vars(cfg).items()}}})
    assert sorted(ck["hyper_parameters"]) == sorted(man["hyper_parameters_keys"])
    path = str(tmp_path / "ref_layout.ckpt")
    torch.save(ck, path)
    loaded = LitUnsupervisedSegmenter.load_from_checkpoint(path, strict=True)
    assert loaded.global_step == 1234 and loaded.n_classes == m["n_classes"]
    for k, v in loaded.state_dict().items():
        assert torch.equal(v, sd[k]), k


def test_linear_probe_loss_in_its_spatial_form_is_the_reference_s_masked_mean():
    """training_step's linear-probe loss: the reference flattens the upsampled logits to [pixels, classes], boolean-indexes the valid
    pixels and takes nn.CrossEntropyLoss (train_segmentation.py:199-203); the trainer here calls F.cross_entropy on [B, C, H, W] with the
    invalid labels as ignore_index.  Same value and same gradient, labels outside [0, n_classes) included."""
    import torch.nn.functional as F
    torch.manual_seed(3)
    n_classes = 27
    logits = torch.randn(3, n_classes, 20, 24, requires_grad=True)
    label = torch.randint(-1, n_classes + 2, (3, 20, 24))
    flat = label.reshape(-1)
    mask = (flat >= 0) & (flat < n_classes)
    ref = torch.nn.CrossEntropyLoss()(logits.permute(0, 2, 3, 1).reshape(-1, n_classes)[mask], flat[mask]).mean()
    g_ref, = torch.autograd.grad(ref, logits)
    valid = (label >= 0) & (label < n_classes)
    new = F.cross_entropy(logits, torch.where(valid, label, torch.full_like(label, -100)), ignore_index=-100)
    g_new, = torch.autograd.grad(new, logits)
    assert abs(float(ref) - float(new)) <= 1e-6 * abs(float(ref))
    assert torch.allclose(g_ref, g_new, rtol=1e-5, atol=1e-9)
