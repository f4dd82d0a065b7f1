"""PL-free mirror of the reference's ``src/train_segmentation.py``: ``LitUnsupervisedSegmenter`` with the same
constructor, attributes, state-dict keys and ``training_step`` arithmetic (:53-245,373-383), a small
``Trainer`` (the reference delegates to pytorch_lightning, which is not in this image), the yaml+override
config loader (hydra is absent) and a synthetic stand-in for ``ContrastiveSegDataset`` (no datasets offline).

The correspondence loss inside ``training_step`` is the HIP path (stego_amd.modules); everything else is stock
PyTorch-ROCm.  Data parallelism: one process per GPU, batch sharded, ONE flat RCCL all-reduce of the
trainable (head + probe) gradients per step (stego_amd.ddp) - the backbone is frozen, nothing inside the loss
is synchronised.

    python -m stego_amd.train_segmentation max_steps=20 batch_size=8            # synthetic data, 1 GPU
    torchrun --standalone --nproc-per-node 8 -m stego_amd.train_segmentation max_steps=20
"""
import os
import sys
import types
from os.path import dirname, join

import torch
import torch.nn as nn
import torch.nn.functional as F
import yaml

from . import ddp
from .featurizers import ClusterLookup, ContrastiveCRFLoss, DinoFeaturizer, FeaturePyramidNet
from .modules import ContrastiveCorrelationLoss, norm, sample
from .utils import UnsupervisedMetrics, one_hot_feats, prep_args, resize


def get_class_labels(dataset_name):
    """n_classes per dataset as the reference sets them (train_segmentation.py:33-50, data.py)."""
    if dataset_name.startswith("cityscapes"):
        return 27
    if dataset_name == "cocostuff27":
        return 27
    if dataset_name == "cocostuff3":
        return 3
    if dataset_name == "potsdam":
        return 3
    raise ValueError("Unknown Dataset {}".format(dataset_name))


def load_config(path=None, overrides=()):
    """yaml defaults + `key=value` overrides -> attribute-style cfg (what hydra/OmegaConf gives the reference)."""
    path = path or join(dirname(__file__), "configs", "train_config.yml")
    with open(path) as f:
        d = yaml.safe_load(f)
    for ov in overrides:
        k, v = ov.split("=", 1)
        d[k] = yaml.safe_load(v)
    return types.SimpleNamespace(**d)


MAX_CODE_DIM = 128               # include/stego_corr.h "Limits of this build" (above 72: channels-last ViT-width maps)
MAX_FEATURE_SAMPLES = 11
MAX_FEATURE_SAMPLES_WIDE = 16    # 12 .. 16: csrc/corr_wide.hip behind the same entry points (a few launches instead of one)
MAX_CODE_DIM_WIDE = 128
MAX_CODE_DIM_ANY_PATH = 72       # above it: the single-launch forward only (its conditions are checked in __init__)


class LitUnsupervisedSegmenter(nn.Module):
    def __init__(self, n_classes, cfg):
        super().__init__()
        self.cfg = cfg
        self.n_classes = n_classes
        dim = cfg.dim if cfg.continuous else n_classes
        if cfg.arch == "feature-pyramid":           # train_segmentation.py:65-67
            from .trunks import load_model
            data_dir = os.path.join(getattr(cfg, "output_root", "."), "data")
            cut_model = load_model(cfg.model_type, data_dir, allow_random_init=getattr(cfg, "allow_random_trunk", False))
            self.net = FeaturePyramidNet(cfg.granularity, cut_model, dim, cfg.continuous)
        elif cfg.arch == "dino":
            self.net = DinoFeaturizer(dim, cfg)
        else:
            raise ValueError("Unknown arch {}".format(cfg.arch))
        self.train_cluster_probe = ClusterLookup(dim, n_classes)
        self.cluster_probe = ClusterLookup(dim, n_classes + cfg.extra_clusters)
        self.linear_probe = nn.Conv2d(dim, n_classes, (1, 1))
        self.decoder = nn.Conv2d(dim, self.net.n_feats, (1, 1))
        self.cluster_metrics = UnsupervisedMetrics("test/cluster/", n_classes, cfg.extra_clusters, True)
        self.linear_metrics = UnsupervisedMetrics("test/linear/", n_classes, 0, False)
        self.test_cluster_metrics = UnsupervisedMetrics("final/cluster/", n_classes, cfg.extra_clusters, True)
        self.test_linear_metrics = UnsupervisedMetrics("final/linear/", n_classes, 0, False)
        self.linear_probe_loss_fn = nn.CrossEntropyLoss()
        self.crf_loss_fn = ContrastiveCRFLoss(cfg.crf_samples, cfg.alpha, cfg.beta, cfg.gamma, cfg.w1, cfg.w2, cfg.shift)
        self.contrastive_corr_loss_fn = ContrastiveCorrelationLoss(cfg)
        # limits of this build of libstego_corr (include/stego_corr.h): fail here, with the cfg keys named, not with
        # STEGO_ERR_UNSUPPORTED inside the first training_step
        if cfg.correspondence_weight > 0:
            if dim > MAX_CODE_DIM:
                import warnings
                warnings.warn("cfg.dim=%d: the fused loss kernels cover code dimensions up to %d (train_config.yml:39 ships 70; "
                              "include/stego_corr.h); this configuration runs on ContrastiveCorrelationLoss.generic_forward: same "
                              "results, several times slower" % (dim, MAX_CODE_DIM))
            if cfg.feature_samples > MAX_FEATURE_SAMPLES_WIDE or (cfg.feature_samples > MAX_FEATURE_SAMPLES and dim > MAX_CODE_DIM_WIDE):
                import warnings
                warnings.warn("cfg.feature_samples=%d, cfg.dim=%d: the single-launch loss kernels cover S <= %d (train_config.yml:51 ships 11), the "
                              "multi-launch kernels S <= %d with dim <= %d; this configuration runs on ContrastiveCorrelationLoss.generic_forward "
                              "(the same native samplers and correlation kernels composed in Python): same results, slower and host-bound"
                              % (cfg.feature_samples, dim, MAX_FEATURE_SAMPLES, MAX_FEATURE_SAMPLES_WIDE, MAX_CODE_DIM_WIDE))
            # (feature_samples 12 .. 16 run on csrc/corr_wide.hip, which takes any layout with dim <= 128: nothing to warn about - ADVICE round 5)
            if MAX_CODE_DIM_ANY_PATH < dim <= MAX_CODE_DIM and cfg.feature_samples <= MAX_FEATURE_SAMPLES:
                # 72 < dim <= 128 exists on the single-launch forward only (plan_fwd / fused_supported, csrc/c_api.hip,
                # csrc/corr_fused.hip); what it does not take runs on generic_forward like dim > 128 does (fused_kernels_cover decides per
                # call, from the tensors): the same conditions here, with the cfg keys named, as a warning
                from .modules import _pair_set_bound
                why = []
                if cfg.arch != "dino":
                    why.append("cfg.arch is not 'dino' (channels-last feature maps of width 192 / 384 / 768)")
                elif getattr(self.net, "n_feats", 384) not in (192, 384, 768):
                    why.append("cfg.model_type=%s has feature width %d (the single-launch forward takes 192 / 384 / 768)"
                               % (getattr(cfg, "model_type", "?"), self.net.n_feats))
                if cfg.batch_size > _pair_set_bound():
                    why.append("cfg.batch_size = %d exceeds the %d compute units (the tiles of one pair-set run at the same time)"
                               % (cfg.batch_size, _pair_set_bound()))
                if why:
                    import warnings
                    warnings.warn("cfg.dim=%d: code dimensions above %d run on the single-launch forward only, and %s: this configuration "
                                  "runs on ContrastiveCorrelationLoss.generic_forward: same results, several times slower"
                                  % (dim, MAX_CODE_DIM_ANY_PATH, "; ".join(why)))
        for p in self.contrastive_corr_loss_fn.parameters():
            p.requires_grad = False
        self.automatic_optimization = False
        self.val_steps = 0
        self.global_step = 0
        self.logged = {}
        self._optims = None
        self._reducer = None

    # ---- checkpoints in the layout Lightning writes for the reference (train_segmentation.py:106 save_hyperparameters,
    #      :487 ModelCheckpoint) and that eval_segmentation.py:67 / demo_segmentation.py:41 read back with
    #      LitUnsupervisedSegmenter.load_from_checkpoint: a torch-pickled dict with `state_dict` (same keys: the module tree
    #      above is the reference's) and `hyper_parameters = {n_classes, cfg}`
    def checkpoint_dict(self, epoch=0):
        cfg = {k: v for k, v in vars(self.cfg).items()} if not isinstance(self.cfg, dict) else dict(self.cfg)
        ck = {"epoch": int(epoch), "global_step": int(self.global_step), "pytorch-lightning_version": "1.2.10",
              "state_dict": self.state_dict(), "hyper_parameters": {"n_classes": self.n_classes, "cfg": cfg}}
        if self._optims is not None:
            ck["optimizer_states"] = [o.state_dict() for o in self._optims]
        return ck

    def save_checkpoint(self, path, epoch=0):
        torch.save(self.checkpoint_dict(epoch), path)

    @classmethod
    def load_from_checkpoint(cls, path, map_location="cpu", strict=True, **cfg_overrides):
        """Accepts this build's checkpoints and the reference's Lightning ones (whose `cfg` is an OmegaConf DictConfig
        or a plain dict: anything with .items()).  Keys this build does not have (e.g. torchmetrics buffers) are
        reported, not fatal, with strict=False."""
        ck = torch.load(path, map_location=map_location, weights_only=False)
        hp = ck["hyper_parameters"]
        cfg = hp["cfg"]
        cfg = types.SimpleNamespace(**{k: v for k, v in (cfg.items() if hasattr(cfg, "items") else vars(cfg).items())})
        for k, v in cfg_overrides.items():
            setattr(cfg, k, v)
        model = cls(int(hp["n_classes"]), cfg)
        result = model.load_state_dict(ck["state_dict"], strict=strict)
        model.global_step = int(ck.get("global_step", 0))
        if "optimizer_states" in ck:
            for o, sd in zip(model.optimizers(), ck["optimizer_states"]):
                o.load_state_dict(sd)
        model.load_result = result
        return model

    # ---- the slice of the LightningModule protocol the reference uses
    def forward(self, x):
        return self.net(x)[1]

    def log(self, name, value, **_):
        self.logged[name] = value.detach() if torch.is_tensor(value) else value

    def configure_optimizers(self):
        main_params = list(self.net.parameters())
        if self.cfg.rec_weight > 0:
            main_params.extend(self.decoder.parameters())
        net_optim = torch.optim.Adam(main_params, lr=self.cfg.lr)
        linear_probe_optim = torch.optim.Adam(list(self.linear_probe.parameters()), lr=5e-3)
        cluster_probe_optim = torch.optim.Adam(list(self.cluster_probe.parameters()), lr=5e-3)
        return net_optim, linear_probe_optim, cluster_probe_optim

    def optimizers(self):
        if self._optims is None:
            self._optims = list(self.configure_optimizers())
        return self._optims

    def setup_distributed(self):
        """Flat gradient bucket over every trainable parameter + startup broadcast (what DDP does on wrap)."""
        self._reducer = ddp.FlatGradReducer([p for p in self.parameters() if p.requires_grad])
        self._reducer.broadcast_params(0)
        return self._reducer

    def manual_backward(self, loss):
        """loss.backward() + the DDP gradient exchange (Lightning's manual_backward under accelerator='ddp',
        train_segmentation.py:227): ONE asynchronous averaged all-reduce of the flat bucket; wait_gradients() joins it."""
        loss.backward()
        if self._reducer is not None:
            self._reducer.reattach()
            # (force_collective: run the collective on a process group of ONE - the single-GPU dress rehearsal of the N > 1 path)
            self._grad_work = self._reducer.allreduce_mean(async_op=True, single_rank_ok=getattr(self, "force_collective", False))

    def wait_gradients(self):
        work, self._grad_work = getattr(self, "_grad_work", None), None
        if work is not None:
            work.wait()                 # stream dependency on RCCL's stream; the host keeps enqueueing

    def training_step(self, batch, batch_idx):
        net_optim, linear_probe_optim, cluster_probe_optim = self.optimizers()
        if self._reducer is not None:
            self._reducer.zero_grad()
        else:
            net_optim.zero_grad(); linear_probe_optim.zero_grad(); cluster_probe_optim.zero_grad()
        cfg = self.cfg
        with torch.no_grad():
            img, img_pos, label = batch["img"], batch["img_pos"], batch["label"]
        # cfg.cache_backbone_tokens: img / img_pos of a dataset index are the same pixels every epoch -> their frozen-backbone
        # tokens come from the HBM table after the first epoch (featurizers.TokenCache)
        caching = getattr(cfg, "cache_backbone_tokens", False) and getattr(self.net, "token_cache", None) is not None
        feats, code = self.net(img, cache_index=batch["ind"]) if caching else self.net(img)
        if cfg.correspondence_weight > 0:
            feats_pos, code_pos = self.net(img_pos, cache_index=batch["ind_pos"]) if caching else self.net(img_pos)
        log_args = dict(sync_dist=False, rank_zero_only=True)
        if cfg.use_true_labels:
            signal = one_hot_feats(label + 1, self.n_classes + 1)
            signal_pos = one_hot_feats(batch["label_pos"] + 1, self.n_classes + 1)
        else:
            signal, signal_pos = feats, feats_pos
        loss = 0
        if cfg.use_salience:
            salience = batch["mask"].to(torch.float32).squeeze(1)
            salience_pos = batch["mask_pos"].to(torch.float32).squeeze(1)
        else:
            salience = salience_pos = None

        if cfg.correspondence_weight > 0:
            # train_segmentation.py:163-181.  The three .mean()s of the reference come out of the forward launch itself
            # (ContrastiveCorrelationLoss.total): same values, one dot product instead of a 9 MB reduction + five scalar ops.
            (corr_total, means, pos_intra_cd, pos_inter_cd, neg_inter_cd) = self.contrastive_corr_loss_fn.total(
                signal, signal_pos, salience, salience_pos, code, code_pos,
                (cfg.pos_intra_weight, cfg.pos_inter_weight, cfg.neg_inter_weight))
            pos_intra_loss, pos_inter_loss, neg_inter_loss = means[0], means[1], means[2]
            self.log('loss/pos_intra', pos_intra_loss, **log_args)
            self.log('loss/pos_inter', pos_inter_loss, **log_args)
            self.log('loss/neg_inter', neg_inter_loss, **log_args)
            self.log('cd/pos_intra', pos_intra_cd.mean(), **log_args)
            self.log('cd/pos_inter', pos_inter_cd.mean(), **log_args)
            self.log('cd/neg_inter', neg_inter_cd.mean(), **log_args)
            loss += corr_total * cfg.correspondence_weight

        if cfg.rec_weight > 0:
            rec_loss = -(norm(self.decoder(code)) * norm(feats)).sum(1).mean()
            self.log('loss/rec', rec_loss, **log_args)
            loss += cfg.rec_weight * rec_loss

        if cfg.aug_alignment_weight > 0:
            _, code_aug = self.net(batch["img_aug"])
            coord = resize(batch["coord_aug"].permute(0, 3, 1, 2), code_aug.shape[2]).permute(0, 2, 3, 1)
            aug_alignment = -torch.einsum("bkhw,bkhw->bhw", norm(sample(code, coord)), norm(code_aug)).mean()
            self.log('loss/aug_alignment', aug_alignment, **log_args)
            loss += cfg.aug_alignment_weight * aug_alignment

        if cfg.crf_weight > 0:
            crf = self.crf_loss_fn(resize(img, 56), norm(resize(code, 56))).mean()
            self.log('loss/crf', crf, **log_args)
            loss += cfg.crf_weight * crf

        detached_code = torch.clone(code.detach())
        linear_logits = self.linear_probe(detached_code)
        linear_logits = F.interpolate(linear_logits, label.shape[-2:], mode='bilinear', align_corners=False)
        # train_segmentation.py:199-203 flattens to [pixels, classes], boolean-indexes the valid pixels (a host sync) and takes the mean
        # cross-entropy.  The same number from the spatial form: invalid labels become ignore_index, the mean runs over the rest - no
        # sync, no 170 MB permute / gather, and the 2-D NLL kernels instead of the one-block reduction ATen runs on [1.6 M, 27] (7 of the
        # 9 ms of a cached-backbone step, rocprofv3)
        valid = (label >= 0) & (label < self.n_classes)
        linear_loss = F.cross_entropy(linear_logits, torch.where(valid, label, torch.full_like(label, -100)), ignore_index=-100)
        loss += linear_loss
        self.log('loss/linear', linear_loss, **log_args)
        cluster_loss, _ = self.cluster_probe(detached_code, None)
        loss += cluster_loss
        self.log('loss/cluster', cluster_loss, **log_args)
        self.log('loss/total', loss, **log_args)

        self.manual_backward(loss)
        self.wait_gradients()
        net_optim.step()
        cluster_probe_optim.step()
        linear_probe_optim.step()

        if cfg.reset_probe_steps is not None and self.global_step == cfg.reset_probe_steps:
            print("RESETTING PROBES")
            self.linear_probe.reset_parameters()
            self.cluster_probe.reset_parameters()
            self._optims[1] = torch.optim.Adam(list(self.linear_probe.parameters()), lr=5e-3)
            self._optims[2] = torch.optim.Adam(list(self.cluster_probe.parameters()), lr=5e-3)
        self.global_step += 1
        return loss

    def validation_step(self, batch, batch_idx):
        img, label = batch["img"], batch["label"]
        self.net.eval()
        with torch.no_grad():
            _, code = self.net(img)
            code = F.interpolate(code, label.shape[-2:], mode='bilinear', align_corners=False)
            linear_preds = self.linear_probe(code).argmax(1)
            self.linear_metrics.update(linear_preds, label)
            _, cluster_preds = self.cluster_probe(code, None)
            cluster_preds = cluster_preds.argmax(1)
            self.cluster_metrics.update(cluster_preds, label)
            n = self.cfg.n_images
            return {'img': img[:n].detach().cpu(), 'linear_preds': linear_preds[:n].detach().cpu(),
                    "cluster_preds": cluster_preds[:n].detach().cpu(), "label": label[:n].detach().cpu()}

    def validation_epoch_end(self, outputs):
        with torch.no_grad():
            metrics = {**self.linear_metrics.compute(), **self.cluster_metrics.compute()}
            self.linear_metrics.reset()
            self.cluster_metrics.reset()
        self.val_steps += 1
        for k, v in metrics.items():
            self.log(k, v)
        return metrics


class SyntheticContrastiveDataset(torch.utils.data.Dataset):
    """Offline stand-in for ContrastiveSegDataset (data.py:419-565): same batch keys, seeded random content.
    The 'KNN positive' of image i is a noisy copy of it so that the positive pair is actually related."""

    def __init__(self, n, res, n_classes, seed=0):
        self.n, self.res, self.n_classes, self.seed = n, res, n_classes, seed

    def __len__(self):
        return self.n

    n_cache_items = property(lambda self: 2 * self.n)          # anchors [0, n) + their positives [n, 2n)
    deterministic_items = True                                 # the same index always yields the same pixels
    per_rank = True                                            # my_app seeds it by rank: already a different slice on every rank

    def __getitem__(self, ind):
        g = torch.Generator().manual_seed(self.seed * 100003 + ind)
        img = torch.randn(3, self.res, self.res, generator=g)
        img_pos = img + 0.3 * torch.randn(3, self.res, self.res, generator=g)
        label = torch.randint(0, self.n_classes, (self.res, self.res), generator=g)
        # the positive is a different image: it gets its own index (n + ind), as a real KNN positive has (data.py:547-549),
        # so that anything keyed by index (TokenCache) never confuses it with the anchor
        return dict(ind=ind, img=img, label=label, img_pos=img_pos, ind_pos=self.n + ind, label_pos=label,
                    mask=torch.ones(1, self.res, self.res), mask_pos=torch.ones(1, self.res, self.res))


class Trainer:
    """Minimal stand-in for the Lightning Trainer the reference builds at train_segmentation.py:476-497."""

    def __init__(self, max_steps, device=None, log_every=10, val_loader=None, val_check_interval=None, checkpoint_path=None):
        """val_loader / val_check_interval: every that many steps the model's validation_step runs over val_loader and
        validation_epoch_end reports the metrics (train_segmentation.py:247-330, Trainer(val_check_interval=cfg.val_freq) :489);
        checkpoint_path: rank 0 writes a Lightning-layout checkpoint there after every validation and at the end (the
        ModelCheckpoint callback of :482-486)."""
        self.max_steps, self.log_every = max_steps, log_every
        self.val_loader, self.val_check_interval, self.checkpoint_path = val_loader, val_check_interval, checkpoint_path
        self.val_history = []
        self.rank, self.world, self.local_rank = ddp.init_from_env()
        self.device = device or (torch.device("cuda", self.local_rank) if torch.cuda.is_available() else torch.device("cpu"))

    def fit(self, model, loader):
        model.to(self.device)
        model.train()
        if self.world > 1:
            model.setup_distributed()
        if getattr(model.cfg, "cache_backbone_tokens", False) and hasattr(model.net, "enable_token_cache"):
            # The cache is keyed by dataset index: valid only if an index always yields the same pixels (fixed crops; the
            # reference's random-crop loaders, data.py loader_crop_type "random", re-crop every epoch) and if ind_pos
            # addresses the positive's own pixels.
            ds = loader.dataset
            crop = getattr(model.cfg, "loader_crop_type", "center")
            if not getattr(ds, "deterministic_items", crop != "random"):
                raise ValueError("cfg.cache_backbone_tokens needs a dataset whose items are fixed per index "
                                 "(loader_crop_type=%r re-crops every epoch)" % crop)
            n_items = int(getattr(ds, "n_cache_items", len(ds)))
            model.net.enable_token_cache(n_items, (model.cfg.res, model.cfg.res), self.device)
        loader = self._shard_loader(loader)
        if len(loader) == 0:
            raise ValueError("empty loader (dataset of %d items, batch size %s, drop_last): nothing to train on"
                             % (len(loader.dataset), getattr(loader, "batch_size", "?")))
        step = 0
        epoch = 0
        history = []
        while step < self.max_steps:
            if isinstance(getattr(loader, "sampler", None), torch.utils.data.distributed.DistributedSampler):
                loader.sampler.set_epoch(epoch)      # a new shuffle per epoch, the same on every rank (Lightning does this)
            epoch += 1
            for batch in loader:
                batch = {k: (v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}
                loss = model.training_step(batch, step)
                # the loss stays on the device (a float() here is a host sync every step: the CPU could never run ahead of the GPU);
                # values are read when they are printed and once at the end
                history.append(loss.detach())
                if self.rank == 0 and step % self.log_every == 0:
                    print("step %5d  loss %.5f  " % (step, float(history[-1])) +
                          "  ".join("%s %.5f" % (k, float(v)) for k, v in model.logged.items() if k.startswith("loss/pos") or k.startswith("loss/neg")))
                step += 1
                if step % self.event_check_every == 0 or step >= self.max_steps:
                    self._check_loss_events(step)
                if self.val_loader is not None and self.val_check_interval and step % self.val_check_interval == 0:
                    self._validate(model)
                if step >= self.max_steps:
                    break
        if self.checkpoint_path and self.rank == 0:
            model.save_checkpoint(self.checkpoint_path)
        return [float(v) for v in torch.stack(history).cpu()] if history else []

    event_check_every = 200        # steps between two looks at the loss kernels' event counters (a device -> host copy of a few bytes)
    _events_warned = False

    def _check_loss_events(self, step):
        """The single-launch forward of the correspondence loss has two bounded waits between workgroups; a launch whose workgroups are not
        all on the device at once (another kernel holds compute units: a shared or partitioned device) gives them up and takes its fallback
        paths - same bytes, ~30 us slower per step, and silent.  The library counts those events per workspace; said once here."""
        if self._events_warned or self.device.type != "cuda":
            return
        try:
            from . import capi
            gave_up, repaired = capi.event_counters_total()
        except Exception:       # noqa: BLE001 - a diagnostic must not stop a training run
            return
        if gave_up or repaired:
            import warnings
            self._events_warned = True
            warnings.warn("correspondence loss (rank %d, step %d): %d tiles sampled their anchor themselves and %d old_mean rendezvous timed out "
                          "since the run started - the forward's workgroups are not co-resident (device shared with other kernels, or "
                          "partitioned?).  Results are unaffected; set cfg.shared_device = True (STEGO_FLAG_SHARED_DEVICE) when kernels "
                          "of other streams run beside the loss." % (self.rank, step, gave_up, repaired))

    def _shard_loader(self, loader):
        """Data parallelism needs every rank to see its own slice: a loader without a DistributedSampler over a dataset that is
        not already different per rank (``dataset.per_rank``, the synthetic set is seeded by rank) is rebuilt with one - what
        Lightning's ``replace_sampler_ddp`` does for the reference (train_segmentation.py:461-468)."""
        ds = loader.dataset
        if self.world <= 1 or getattr(ds, "per_rank", False) or \
                isinstance(getattr(loader, "sampler", None), torch.utils.data.distributed.DistributedSampler):
            return loader
        if isinstance(ds, torch.utils.data.IterableDataset):
            raise ValueError("data-parallel training over an IterableDataset: shard it per rank yourself (set dataset.per_rank = True)")
        if loader.batch_size is None:                  # built with batch_sampler=...: batching is not ours to redo
            raise ValueError("data-parallel training needs a loader built with batch_size (not batch_sampler), a DistributedSampler, "
                             "or a dataset with per_rank = True")
        shuffle = isinstance(getattr(loader, "sampler", None), torch.utils.data.RandomSampler)
        # DistributedSampler's default drop_last=False pads the last rank(s) by repeating samples - what Lightning's
        # replace_sampler_ddp gives the reference; whether incomplete BATCHES are dropped stays the loader's own drop_last
        sampler = torch.utils.data.distributed.DistributedSampler(ds, num_replicas=self.world, rank=self.rank, shuffle=shuffle, seed=0)
        kw = dict(batch_size=loader.batch_size, sampler=sampler, num_workers=loader.num_workers, collate_fn=loader.collate_fn,
                  pin_memory=loader.pin_memory, drop_last=loader.drop_last, timeout=loader.timeout,
                  worker_init_fn=loader.worker_init_fn, generator=loader.generator)
        if loader.num_workers > 0:
            kw.update(prefetch_factor=loader.prefetch_factor, persistent_workers=loader.persistent_workers)
        return torch.utils.data.DataLoader(ds, **kw)

    def _validate(self, model):
        model.eval()
        outs = []
        with torch.no_grad():
            for i, batch in enumerate(self.val_loader):
                batch = {k: (v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}
                outs.append(model.validation_step(batch, i))
        metrics = model.validation_epoch_end(outs)
        self.val_history.append(metrics)
        model.train()
        if self.checkpoint_path and self.rank == 0:
            model.save_checkpoint(self.checkpoint_path)
        return metrics


def my_app(cfg):
    torch.manual_seed(0)            # seed_everything(0), train_segmentation.py:403
    n_classes = get_class_labels(cfg.dataset_name)
    trainer = Trainer(cfg.max_steps)
    ds = SyntheticContrastiveDataset(max(cfg.batch_size * 8, 64), cfg.res, n_classes, seed=trainer.rank)
    loader = torch.utils.data.DataLoader(ds, cfg.batch_size, shuffle=True, num_workers=0, drop_last=True)
    model = LitUnsupervisedSegmenter(n_classes, cfg)
    return trainer.fit(model, loader)


if __name__ == "__main__":
    prep_args()
    my_app(load_config(overrides=sys.argv[1:]))
