"""The few helpers of the reference's ``src/utils.py`` the training path needs (the original imports wget,
torch._six, torchmetrics and torchvision at module top, utils.py:11-19, and cannot be imported on current
PyTorch).  Plain torch/numpy/scipy."""
import sys

import numpy as np
import torch
import torch.nn.functional as F


def resize(classes: torch.Tensor, size: int):
    """utils.py:60-62."""
    return F.interpolate(classes, (size, size), mode="bilinear", align_corners=False)


def one_hot_feats(labels, n_classes):
    """utils.py:65-66."""
    return F.one_hot(labels, n_classes).permute(0, 3, 1, 2).to(torch.float32)


def prep_args(argv=None):
    """utils.py:149-162: turns `--key value` into hydra-style `key=value` (in place on sys.argv by default)."""
    args = sys.argv if argv is None else argv
    out = [args[0]]
    rest = list(args[1:])
    while rest:
        a = rest.pop(0)
        if len(a.split("=")) == 2:
            out.append(a)
        elif a.startswith("--"):
            out.append(a[2:] + "=" + rest.pop(0))
        else:
            raise ValueError("Unexpected arg style {}".format(a))
    if argv is None:
        sys.argv = out
    return out


class UnsupervisedMetrics:
    """Confusion-matrix metric with Hungarian cluster->class matching (utils.py:203-274), without
    torchmetrics: state is an int64 [n_classes+extra, n_classes] matrix, summed across ranks on compute()."""

    def __init__(self, prefix: str, n_classes: int, extra_clusters: int, compute_hungarian: bool):
        self.prefix, self.n_classes, self.extra_clusters = prefix, n_classes, extra_clusters
        self.compute_hungarian = compute_hungarian
        self.stats = torch.zeros(n_classes + extra_clusters, n_classes, dtype=torch.int64)

    def reset(self):
        self.stats.zero_()

    def update(self, preds: torch.Tensor, target: torch.Tensor):
        with torch.no_grad():
            actual, preds = target.reshape(-1), preds.reshape(-1)
            mask = (actual >= 0) & (actual < self.n_classes) & (preds >= 0) & (preds < self.n_classes)
            actual, preds = actual[mask], preds[mask]
            rows = self.n_classes + self.extra_clusters
            hist = torch.bincount(rows * actual + preds, minlength=self.n_classes * rows)
            self.stats += hist.reshape(self.n_classes, rows).t().cpu()

    def compute(self):
        from scipy.optimize import linear_sum_assignment
        stats = self.stats.clone()
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            buf = stats.cuda() if dist.get_backend() == "nccl" else stats
            dist.all_reduce(buf)
            stats = buf.cpu()
        if self.compute_hungarian:
            self.assignments = linear_sum_assignment(stats.numpy(), maximize=True)
            if self.extra_clusters == 0:
                hist = stats[np.argsort(self.assignments[1]), :]
            else:
                at = linear_sum_assignment(stats.numpy().T, maximize=True)
                hist = stats[at[1], :]
                missing = sorted(set(range(self.n_classes + self.extra_clusters)) - set(self.assignments[0]))
                hist = torch.cat([hist, stats[missing, :].sum(0, keepdim=True)], 0)
                hist = torch.cat([hist, torch.zeros(self.n_classes + 1, 1, dtype=hist.dtype)], 1)
        else:
            hist = stats
        hist = hist.double()
        tp = torch.diag(hist)
        fp, fn = hist.sum(0) - tp, hist.sum(1) - tp
        iou = tp / (tp + fp + fn)
        acc = tp.sum() / hist.sum()
        return {self.prefix + "mIoU": 100 * iou[~torch.isnan(iou)].mean().item(),
                self.prefix + "Accuracy": 100 * acc.item()}
