"""CPU oracle for STEGO's feature-correspondence loss  --  TEST INFRASTRUCTURE ONLY.

This file is a plain-numpy restatement of the algorithm in the reference's
``src/modules.py:275-398`` (``norm``, ``tensor_correlation``, ``sample``,
``super_perm``, ``ContrastiveCorrelationLoss.helper/forward``) and of the autograd
backward PyTorch derives for it.  It is the checker the
``tests/`` suite, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg compare the HIP path against; nothing under ``stego_amd/`` may import it.

Parity status: PINNED.  ``oracle/make_golden.py`` imports the unmodified reference
from ``/root/reference/src`` (through a stub ``utils`` module) in the build
container, runs it on seeded inputs and commits inputs+outputs under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks this restatement against
those fixtures and against the known-answer anchors of SURVEY.md section 8(c).

The arithmetic below ATen (``F.grid_sample``, ``F.normalize``, ``einsum``) is not
vendored in the reference (it pins ``pytorch==1.7.1``, environment.yml:9); the
semantics restated here are those of ATen's ``grid_sampler_2d`` (bilinear,
``padding_mode='border'``, ``align_corners=True``) and ``normalize`` (eps-clamped
L2), unchanged between 1.7 and 2.10.

All functions take/return numpy arrays; ``dtype`` selects the arithmetic type
(float64 = "truth" for tolerance tests, float32 = same precision as the reference).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np


# --------------------------------------------------------------------------- cfg
@dataclass
class CorrCfg:
    """The cfg keys ContrastiveCorrelationLoss reads (reference modules.py:330-387,
    configs/train_config.yml:41-64).  Field names follow the reference, including
    its spelling of ``stabalize``."""
    pointwise: bool = True
    zero_clamp: bool = True
    stabalize: bool = False
    use_salience: bool = False
    feature_samples: int = 11
    neg_samples: int = 5
    pos_intra_shift: float = 0.18
    pos_inter_shift: float = 0.12
    neg_inter_shift: float = 0.46


# ------------------------------------------------------------------ primitives
def norm(t: np.ndarray) -> np.ndarray:
    """reference modules.py:275-276: F.normalize(t, dim=1, eps=1e-10) = t / max(||t||_2, eps)."""
    n = np.sqrt((t * t).sum(axis=1, keepdims=True))
    return t / np.maximum(n, 1e-10)


def tensor_correlation(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """reference modules.py:283-284: einsum('nchw,ncij->nhwij')."""
    n, c, h, w = a.shape
    _, _, i, j = b.shape
    at = np.ascontiguousarray(a.reshape(n, c, h * w).transpose(0, 2, 1))     # (BLAS path needs dense operands)
    out = np.matmul(at, np.ascontiguousarray(b.reshape(n, c, i * j)))
    return out.reshape(n, h, w, i, j)


def grid_sample_bilinear_border(t: np.ndarray, grid: np.ndarray) -> np.ndarray:
    """ATen grid_sampler_2d, bilinear / padding_mode='border' / align_corners=True.

    t: [N,C,H,W]; grid: [N,Ho,Wo,2] with grid[...,0]=x (width), grid[...,1]=y (height)
    in [-1,1].  Unnormalise: ix=(x+1)/2*(W-1); border clip to [0,W-1]; the four
    corner taps are weighted by the opposite-corner areas, out-of-range taps
    contribute zero (their weight is zero whenever they are out of range).
    """
    N, C, H, W = t.shape
    dt = t.dtype
    x = grid[..., 0].astype(dt)
    y = grid[..., 1].astype(dt)
    ix = (x + 1) / 2 * (W - 1)
    iy = (y + 1) / 2 * (H - 1)
    ix = np.minimum(np.maximum(ix, 0), W - 1)
    iy = np.minimum(np.maximum(iy, 0), H - 1)
    x0 = np.floor(ix)
    y0 = np.floor(iy)
    x1 = x0 + 1
    y1 = y0 + 1
    w_nw = (x1 - ix) * (y1 - iy)
    w_ne = (ix - x0) * (y1 - iy)
    w_sw = (x1 - ix) * (iy - y0)
    w_se = (ix - x0) * (iy - y0)
    out = np.zeros((N, C) + x.shape[1:], dtype=dt)
    nidx = np.arange(N)[:, None, None]
    for xx, yy, ww in ((x0, y0, w_nw), (x1, y0, w_ne), (x0, y1, w_sw), (x1, y1, w_se)):
        inb = (xx >= 0) & (xx <= W - 1) & (yy >= 0) & (yy <= H - 1)
        xi = np.clip(xx, 0, W - 1).astype(np.int64)
        yi = np.clip(yy, 0, H - 1).astype(np.int64)
        vals = t[nidx, :, yi, xi]                 # [N,Ho,Wo,C]
        vals = np.moveaxis(vals, -1, 1)           # [N,C,Ho,Wo]
        out += vals * (ww * inb)[:, None]
    return out


def sample(t: np.ndarray, coords: np.ndarray) -> np.ndarray:
    """reference modules.py:287-288: grid_sample(t, coords.permute(0,2,1,3), border, align_corners)."""
    return grid_sample_bilinear_border(t, coords.transpose(0, 2, 1, 3))


def super_perm_from_randperm(perm: np.ndarray) -> np.ndarray:
    """Deterministic tail of reference modules.py:291-295 given the randperm draw:
    fixed points get +1, then everything is taken mod size (duplicates possible)."""
    perm = perm.astype(np.int64).copy()
    size = perm.shape[0]
    perm[perm == np.arange(size)] += 1
    return perm % size


# ----------------------------------------------------------------------- helper
def _clamp_bounds(cfg) -> Tuple[float, float]:
    lo = 0.0 if cfg.zero_clamp else -9999.0          # modules.py:337-340
    hi = 0.8 if cfg.stabalize else np.inf            # modules.py:342-345
    return lo, hi


def helper(f1, f2, c1, c2, shift, cfg) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """reference modules.py:325-347.  Returns (loss, cd, fd_final) - fd_final is the
    feature correlation after the pointwise shift, exposed for debugging/backward."""
    fd = tensor_correlation(norm(f1), norm(f2))
    if cfg.pointwise:
        old_mean = fd.mean()
        fd = fd - fd.mean(axis=(3, 4), keepdims=True)
        fd = fd - fd.mean() + old_mean
    cd = tensor_correlation(norm(c1), norm(c2))
    lo, hi = _clamp_bounds(cfg)
    loss = -np.clip(cd, lo, hi) * (fd - shift)
    return loss, cd, fd


# ---------------------------------------------------------------------- forward
@dataclass
class CorrOut:
    pos_intra_loss: np.ndarray      # scalar (mean)           modules.py:393
    pos_intra_cd: np.ndarray        # [B,S,S,S,S]
    pos_inter_loss: np.ndarray      # scalar (mean)           modules.py:395
    pos_inter_cd: np.ndarray        # [B,S,S,S,S]
    neg_inter_loss: np.ndarray      # [N_neg*B,S,S,S,S]       modules.py:390
    neg_inter_cd: np.ndarray        # [N_neg*B,S,S,S,S]       modules.py:391
    fd: List[np.ndarray]            # debug: final fd per pair-set (intra, inter, neg0..)

    def as_tuple(self):
        return (self.pos_intra_loss, self.pos_intra_cd, self.pos_inter_loss,
                self.pos_inter_cd, self.neg_inter_loss, self.neg_inter_cd)


def corr_loss_forward(feats, feats_pos, code, code_pos,
                      coords1, coords2, perms: Sequence[np.ndarray], cfg,
                      dtype=np.float64) -> CorrOut:
    """reference modules.py:349-398 with the RNG draws (coords1, coords2, perm x N_neg)
    passed in explicitly (they are torch RNG draws at :366,:367,:383)."""
    f = np.asarray(feats, dtype=dtype)        # orig_feats
    fp = np.asarray(feats_pos, dtype=dtype)   # orig_feats_pos
    c = np.asarray(code, dtype=dtype)         # orig_code
    cp = np.asarray(code_pos, dtype=dtype)    # orig_code_pos
    c1 = np.asarray(coords1, dtype=dtype)
    c2 = np.asarray(coords2, dtype=dtype)

    s_f = sample(f, c1)                       # :369
    s_c = sample(c, c1)                       # :370
    s_fp = sample(fp, c2)                     # :372
    s_cp = sample(cp, c2)                     # :373

    intra_loss, intra_cd, fd0 = helper(s_f, s_f, s_c, s_c, cfg.pos_intra_shift, cfg)        # :375
    inter_loss, inter_cd, fd1 = helper(s_f, s_fp, s_c, s_cp, cfg.pos_inter_shift, cfg)      # :377
    fds = [fd0, fd1]
    neg_losses, neg_cds = [], []
    for perm in perms:                        # :382-388
        perm = np.asarray(perm, dtype=np.int64)
        feats_neg = sample(f[perm], c2)
        code_neg = sample(c[perm], c2)
        nl, ncd, nfd = helper(s_f, feats_neg, s_c, code_neg, cfg.neg_inter_shift, cfg)
        neg_losses.append(nl)
        neg_cds.append(ncd)
        fds.append(nfd)
    S = c1.shape[1]
    if neg_losses:
        neg_loss = np.concatenate(neg_losses, axis=0)
        neg_cd = np.concatenate(neg_cds, axis=0)
    else:
        neg_loss = np.zeros((0, S, S, S, S), dtype=dtype)
        neg_cd = np.zeros((0, S, S, S, S), dtype=dtype)
    return CorrOut(intra_loss.mean(), intra_cd, inter_loss.mean(), inter_cd, neg_loss, neg_cd, fds)


# --------------------------------------------------------------------- backward
def _normalize_bwd(t: np.ndarray, g_n: np.ndarray) -> np.ndarray:
    """Backward of norm() along dim 1: d t = (g - n <n,g>) / ||t||  (for ||t|| > eps;
    with ||t|| <= eps the forward is t/eps and the backward g/eps)."""
    nrm = np.sqrt((t * t).sum(axis=1, keepdims=True))
    big = nrm > 1e-10
    safe = np.where(big, nrm, 1.0)
    n = t / safe
    dot = (n * g_n).sum(axis=1, keepdims=True)
    return np.where(big, (g_n - n * dot) / safe, g_n / 1e-10)


def _grid_sample_bwd_input(shape, grid, g_out) -> np.ndarray:
    """Scatter-add adjoint of grid_sample_bilinear_border w.r.t. the input map."""
    N, C, H, W = shape
    dt = g_out.dtype
    x = grid[..., 0].astype(dt)
    y = grid[..., 1].astype(dt)
    ix = np.minimum(np.maximum((x + 1) / 2 * (W - 1), 0), W - 1)
    iy = np.minimum(np.maximum((y + 1) / 2 * (H - 1), 0), H - 1)
    x0 = np.floor(ix); y0 = np.floor(iy); x1 = x0 + 1; y1 = y0 + 1
    taps = ((x0, y0, (x1 - ix) * (y1 - iy)), (x1, y0, (ix - x0) * (y1 - iy)),
            (x0, y1, (x1 - ix) * (iy - y0)), (x1, y1, (ix - x0) * (iy - y0)))
    g_in = np.zeros((N, H, W, C), dtype=dt)
    g_perm = np.moveaxis(g_out, 1, -1)        # [N,Ho,Wo,C]
    nidx = np.broadcast_to(np.arange(N)[:, None, None], x.shape)
    for xx, yy, ww in taps:
        inb = (xx >= 0) & (xx <= W - 1) & (yy >= 0) & (yy <= H - 1)
        xi = np.clip(xx, 0, W - 1).astype(np.int64)
        yi = np.clip(yy, 0, H - 1).astype(np.int64)
        np.add.at(g_in, (nidx, yi, xi), g_perm * (ww * inb)[..., None])
    return np.moveaxis(g_in, -1, 1)


def sample_bwd(shape, coords, g_out):
    return _grid_sample_bwd_input(shape, coords.transpose(0, 2, 1, 3), g_out)


def _helper_bwd_codes(c1s, c2s, fd_final, cd, shift, cfg, g_loss, g_cd):
    """d(loss,cd)/d(sampled c1, c2) for one helper call; c1s,c2s are the *sampled raw*
    codes [N,K,S,S]; g_loss/g_cd are upstream gradients on helper's two outputs."""
    lo, hi = _clamp_bounds(cfg)
    passes = (cd >= lo) & (cd <= hi)          # clamp backward mask (inclusive bounds)
    G = -(fd_final - shift) * passes * g_loss
    if g_cd is not None:
        G = G + g_cd
    n = cd.shape[0]
    P1 = cd.shape[1] * cd.shape[2]            # A-side points (h,w)
    P2 = cd.shape[3] * cd.shape[4]            # B-side points (i,j)
    Gm = G.reshape(n, P1, P2)
    n1 = norm(c1s).reshape(n, -1, P1)         # [n,K,P1]
    n2 = norm(c2s).reshape(n, -1, P2)
    g_n1 = np.matmul(n2, Gm.transpose(0, 2, 1)).reshape(c1s.shape)   # dA[k,hw] = sum_ij G[hw,ij] B[k,ij]
    g_n2 = np.matmul(n1, Gm).reshape(c2s.shape)                      # dB[k,ij] = sum_hw G[hw,ij] A[k,hw]
    return _normalize_bwd(c1s, g_n1), _normalize_bwd(c2s, g_n2)


def corr_loss_backward(feats, feats_pos, code, code_pos,
                       coords1, coords2, perms, cfg,
                       g_intra: float, g_inter: float, g_neg_loss: Optional[np.ndarray],
                       g_intra_cd=None, g_inter_cd=None, g_neg_cd=None,
                       dtype=np.float64):
    """Gradients of  g_intra*out[0] + g_inter*out[2] + <g_neg_loss,out[4]> (+ cd terms)
    w.r.t. orig_code and orig_code_pos  (what autograd computes through
    modules.py:349-398; the feature side is no_grad, :326)."""
    f = np.asarray(feats, dtype=dtype); fp = np.asarray(feats_pos, dtype=dtype)
    c = np.asarray(code, dtype=dtype); cp = np.asarray(code_pos, dtype=dtype)
    c1 = np.asarray(coords1, dtype=dtype); c2 = np.asarray(coords2, dtype=dtype)
    B = c.shape[0]
    S = c1.shape[1]
    numel = B * S ** 4

    feats = sample(f, c1); code = sample(c, c1)              # from here on: the sampled tensors
    feats_pos = sample(fp, c2); code_pos = sample(cp, c2)
    d_code = np.zeros_like(c)
    d_code_pos = np.zeros_like(cp)

    # intra
    _, cd, fd = helper(feats, feats, code, code, cfg.pos_intra_shift, cfg)
    ga, gb = _helper_bwd_codes(code, code, fd, cd, cfg.pos_intra_shift, cfg,
                               np.full(cd.shape, g_intra / numel, dtype=dtype), g_intra_cd)
    d_code += sample_bwd(c.shape, c1, ga + gb)
    # inter
    _, cd, fd = helper(feats, feats_pos, code, code_pos, cfg.pos_inter_shift, cfg)
    ga, gb = _helper_bwd_codes(code, code_pos, fd, cd, cfg.pos_inter_shift, cfg,
                               np.full(cd.shape, g_inter / numel, dtype=dtype), g_inter_cd)
    d_code += sample_bwd(c.shape, c1, ga)
    d_code_pos += sample_bwd(cp.shape, c2, gb)
    # negatives
    for i, perm in enumerate(perms):
        perm = np.asarray(perm, dtype=np.int64)
        feats_neg = sample(f[perm], c2)
        code_neg = sample(c[perm], c2)
        _, cd, fd = helper(feats, feats_neg, code, code_neg, cfg.neg_inter_shift, cfg)
        gl = (np.asarray(g_neg_loss[i * B:(i + 1) * B], dtype=dtype) if g_neg_loss is not None
              else np.zeros(cd.shape, dtype=dtype))
        gc = None if g_neg_cd is None else np.asarray(g_neg_cd[i * B:(i + 1) * B], dtype=dtype)
        ga, gb = _helper_bwd_codes(code, code_neg, fd, cd, cfg.neg_inter_shift, cfg, gl, gc)
        d_code += sample_bwd(c.shape, c1, ga)
        g_gathered = sample_bwd(c.shape, c2, gb)          # grad of orig_code[perm]
        np.add.at(d_code, perm, g_gathered)               # index backward = index_put(accumulate)
    return d_code, d_code_pos


# --------------------------------------------------------------- synthetic data
def synth_inputs(B, C, H, W, K, S, n_neg, seed, dino_like=False, dtype=np.float32):
    """Seeded synthetic inputs.  Returns a dict with feats,
    feats_pos, code, code_pos, coords1, coords2, perms (already super_perm'ed)."""
    rng = np.random.default_rng(seed)
    if dino_like:
        R = 8
        proto = rng.standard_normal((R, C))
        def field():
            z = rng.standard_normal((B, R, H // 4 + 2, W // 4 + 2))
            z = np.repeat(np.repeat(z, 4, axis=2), 4, axis=3)[:, :, :H, :W]
            return z
        def mk():
            x = np.einsum("brhw,rc->bchw", field(), proto) + 0.3 * rng.standard_normal((B, C, H, W))
            keep = (rng.random((B, C, 1, 1)) > 0.1) / 0.9
            return x * keep
        feats, feats_pos = mk(), mk()
        head = rng.standard_normal((K, C)) / np.sqrt(C)
        code = np.einsum("kc,bchw->bkhw", head, feats)
        code_pos = np.einsum("kc,bchw->bkhw", head, feats_pos)
    else:
        feats = rng.standard_normal((B, C, H, W))
        feats_pos = rng.standard_normal((B, C, H, W))
        code = rng.standard_normal((B, K, H, W))
        code_pos = rng.standard_normal((B, K, H, W))
    coords1 = rng.random((B, S, S, 2)) * 2 - 1
    coords2 = rng.random((B, S, S, 2)) * 2 - 1
    perms = [super_perm_from_randperm(rng.permutation(B)) for _ in range(n_neg)]
    return dict(feats=feats.astype(dtype), feats_pos=feats_pos.astype(dtype),
                code=code.astype(dtype), code_pos=code_pos.astype(dtype),
                coords1=coords1.astype(dtype), coords2=coords2.astype(dtype),
                perms=np.stack(perms).astype(np.int64) if n_neg else np.zeros((0, B), np.int64))
