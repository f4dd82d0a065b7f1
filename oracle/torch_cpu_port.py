"""Torch-on-CPU port of the reference correspondence loss  --  TEST / BASELINE INFRASTRUCTURE.

Purpose: the ``cpu_baseline`` leg of bench.py.  The reference's CPU path *is* a chain of
ATen CPU kernels (grid_sampler_2d, normalize, bmm, clamp, mean), and the
reference itself cannot travel to the GPU box, so this file re-expresses the same chain of
ATen calls, pair-set by pair-set, with the RNG draws passed in.  It follows
``/root/reference/src/modules.py``: sample :287-288, norm :275-276, tensor_correlation
:283-284, helper :325-347, forward :369-398 (including the full-tensor
``orig_feats[perm]`` gather of :384-385 that dominates the reference's CPU time).
It is checked against the reference's golden vectors in tests/test_oracle_golden.py.
Nothing under stego_amd/ imports it.
"""
import torch
import torch.nn.functional as F


def _grid(coords):
    return coords.permute(0, 2, 1, 3)


def _bilinear(t, coords):
    return F.grid_sample(t, _grid(coords), padding_mode="border", align_corners=True)


def _unit(t):
    return F.normalize(t, dim=1, eps=1e-10)


def _pairwise(a, b):
    return torch.einsum("nchw,ncij->nhwij", a, b)


def _pair_term(fa, fb, ca, cb, shift, cfg):
    with torch.no_grad():
        fd = _pairwise(_unit(fa), _unit(fb))
        if cfg.pointwise:
            before = fd.mean()
            fd -= fd.mean([3, 4], keepdim=True)
            fd = fd - fd.mean() + before
    cd = _pairwise(_unit(ca), _unit(cb))
    lo = 0.0 if cfg.zero_clamp else -9999.0
    clamped = cd.clamp(lo, .8) if cfg.stabalize else cd.clamp(lo)
    return -clamped * (fd - shift), cd


def corr_loss_torch_cpu(feats, feats_pos, code, code_pos, coords1, coords2, perms, cfg):
    """Returns the reference's 6-tuple. All tensors torch CPU; perms: iterable of int64 [B]."""
    fa, ca = _bilinear(feats, coords1), _bilinear(code, coords1)
    fp, cp = _bilinear(feats_pos, coords2), _bilinear(code_pos, coords2)
    intra, intra_cd = _pair_term(fa, fa, ca, ca, cfg.pos_intra_shift, cfg)
    inter, inter_cd = _pair_term(fa, fp, ca, cp, cfg.pos_inter_shift, cfg)
    nl, ncd = [], []
    for perm in perms:
        fn = _bilinear(feats[perm], coords2)
        cn = _bilinear(code[perm], coords2)
        l, c = _pair_term(fa, fn, ca, cn, cfg.neg_inter_shift, cfg)
        nl.append(l)
        ncd.append(c)
    S = coords1.shape[1]
    if nl:
        nl, ncd = torch.cat(nl, 0), torch.cat(ncd, 0)
    else:
        nl, ncd = feats.new_zeros(0, S, S, S, S), feats.new_zeros(0, S, S, S, S)
    return intra.mean(), intra_cd, inter.mean(), inter_cd, nl, ncd
