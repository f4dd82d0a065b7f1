"""Test double for ``stego_amd.capi`` backed by the numpy oracle, so the HOST logic of
``stego_amd.modules`` (RNG draw order, argument packing, autograd wiring, output shapes)
can be exercised on CPU tensors.  Lives under tests/ on purpose: the product has no
CPU path."""
import numpy as np
import torch

from oracle import corr_oracle as O


def _cfg_from_desc(desc, n_neg=None):
    return O.CorrCfg(pointwise=bool(desc.pointwise), zero_clamp=bool(desc.zero_clamp), stabalize=bool(desc.stabalize),
                     feature_samples=desc.S, neg_samples=desc.n_neg if n_neg is None else n_neg,
                     pos_intra_shift=desc.pos_intra_shift, pos_inter_shift=desc.pos_inter_shift,
                     neg_inter_shift=desc.neg_inter_shift)


def _np(t):
    return None if t is None else t.detach().cpu().numpy()


def _t(a, like):
    return torch.from_numpy(np.ascontiguousarray(a)).to(torch.float32).to(like.device)


calls = []
last_neg_is_mean = None


def corr_fwd(desc, feats, feats_pos, code, code_pos, coords1, coords2, perms, need_grad):
    calls.append("corr_fwd")
    cfg = _cfg_from_desc(desc)
    out = O.corr_loss_forward(_np(feats), _np(feats_pos), _np(code), _np(code_pos), _np(coords1), _np(coords2),
                              list(_np(perms)) if desc.n_neg else [], cfg)
    S = desc.S
    neg_mean = float(np.mean(out.neg_inter_loss)) if desc.n_neg else 0.0
    loss_means = _t(np.array([out.pos_intra_loss, out.pos_inter_loss, neg_mean]), feats)
    saved = None
    if need_grad:   # opaque to the host layer; keep what our corr_bwd below needs
        saved = (_t(np.zeros(1), feats), _t(np.zeros(1), feats), _t(np.zeros(1), feats))
        corr_fwd.stash = (_np(feats), _np(feats_pos))
    return (loss_means, _t(out.pos_intra_cd, feats), _t(out.pos_inter_cd, feats),
            _t(out.neg_inter_loss.reshape(-1, S, S, S, S), feats), _t(out.neg_inter_cd.reshape(-1, S, S, S, S), feats),
            saved)


def corr_bwd(desc, code, code_pos, coords1, coords2, perms, saved, intra_cd, inter_cd, neg_cd,
             g_intra, g_inter, g_neg_loss, g_intra_cd, g_inter_cd, g_neg_cd, neg_is_mean=False):
    calls.append("corr_bwd")
    global last_neg_is_mean
    last_neg_is_mean = bool(neg_is_mean)
    cfg = _cfg_from_desc(desc)
    f, fp = corr_fwd.stash
    S = desc.S
    gnl = None
    if g_neg_loss is not None and desc.n_neg:
        g = _np(g_neg_loss)
        if neg_is_mean:                                   # upstream of the mean over all negative losses
            g = g.reshape(()) / float(desc.n_neg * desc.B * S ** 4)
        gnl = np.broadcast_to(g, (desc.n_neg * desc.B, S, S, S, S))
    dc, dcp = O.corr_loss_backward(
        f, fp, _np(code), _np(code_pos), _np(coords1), _np(coords2), list(_np(perms)) if desc.n_neg else [], cfg,
        0.0 if g_intra is None else float(g_intra), 0.0 if g_inter is None else float(g_inter), gnl,
        _np(g_intra_cd), _np(g_inter_cd), _np(g_neg_cd) if desc.n_neg else None)
    return _t(dc, code), _t(dcp, code)


def helper_fwd(desc, f1, f2, c1, c2, need_grad):
    calls.append("helper_fwd")
    cfg = _cfg_from_desc(desc)
    loss, cd, fd = O.helper(_np(f1).astype(np.float64), _np(f2).astype(np.float64), _np(c1).astype(np.float64),
                            _np(c2).astype(np.float64), desc.pos_intra_shift, cfg)
    helper_fwd.stash = (fd, cd)
    return _t(loss, f1), _t(cd, f1), ((_t(np.zeros(1), f1),) * 3 if need_grad else None)


def helper_bwd(desc, c1, c2, saved, cd, g_loss, g_cd):
    calls.append("helper_bwd")
    cfg = _cfg_from_desc(desc)
    fd, cdv = helper_fwd.stash
    gl = np.zeros_like(cdv) if g_loss is None else _np(g_loss).astype(np.float64)
    ga, gb = O._helper_bwd_codes(_np(c1).astype(np.float64), _np(c2).astype(np.float64), fd, cdv,
                                 desc.pos_intra_shift, cfg, gl, None if g_cd is None else _np(g_cd).astype(np.float64))
    return _t(ga, c1), _t(gb, c1)


def knn_topk(x, k=30, normalize=False, q_begin=0, q_count=None, return_sims=False):
    """Oracle-backed double of capi.knn_topk (CPU tests of the sharding logic only)."""
    import numpy as _np
    import torch as _torch
    from oracle import knn_oracle as _K
    idx, val = _K.knn_topk(x.detach().cpu().numpy(), k, normalize=normalize, q_begin=q_begin, q_count=q_count)
    i = _torch.from_numpy(idx)
    return (i, _torch.from_numpy(val.astype(_np.float32))) if return_sims else i
