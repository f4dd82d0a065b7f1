"""Parity of the HIP path (through the C ABI) against the reference's golden vectors and
the fp64 oracle.  Tolerance (north_star): 1e-3 relative in fp32, written out in
conftest.assert_close (rtol=1e-3, atol = 1e-4 * mean|expected|).  Needs the MI355X."""
import os

import numpy as np
import pytest
import torch

from conftest import ALL_CASES, GoldenCase, assert_close
from oracle import corr_oracle as O
from stego_amd import capi
from stego_amd import modules as M

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def _channels_last(t):
    return t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)


def _run(case_inputs, perms, cfg, layout="nchw", grad=True, upstream=None, precision="f32"):
    """Run the HIP forward (+ backward) on a golden/synthetic case. Returns numpy outputs."""
    import copy
    cfg = copy.copy(cfg)
    cfg.corr_precision = precision
    t = {k: _dev(v) for k, v in case_inputs.items()}
    f, fp, c, cp = t["feats"], t["feats_pos"], t["code"], t["code_pos"]
    if layout == "cl":
        f, fp, c, cp = (_channels_last(x) for x in (f, fp, c, cp))
    c = c.detach().requires_grad_(grad)
    cp = cp.detach().requires_grad_(grad)
    perms_t = _dev(perms) if perms is not None and len(perms) else None
    out = M.ContrastiveCorrelationLoss(cfg).forward_explicit(f, fp, c, cp, t["coords1"], t["coords2"], perms_t)
    res = dict(out=[o.detach().cpu().numpy() for o in out])
    if grad:
        if upstream is None:
            total = 0.67 * out[0] + 0.25 * out[2]
            if out[4].numel():
                total = total + 0.63 * out[4].mean()
        else:
            total = upstream(out)
        total.backward()
        res["d_code"] = c.grad.cpu().numpy()
        res["d_code_pos"] = cp.grad.cpu().numpy()
    torch.cuda.synchronize()
    return res


def test_library_loaded_is_the_in_tree_hip_extension():
    lib = capi.load()
    assert "stego_amd/lib/libstego_corr.so" in capi.library_path()
    assert lib.stego_abi_version() == 7
    assert torch.cuda.is_available()


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("layout", ["nchw", "cl"])
@pytest.mark.parametrize("name", ALL_CASES)
def test_forward_backward_match_reference_golden(name, layout, precision):
    c = GoldenCase(name)
    g = c.g
    r = _run(c.inputs, c.perms, c.cfg, layout=layout, precision=precision)
    out = r["out"]
    scale = float(np.mean(np.abs(g["neg_inter_loss"]))) if c.n_neg else float(np.mean(np.abs(g["pos_inter_cd"])))
    assert abs(float(out[0]) - float(g["pos_intra_loss"])) <= 1e-3 * scale + 1e-3 * abs(float(g["pos_intra_loss"]))
    assert abs(float(out[2]) - float(g["pos_inter_loss"])) <= 1e-3 * scale + 1e-3 * abs(float(g["pos_inter_loss"]))
    # atol: the loss multiplies a cosine by (fd - shift), which cancels to ~0 where fd ~ shift; fp32
    # accumulation noise on fd is ~1e-6 absolute.  The split-fp16 mode is held to the SAME bar as the fp32 MFMA
    # (its products carry 22 bits; bound + adversarial inputs: test_split_fp16_error_bound_on_adversarial_inputs)
    la = 5e-4
    assert_close(c.sub(out[1]), g["pos_intra_cd"], atol_frac=la, what="pos_intra_cd")
    assert_close(c.sub(out[3]), g["pos_inter_cd"], atol_frac=la, what="pos_inter_cd")
    assert_close(c.sub(out[4]), g["neg_inter_loss"], atol_frac=la, what="neg_inter_loss")
    assert_close(c.sub(out[5]), g["neg_inter_cd"], atol_frac=la, what="neg_inter_cd")
    S = c.S
    assert out[1].shape == (c.B, S, S, S, S) and out[4].shape == (c.n_neg * c.B, S, S, S, S)
    # backward vs reference autograd (atomics reorder sums: slightly looser atol)
    assert_close(c.sub(r["d_code"]), g["d_code_train"], rtol=1e-3, atol_frac=1e-3, what="d_code")
    assert_close(c.sub(r["d_code_pos"]), g["d_code_pos_train"], rtol=1e-3, atol_frac=1e-3, what="d_code_pos")
    np.testing.assert_allclose(np.linalg.norm(r["d_code"].astype(np.float64)), g["d_code_train_norm"], rtol=1e-4)


@pytest.mark.parametrize("name", ["small_default", "small_noclamp_stab", "small_stab", "small_dinolike_S11"])
def test_backward_general_upstream(name):
    """Upstream gradients on every output (scalars, neg loss tensor, the three cd tensors)."""
    c = GoldenCase(name)
    g = c.g
    u = {k: _dev(g[k]) for k in ("u_neg_loss", "u_intra_cd", "u_inter_cd", "u_neg_cd")}

    def upstream(out):
        return 1.3 * out[0] - 0.7 * out[2] + (out[4].reshape(-1) * u["u_neg_loss"]).sum() + \
            (out[1].reshape(-1) * u["u_intra_cd"]).sum() + (out[3].reshape(-1) * u["u_inter_cd"]).sum() + \
            (out[5].reshape(-1) * u["u_neg_cd"]).sum()

    r = _run(c.inputs, c.perms, c.cfg, upstream=upstream)
    assert_close(r["d_code"], g["d_code_gen"], rtol=1e-3, atol_frac=1e-3, what="d_code_gen")
    assert_close(r["d_code_pos"], g["d_code_pos_gen"], rtol=1e-3, atol_frac=1e-3, what="d_code_pos_gen")


@pytest.mark.parametrize("scale", [1e-6, 3e4])
@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_feature_scale_invariance(scale, precision):
    """norm() makes the loss independent of the magnitude of the feature maps (modules.py:275-276).  The split-fp16
    mode stages raw samples as fp16 halves, so it must rescale them itself: 1e-6 would vanish, 3e4 overflow."""
    c = GoldenCase("cfg1_B4_vits8_dinolike")
    base = _run(c.inputs, c.perms, c.cfg, layout="cl", grad=False, precision=precision)["out"]
    scaled = dict(c.inputs)
    scaled["feats"] = c.inputs["feats"] * np.float32(scale)
    scaled["feats_pos"] = c.inputs["feats_pos"] * np.float32(scale)
    alt = _run(scaled, c.perms, c.cfg, layout="cl", grad=False, precision=precision)["out"]
    for x, y in zip(base, alt):
        np.testing.assert_allclose(x, y, rtol=1e-4, atol=2e-6)
    assert_close(c.sub(alt[4]), c.g["neg_inter_loss"], atol_frac=5e-4, what="neg_inter_loss (features x %g)" % scale)


def test_generic_strided_path_direct_through_capi():
    """modules.py re-lays large NCHW maps out channels-last (as_channels_last); the kernels' generic strided path
    is still part of the C ABI: call it directly with NCHW-contiguous maps and compare with the channels-last run."""
    d = O.synth_inputs(4, 192, 14, 14, 70, 11, 2, seed=31, dino_like=True)
    cfg = O.CorrCfg(neg_samples=2)
    t = {k: _dev(v) for k, v in d.items() if k != "perms"}
    perms = _dev(d["perms"])
    desc = capi.make_desc(4, 192, 70, 14, 14, 11, 2, cfg, (cfg.pos_intra_shift, cfg.pos_inter_shift, cfg.neg_inter_shift),
                          capi.PREC_F32)
    assert t["feats"].numel() >= 1 << 16 and t["feats"].stride(1) != 1            # would be re-laid out by modules.py
    a = capi.corr_fwd(desc, t["feats"], t["feats_pos"], t["code"], t["code_pos"], t["coords1"], t["coords2"], perms, False)
    cl = [_channels_last(t[k]) for k in ("feats", "feats_pos", "code", "code_pos")]
    b = capi.corr_fwd(desc, *cl, t["coords1"], t["coords2"], perms, False)
    for x, y in zip(a[:5], b[:5]):
        np.testing.assert_allclose(x.cpu().numpy(), y.cpu().numpy(), rtol=1e-5, atol=2e-6)
    ref = O.corr_loss_forward(**{k: d[k] for k in ("feats", "feats_pos", "code", "code_pos", "coords1", "coords2")},
                              perms=d["perms"], cfg=cfg)
    assert_close(a[3].cpu().numpy().reshape(ref.neg_inter_loss.shape), ref.neg_inter_loss, atol_frac=5e-4, what="neg_loss")


_ORACLE_CACHE = {}


def _full_size_oracle(inputs, perms, cfg):
    if "fwd" not in _ORACLE_CACHE:
        _ORACLE_CACHE["fwd"] = O.corr_loss_forward(**inputs, perms=perms, cfg=cfg)
    return _ORACLE_CACHE["fwd"]


def _full_size_oracle_grads(inputs, perms, cfg):
    if "bwd" not in _ORACLE_CACHE:
        B, S, n_neg = inputs["feats"].shape[0], cfg.feature_samples, cfg.neg_samples
        g_nl = np.full((n_neg * B, S, S, S, S), 0.63 / (n_neg * B * S ** 4))
        _ORACLE_CACHE["bwd"] = O.corr_loss_backward(**inputs, perms=perms, cfg=cfg, g_intra=0.67, g_inter=0.25,
                                                    g_neg_loss=g_nl)
    return _ORACLE_CACHE["bwd"]


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_full_size_cfg2_against_fp64_oracle(precision):
    """BASELINE config 2 (B=32, ViT-S/8 224^2: C=384, 28x28, K=70, S=11, 5 negatives), channels-last."""
    B, C, H, W, K, S, n_neg = 32, 384, 28, 28, 70, 11, 5
    d = O.synth_inputs(B, C, H, W, K, S, n_neg, seed=2024, dino_like=True)
    cfg = O.CorrCfg()
    inputs = {k: d[k] for k in ("feats", "feats_pos", "code", "code_pos", "coords1", "coords2")}
    r = _run(inputs, d["perms"], cfg, layout="cl", precision=precision)
    ref = _full_size_oracle(inputs, d["perms"], cfg)
    out = r["out"]
    la = 5e-4                        # one bar for both arithmetic modes
    assert_close(out[1], ref.pos_intra_cd, atol_frac=la, what="intra_cd")
    assert_close(out[3], ref.pos_inter_cd, atol_frac=la, what="inter_cd")
    assert_close(out[4], ref.neg_inter_loss, atol_frac=la, what="neg_loss")
    assert_close(out[5], ref.neg_inter_cd, atol_frac=la, what="neg_cd")
    scale = float(np.abs(ref.neg_inter_loss).mean())
    assert abs(float(out[0]) - float(ref.pos_intra_loss)) < 1e-3 * scale
    assert abs(float(out[2]) - float(ref.pos_inter_loss)) < 1e-3 * scale
    dc, dcp = _full_size_oracle_grads(inputs, d["perms"], cfg)
    assert_close(r["d_code"], dc, rtol=1e-3, atol_frac=1e-3, what="d_code")
    assert_close(r["d_code_pos"], dcp, rtol=1e-3, atol_frac=1e-3, what="d_code_pos")
    # size-independent properties of the path
    icd = out[1].reshape(B, S * S, S * S)
    pa = 2e-6
    np.testing.assert_allclose(icd, icd.transpose(0, 2, 1), atol=pa)              # intra cd is symmetric
    np.testing.assert_allclose(np.diagonal(icd, axis1=1, axis2=2), 1.0, atol=5 * pa)  # unit self-similarity
    assert np.abs(out[5]).max() <= 1.0 + 5 * pa                                   # cosines


_CFG4_CACHE = {}


def _cfg4_case():
    """BASELINE config 4 at its real size: ViT-B/8 at 320^2 -> B=32, C=768, 40x40 map, K=70, S=11, 5 negatives
    (DINO-like values, so the clamp / shift branches are all exercised).  fp64 oracle forward + backward, once."""
    if not _CFG4_CACHE:
        B, C, H, W, K, S, n_neg = 32, 768, 40, 40, 70, 11, 5
        d = O.synth_inputs(B, C, H, W, K, S, n_neg, seed=78, dino_like=True)
        cfg = O.CorrCfg()
        inputs = {k: d[k] for k in ("feats", "feats_pos", "code", "code_pos", "coords1", "coords2")}
        ref = O.corr_loss_forward(**inputs, perms=d["perms"], cfg=cfg)
        g_nl = np.full((n_neg * B, S, S, S, S), 0.63 / (n_neg * B * S ** 4))
        grads = O.corr_loss_backward(**inputs, perms=d["perms"], cfg=cfg, g_intra=0.67, g_inter=0.25, g_neg_loss=g_nl)
        _CFG4_CACHE.update(inputs=inputs, perms=d["perms"], cfg=cfg, ref=ref, grads=grads)
    return _CFG4_CACHE


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_full_size_cfg4_vitb_b32_against_fp64_oracle(precision):
    """BASELINE config 4 (B=32, C=768, 40x40), channels-last, forward AND backward, both arithmetic modes."""
    c = _cfg4_case()
    r = _run(c["inputs"], c["perms"], c["cfg"], layout="cl", precision=precision)
    ref, out = c["ref"], r["out"]
    la = 5e-4
    assert_close(out[1], ref.pos_intra_cd, atol_frac=la, what="intra_cd")
    assert_close(out[3], ref.pos_inter_cd, atol_frac=la, what="inter_cd")
    assert_close(out[4], ref.neg_inter_loss, atol_frac=la, what="neg_loss")
    assert_close(out[5], ref.neg_inter_cd, atol_frac=la, what="neg_cd")
    scale = float(np.abs(ref.neg_inter_loss).mean())
    assert abs(float(out[0]) - float(ref.pos_intra_loss)) < 1e-3 * scale
    assert abs(float(out[2]) - float(ref.pos_inter_loss)) < 1e-3 * scale
    dc, dcp = c["grads"]
    assert_close(r["d_code"], dc, rtol=1e-3, atol_frac=1e-3, what="d_code")
    assert_close(r["d_code_pos"], dcp, rtol=1e-3, atol_frac=1e-3, what="d_code_pos")


def test_split_fp16_error_bound_on_adversarial_inputs():
    """The f16x3 mode replaces each fp32 product a*b by ah*bh + ah*bl + al*bh with fp16 halves (x = xh + xl + e,
    |e| <= 2^-22 |x| after the per-point power-of-two prescale) accumulated in fp32.  Bound: for L2-normalised
    vectors |fd_split - fd_exact| <= (3 * 2^-22 + 2^-22) * sum|a_c b_c| + fp32 accumulation error <= ~1e-6,
    the same class as the fp32 MFMA chain (~C * 2^-24 * sum|a_c b_c|).  Checked where the split is stressed:
    (i) dynamic range > 2^11 inside a point (small channels fall into fp16-subnormal lo halves),
    (ii) globally tiny / huge feature magnitudes, (iii) channels that are exactly zero, (iv) one dominant channel.
    Both modes must stay within 2.5e-6 of the fp64 oracle on every fd-derived output, and f16x3 within 2x of f32."""
    B, C, H, W, K, S, n_neg = 4, 384, 12, 12, 70, 11, 2
    d = O.synth_inputs(B, C, H, W, K, S, n_neg, seed=404, dino_like=True)
    rng = np.random.default_rng(404)
    f = d["feats"].copy()
    fp = d["feats_pos"].copy()
    spread = np.exp2(rng.integers(-14, 1, size=(1, C, 1, 1))).astype(np.float32)        # per-channel 2^-14 .. 1
    f[0] *= spread[0]
    fp[0] *= spread[0]
    f[1] *= np.float32(3e-7)                                                            # tiny overall magnitude
    fp[1] *= np.float32(2e4)                                                            # huge overall magnitude
    f[2, ::3] = 0.0                                                                     # exact zeros
    f[3, 5] += 500.0                                                                    # one dominant channel
    fp[3, 5] -= 500.0
    inputs = dict(feats=f, feats_pos=fp, code=d["code"], code_pos=d["code_pos"], coords1=d["coords1"], coords2=d["coords2"])
    cfg = O.CorrCfg(neg_samples=n_neg)
    ref = O.corr_loss_forward(**inputs, perms=d["perms"], cfg=cfg)
    err = {}
    for precision in ("f32", "f16x3"):
        out = _run(inputs, d["perms"], cfg, layout="cl", grad=False, precision=precision)["out"]
        # the loss is -clamp(cd) * (fd - shift) with |clamp(cd)| <= 1: its error is the fd error
        e = np.abs(out[4].astype(np.float64) - ref.neg_inter_loss).max()
        err[precision] = e
        assert e < 2.5e-6, (precision, e)
        assert_close(out[4], ref.neg_inter_loss, atol_frac=5e-4, what="neg_loss %s" % precision)
    assert err["f16x3"] <= 2.0 * err["f32"] + 2e-7, err


def test_split_fp16_code_correlation_on_adversarial_codes():
    """Round 2b: in f16x3 mode the fused forward also multiplies the CODE K-chunks as fp16 hi/lo halves (the MFMA team splits the
    fp32 operands in registers; raw B-side codes get a per-point power-of-two prescale from their first chunk).  cd is a cosine:
    both modes must stay within 2.5e-6 of the fp64 oracle on every cd output where the split is stressed - per-channel dynamic
    range 2^14 inside a point, tiny / huge code magnitudes, exact zeros, one dominant channel - and f16x3 within 2x of f32."""
    B, C, H, W, K, S, n_neg = 4, 384, 12, 12, 70, 11, 2
    d = O.synth_inputs(B, C, H, W, K, S, n_neg, seed=405, dino_like=True)
    rng = np.random.default_rng(405)
    c = d["code"].copy()
    cp = d["code_pos"].copy()
    spread = np.exp2(rng.integers(-14, 1, size=(1, K, 1, 1))).astype(np.float32)        # per-channel 2^-14 .. 1
    c[0] *= spread[0]
    cp[0] *= spread[0]
    c[1] *= np.float32(3e-7)                                                            # tiny overall magnitude
    cp[1] *= np.float32(2e4)                                                            # huge overall magnitude
    c[2, ::3] = 0.0                                                                     # exact zeros
    c[3, 5] += 50.0                                                                     # one dominant channel
    cp[3, 40] -= 50.0                                                                   # ... in another K-chunk
    inputs = dict(feats=d["feats"], feats_pos=d["feats_pos"], code=c, code_pos=cp, coords1=d["coords1"], coords2=d["coords2"])
    cfg = O.CorrCfg(neg_samples=n_neg)
    ref = O.corr_loss_forward(**inputs, perms=d["perms"], cfg=cfg)
    err = {}
    for precision in ("f32", "f16x3"):
        r = _run(inputs, d["perms"], cfg, layout="cl", grad=False, precision=precision)
        out = r["out"]
        e = max(np.abs(out[1].astype(np.float64) - ref.pos_intra_cd).max(), np.abs(out[3].astype(np.float64) - ref.pos_inter_cd).max(),
                np.abs(out[5].astype(np.float64) - ref.neg_inter_cd).max())
        err[precision] = e
        assert np.isfinite(out[5]).all()
        assert e < 2.5e-6, (precision, e)
    assert err["f16x3"] <= 2.0 * err["f32"] + 2e-7, err


@pytest.mark.parametrize("shape", [
    dict(B=1, C=8, H=4, W=4, K=4, S=1, n_neg=1),        # single sample point, B=1 (perm = [0])
    dict(B=2, C=5, H=3, W=9, K=3, S=2, n_neg=3),        # odd channel counts -> scalar gather path
    dict(B=5, C=130, H=7, W=6, K=66, S=7, n_neg=2),     # C, K straddle the 64-wide chunk
    dict(B=3, C=64, H=1, W=1, K=72, S=3, n_neg=1),      # 1x1 map (every tap clamps), K at the limit
    dict(B=2, C=16, H=5, W=5, K=2, S=11, n_neg=0),      # no negatives
    dict(B=2, C=8, H=3, W=70, K=6, S=4, n_neg=1),       # W > 64: the backward's band (LDS) unsample fallback
    dict(B=2, C=64, H=40, W=40, K=70, S=11, n_neg=2),   # 32 < W <= 64: 4 pixel tiles per row in the unsample kernel
])
@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_edge_shapes(shape, precision):
    d = O.synth_inputs(seed=5, **shape)
    cfg = O.CorrCfg(feature_samples=shape["S"], neg_samples=shape["n_neg"])
    inputs = {k: d[k] for k in ("feats", "feats_pos", "code", "code_pos", "coords1", "coords2")}
    r = _run(inputs, d["perms"], cfg, precision=precision)
    ref = O.corr_loss_forward(**inputs, perms=d["perms"], cfg=cfg)
    la = 5e-4
    assert_close(r["out"][1], ref.pos_intra_cd, atol_frac=la, what="intra_cd")
    assert_close(r["out"][3], ref.pos_inter_cd, atol_frac=la, what="inter_cd")
    assert_close(r["out"][4], ref.neg_inter_loss, atol_frac=la, what="neg_loss")
    numel = shape["B"] * shape["S"] ** 4
    g_nl = None
    if shape["n_neg"]:
        g_nl = np.full(ref.neg_inter_loss.shape, 0.63 / (shape["n_neg"] * numel))
    dc, dcp = O.corr_loss_backward(**inputs, perms=d["perms"], cfg=cfg, g_intra=0.67, g_inter=0.25, g_neg_loss=g_nl)
    assert_close(r["d_code"], dc, rtol=1e-3, atol_frac=1e-3, what="d_code")
    assert_close(r["d_code_pos"], dcp, rtol=1e-3, atol_frac=1e-3, what="d_code_pos")


def test_border_coords_zero_vectors_and_duplicate_perm():
    """coords outside [-1,1] (border clip), exactly +-1, all-zero feature/code vectors (eps branch
    of normalize), and a perm with duplicates (super_perm can produce them)."""
    B, C, H, W, K, S, n_neg = 4, 32, 6, 6, 8, 5, 2
    d = O.synth_inputs(B, C, H, W, K, S, n_neg, seed=9)
    rng = np.random.default_rng(1)
    d["coords1"] = (rng.random((B, S, S, 2)) * 2.6 - 1.3).astype(np.float32)
    d["coords2"][0, 0, 0] = [1.0, 1.0]
    d["coords2"][0, 0, 1] = [-1.0, 1.0]
    d["coords2"][0, 1, 0] = [1.0, -1.0]
    d["feats"][1] = 0.0
    d["code_pos"][2] = 0.0
    d["perms"] = np.array([[1, 1, 3, 0], [2, 0, 0, 1]], dtype=np.int64)
    cfg = O.CorrCfg(feature_samples=S, neg_samples=n_neg)
    inputs = {k: d[k] for k in ("feats", "feats_pos", "code", "code_pos", "coords1", "coords2")}
    r = _run(inputs, d["perms"], cfg)
    ref = O.corr_loss_forward(**inputs, perms=d["perms"], cfg=cfg)
    for i, name in ((1, "pos_intra_cd"), (3, "pos_inter_cd"), (4, "neg_inter_loss"), (5, "neg_inter_cd")):
        assert np.isfinite(r["out"][i]).all(), name
        assert_close(r["out"][i], getattr(ref, name), what=name)
    numel = B * S ** 4
    g_nl = np.full(ref.neg_inter_loss.shape, 0.63 / (n_neg * numel))
    dc, dcp = O.corr_loss_backward(**inputs, perms=d["perms"], cfg=cfg, g_intra=0.67, g_inter=0.25, g_neg_loss=g_nl)
    assert np.isfinite(r["d_code"]).all() and np.isfinite(r["d_code_pos"]).all()
    assert_close(r["d_code"], dc, rtol=1e-3, atol_frac=1e-3, what="d_code")
    # code_pos[2] == 0 puts its samples on the eps branch (grad = g/eps, huge but finite): compare the rest
    keep = [0, 1, 3]
    assert_close(r["d_code_pos"][keep], dcp[keep], rtol=1e-3, atol_frac=1e-3, what="d_code_pos")


def test_helper_matches_oracle_and_reference_semantics():
    """ContrastiveCorrelationLoss.helper on pre-sampled tensors (modules.py:325-347), S1 != S2."""
    rng = np.random.default_rng(3)
    N, C, K, S1, S2 = 3, 48, 10, 5, 7
    f1, f2 = (rng.standard_normal((N, C, S1, S2)).astype(np.float32) for _ in range(2))
    c1, c2 = (rng.standard_normal((N, K, S1, S2)).astype(np.float32) for _ in range(2))
    for kw in (dict(), dict(pointwise=False), dict(zero_clamp=False, stabalize=True)):
        cfg = O.CorrCfg(**kw)
        tc1 = _dev(c1).requires_grad_(True)
        tc2 = _dev(c2).requires_grad_(True)
        loss, cd = M.ContrastiveCorrelationLoss(cfg).helper(_dev(f1), _dev(f2), tc1, tc2, 0.31)
        el, ecd, efd = O.helper(f1.astype(np.float64), f2.astype(np.float64), c1.astype(np.float64),
                                c2.astype(np.float64), 0.31, cfg)
        assert tuple(loss.shape) == (N, S1, S2, S1, S2)
        assert_close(loss.detach().cpu().numpy(), el, what="helper loss %s" % kw)
        assert_close(cd.detach().cpu().numpy(), ecd, what="helper cd %s" % kw)
        u = rng.standard_normal(el.shape) / el.size
        v = rng.standard_normal(el.shape) / el.size
        ((loss * _dev(u.astype(np.float32))).sum() + (cd * _dev(v.astype(np.float32))).sum()).backward()
        ga, gb = O._helper_bwd_codes(c1.astype(np.float64), c2.astype(np.float64), efd, ecd, 0.31, cfg, u, v)
        assert_close(tc1.grad.cpu().numpy(), ga, rtol=1e-3, atol_frac=1e-3, what="helper d_c1 %s" % kw)
        assert_close(tc2.grad.cpu().numpy(), gb, rtol=1e-3, atol_frac=1e-3, what="helper d_c2 %s" % kw)


def test_forward_is_deterministic_and_batch_equivariant():
    B, C, H, W, K, S, n_neg = 8, 64, 10, 10, 12, 6, 2
    d = O.synth_inputs(B, C, H, W, K, S, n_neg, seed=21)
    cfg = O.CorrCfg(feature_samples=S, neg_samples=n_neg)
    inputs = {k: d[k] for k in ("feats", "feats_pos", "code", "code_pos", "coords1", "coords2")}
    a = _run(inputs, d["perms"], cfg, grad=False)["out"]
    b = _run(inputs, d["perms"], cfg, grad=False)["out"]
    for x, y in zip(a, b):
        assert np.array_equal(x, y)            # no atomics in the forward: bitwise repeatable
    # relabelling the batch permutes the per-image outputs (reductions are per image except old_mean)
    sigma = np.array([3, 0, 7, 1, 6, 2, 5, 4])
    inv = np.argsort(sigma)
    inputs2 = {k: v[sigma] for k, v in inputs.items()}
    perms2 = inv[d["perms"][:, sigma]]
    c = _run(inputs2, perms2, cfg, grad=False)["out"]
    np.testing.assert_allclose(c[1], a[1][sigma], atol=1e-6)
    np.testing.assert_allclose(c[5].reshape(n_neg, B, -1), a[5].reshape(n_neg, B, -1)[:, sigma], atol=1e-6)
    np.testing.assert_allclose(c[4].reshape(n_neg, B, -1), a[4].reshape(n_neg, B, -1)[:, sigma], atol=2e-6)


def test_backward_is_linear_in_upstream():
    B, C, H, W, K, S, n_neg = 4, 32, 8, 8, 16, 5, 2
    d = O.synth_inputs(B, C, H, W, K, S, n_neg, seed=33)
    cfg = O.CorrCfg(feature_samples=S, neg_samples=n_neg)
    inputs = {k: d[k] for k in ("feats", "feats_pos", "code", "code_pos", "coords1", "coords2")}
    g1 = _run(inputs, d["perms"], cfg, upstream=lambda o: o[0])
    g2 = _run(inputs, d["perms"], cfg, upstream=lambda o: o[2] + o[4].sum())
    g3 = _run(inputs, d["perms"], cfg, upstream=lambda o: 2.0 * o[0] - 3.0 * (o[2] + o[4].sum()))
    # (g1 has scalar upstreams only and runs on the lists-first unsample, g2 / g3 carry a dense upstream and run on the row kernel: two
    # fixed summation orders - the bar is fp32 rounding of a ~12-term sum relative to the largest gradient, not 1e-7 absolute)
    for k in ("d_code", "d_code_pos"):
        want = 2.0 * g1[k] - 3.0 * g2[k]
        np.testing.assert_allclose(g3[k], want, rtol=1e-4, atol=4e-6 * float(np.abs(want).max()))


def test_on_device_rng_path_equals_explicit_draws():
    """forward() draws coords/perms on the device generator in the reference's order; replaying the
    same seed by hand and calling forward_explicit must give identical outputs."""
    B, C, H, W, K = 6, 32, 8, 8, 10
    cfg = O.CorrCfg(feature_samples=4, neg_samples=3)
    g = torch.Generator(device=DEV).manual_seed(5)
    f = torch.randn(B, C, H, W, device=DEV, generator=g)
    fp = torch.randn(B, C, H, W, device=DEV, generator=g)
    c = torch.randn(B, K, H, W, device=DEV, generator=g)
    cp = torch.randn(B, K, H, W, device=DEV, generator=g)
    loss = M.ContrastiveCorrelationLoss(cfg)
    torch.manual_seed(99)
    a = loss(f, fp, None, None, c, cp)
    torch.manual_seed(99)
    coords1 = torch.rand(B, 4, 4, 2, device=DEV) * 2 - 1
    coords2 = torch.rand(B, 4, 4, 2, device=DEV) * 2 - 1
    perms = torch.stack([M.super_perm(B, f.device) for _ in range(3)])
    b = loss.forward_explicit(f, fp, c, cp, coords1, coords2, perms)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert not (perms == torch.arange(B, device=DEV)).any()


def test_salience_path_on_device_matches_oracle():
    """cfg.use_salience (modules.py:355-365): coordinates come from the salience maps (torch, reference order) and go
    through the same kernels; compared with the CPU oracle fed the very same coordinates."""
    B, C, H, W, K = 4, 64, 10, 10, 12
    cfg = O.CorrCfg(feature_samples=5, neg_samples=2)
    cfg.use_salience = True
    g = torch.Generator(device=DEV).manual_seed(8)
    f = torch.randn(B, C, H, W, device=DEV, generator=g)
    fp = torch.randn(B, C, H, W, device=DEV, generator=g)
    c = torch.randn(B, K, H, W, device=DEV, generator=g)
    cp = torch.randn(B, K, H, W, device=DEV, generator=g)
    sal = (torch.rand(B, 40, 40, device=DEV, generator=g) > 0.6).float()
    sal_pos = (torch.rand(B, 40, 40, device=DEV, generator=g) > 0.6).float()
    sal[2] = 0
    loss = M.ContrastiveCorrelationLoss(cfg)
    torch.manual_seed(21)
    out = loss(f, fp, sal, sal_pos, c, cp)
    torch.manual_seed(21)
    coords1, coords2 = loss.draw_coords(f, sal, sal_pos)
    perms = torch.stack([M.super_perm(B, f.device) for _ in range(2)])
    assert coords1.abs().max() <= 1 and coords2.abs().max() <= 1
    ref = O.corr_loss_forward(f.cpu().numpy(), fp.cpu().numpy(), c.cpu().numpy(), cp.cpu().numpy(), coords1.cpu().numpy(),
                              coords2.cpu().numpy(), perms.cpu().numpy(), cfg)
    assert_close(out[1].cpu().numpy(), ref.pos_intra_cd, what="pos_intra_cd")
    assert_close(out[3].cpu().numpy(), ref.pos_inter_cd, what="pos_inter_cd")
    assert_close(out[4].cpu().numpy(), ref.neg_inter_loss, what="neg_inter_loss")
    assert_close(out[5].cpu().numpy(), ref.neg_inter_cd, what="neg_inter_cd")


@pytest.mark.parametrize("S,K,C,layout", [(12, 70, 384, "cl"), (13, 70, 384, "cl"), (16, 70, 384, "cl"), (16, 70, 768, "cl"), (16, 24, 64, "cl"),
                                          (15, 88, 384, "cl"), (14, 70, 192, "nchw"), (16, 101, 384, "cl"), (13, 128, 64, "cl"), (12, 89, 64, "nchw"), (5, 130, 384, "cl"),
                                          (3, 96, 16, "cl")])
def test_feature_samples_above_11_and_wide_codes(S, K, C, layout):
    """cfg.feature_samples and cfg.dim are free in the reference (train_config.yml:39,51).  feature_samples 12 .. 16 (144 .. 256 points per
    image, any K <= 128) run on the multi-launch kernels of csrc/corr_wide.hip behind stego_corr_fwd / _bwd - the same entry points as S <= 11 (maps of one pixel row or column, any layout);
    beyond those limits (K > 128, or K > 72 on maps the S <= 11 kernels do not take) generic_forward computes the loss (native samplers + dense-correlation kernel +
    elementwise launches, gradient through autograd).  Forward and gradients against the fp64 oracle, any map layout."""
    B, H, W, n_neg = 3, 10, 9, 2
    d = O.synth_inputs(B, C, H, W, K, S, n_neg, seed=900 + S + K, dino_like=True)
    cfg = O.CorrCfg(feature_samples=S, neg_samples=n_neg)
    native = S * S > 128 and K <= 128
    assert M.ContrastiveCorrelationLoss.fused_kernels_cover(B, C, K, H, W, S) == native
    inputs = {k: d[k] for k in ("feats", "feats_pos", "code", "code_pos", "coords1", "coords2")}
    if native:
        tt = {k: torch.from_numpy(v).to(DEV) for k, v in inputs.items()}
        desc = capi.make_desc(B, C, K, H, W, S, n_neg, cfg, (cfg.pos_intra_shift, cfg.pos_inter_shift, cfg.neg_inter_shift), capi.PREC_F16X3)
        assert capi.corr_fwd_launches(desc, tt["feats"], tt["feats_pos"], tt["code"], tt["code_pos"]) == 8
    r = _run(inputs, d["perms"], cfg, layout=layout, precision="f16x3")
    ref = O.corr_loss_forward(**inputs, perms=d["perms"], cfg=cfg)
    assert_close(r["out"][1], ref.pos_intra_cd, atol_frac=5e-4, what="intra_cd")
    assert_close(r["out"][3], ref.pos_inter_cd, atol_frac=5e-4, what="inter_cd")
    assert_close(r["out"][4], ref.neg_inter_loss, atol_frac=5e-4, what="neg_loss")
    assert_close(r["out"][5], ref.neg_inter_cd, atol_frac=5e-4, what="neg_cd")
    scale = float(np.mean(np.abs(ref.neg_inter_loss)))
    assert abs(float(r["out"][0]) - float(ref.pos_intra_loss)) <= 1e-3 * scale + 1e-3 * abs(float(ref.pos_intra_loss))
    assert abs(float(r["out"][2]) - float(ref.pos_inter_loss)) <= 1e-3 * scale + 1e-3 * abs(float(ref.pos_inter_loss))
    numel = B * S ** 4
    g_nl = np.full((n_neg * B,) + (S,) * 4, 0.63 / (n_neg * numel))
    dc, dcp = O.corr_loss_backward(**inputs, perms=d["perms"], cfg=cfg, g_intra=0.67, g_inter=0.25, g_neg_loss=g_nl)
    assert_close(r["d_code"], dc, rtol=2e-3, atol_frac=1e-3, what="d_code")
    assert_close(r["d_code_pos"], dcp, rtol=2e-3, atol_frac=1e-3, what="d_code_pos")


@pytest.mark.parametrize("native_backbone,S", [(False, 5), (True, 5), (True, 12)])
def test_training_loop_on_device_matches_cpu_oracle_step(native_backbone, S):
    """The drop-in surface end to end on the MI355X: LitUnsupervisedSegmenter.training_step (reference
    train_segmentation.py:112-245) with the HIP loss inside, against the same step computed on CPU with the
    oracle-backed backend double (same weights, same batch, same RNG draws fed explicitly)."""
    import warnings
    import oracle_backend
    from stego_amd.train_segmentation import LitUnsupervisedSegmenter, SyntheticContrastiveDataset, load_config
    warnings.filterwarnings("ignore", message="DinoFeaturizer")
    # (S = 12: the multi-launch loss path of csrc/corr_wide.hip inside the same training step)
    ov = ["model_type=vit_tiny", "dino_patch_size=16", "res=64", "batch_size=4", "feature_samples=%d" % S, "neg_samples=2",
          "dim=10", "dropout=False", "native_backbone=%s" % native_backbone]
    cfg = load_config(overrides=ov)
    # native_backbone=False isolates the loss path (fp32 torch backbone on both sides); True (the default) runs the whole device step
    # on the native kernels, backbone included (precision f16x3: the fp32 class, tests/test_vit_native.py) - the same 1e-3 bars
    tol = 1.0
    torch.manual_seed(0)
    ref = LitUnsupervisedSegmenter(27, cfg).cpu()            # DinoFeaturizer puts its backbone on the GPU when one exists (modules.py:32)
    ref.net.dropout.p = 0.0                                   # no dropout noise: CPU and GPU RNG streams differ
    dev_model = LitUnsupervisedSegmenter(27, cfg)
    dev_model.net.dropout.p = 0.0
    dev_model.load_state_dict(ref.state_dict())
    dev_model.to(DEV)
    ref_w0 = ref.net.cluster1[0].weight.detach().clone()
    ds = SyntheticContrastiveDataset(4, cfg.res, 27)
    batch = torch.utils.data.default_collate([ds[i] for i in range(4)])
    # identical RNG draws on both sides
    g = torch.Generator().manual_seed(5)
    coords1 = torch.rand(4, S, S, 2, generator=g) * 2 - 1
    coords2 = torch.rand(4, S, S, 2, generator=g) * 2 - 1
    perms = torch.tensor([[1, 2, 3, 0], [2, 3, 0, 1]])

    def patched_draw(dev):            # (training_step goes through loss.total() -> loss.draw(): feed the same draws to both sides)
        return lambda of, s1, s2: (coords1.to(dev), coords2.to(dev), perms.to(dev))

    ref.contrastive_corr_loss_fn.draw = patched_draw("cpu")
    dev_model.contrastive_corr_loss_fn.draw = patched_draw(DEV)
    try:
        M._backend = oracle_backend
        loss_ref = ref.training_step(batch, 0)
    finally:
        M._backend = capi
    loss_dev = dev_model.training_step({k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}, 0)
    assert dev_model.net.backbone_path == ("native" if native_backbone else "torch")
    assert abs(float(loss_dev) - float(loss_ref)) < tol * 2e-3 * max(1.0, abs(float(loss_ref)))
    for k in ("loss/pos_intra", "loss/pos_inter", "loss/neg_inter"):
        assert abs(float(dev_model.logged[k]) - float(ref.logged[k])) < tol * 1e-3 * max(0.05, abs(float(ref.logged[k]))), k
    w_ref = ref.net.cluster1[0].weight.detach()
    w_dev = dev_model.net.cluster1[0].weight.detach().cpu()
    assert torch.allclose(w_dev, w_ref, rtol=1e-3, atol=2e-4)          # one Adam step on the head, same direction
    d_ref, d_dev = w_ref - ref_w0, w_dev - ref_w0
    assert float(torch.nn.functional.cosine_similarity(d_dev.flatten(), d_ref.flatten(), dim=0)) > 0.999


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_fused_forward_stress_rotating_inputs_against_three_launch_path(precision):
    """The fused forward overlaps two wave teams through an LDS ring, in-launch hand-offs between workgroups and asynchronous
    copies.  A race there shows up rarely and only when consecutive launches see DIFFERENT data (stale bytes of an
    identical previous launch are the right bytes): rotate four input sets for 400 launches and compare every output with
    the three-launch path (separate sampling / tile / finalize kernels, no in-launch hand-off) of the same library."""
    import bench
    dev = torch.device("cuda:0")
    C, H, W, K = bench.WORKLOADS["vits8_224"]
    B, S, n_neg = 32, 11, 5
    cfg = bench.Cfg()
    sets = [bench.make_inputs(B, C, H, W, K, S, n_neg, 4000 + i, dev) for i in range(4)]
    prec = capi.PREC_F16X3 if precision == "f16x3" else capi.PREC_F32
    desc = capi.make_desc(B, C, K, H, W, S, n_neg, cfg, (.18, .12, .46), prec)

    def run(d):
        out = capi.corr_fwd(desc, d["feats"], d["feats_pos"], d["code"], d["code_pos"], d["coords1"], d["coords2"], d["perms"], True)
        torch.cuda.synchronize()
        return out

    try:
        capi.debug_set("STEGO_FWD_VARIANT", 1)
        refs = [[t.clone() for t in (o[0],) + tuple(o[1:5]) + (o[5][0],)] for o in (run(d) for d in sets)]
        capi.debug_set("STEGO_FWD_VARIANT", 0)
        worst = 0.0
        for rep in range(400):
            o = run(sets[rep % 4])
            got = (o[0],) + tuple(o[1:5]) + (o[5][0],)
            for g, r in zip(got, refs[rep % 4]):
                assert not torch.isnan(g).any()
                worst = max(worst, float((g - r).abs().max()))
            assert worst < 2e-6, (rep, worst)
    finally:
        capi.debug_set("STEGO_FWD_VARIANT", 0)


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_loss_curve_of_20_training_steps_overlaps_the_reference(precision):
    """north_star: 'loss-curve overlap within 1e-3 of upstream'.  tests/golden/loss_curve.npz holds the UNMODIFIED reference
    ContrastiveCorrelationLoss (modules.py:349-398) inside a 20-step Adam loop on the segmentation head
    (train_segmentation.py:163-181,391; oracle/make_golden.py:loss_curve_case) together with the RNG draws it made.  The same
    loop on the HIP path (fused forward + backward through the C ABI, torch Adam on the device) must log the same losses at
    every step and end with the same head."""
    from oracle.make_golden import LOSS_CURVE as p, loss_curve_inputs
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loss_curve.npz"))
    feats, feats_pos, cluster1, cluster2 = loss_curve_inputs(p)
    feats, feats_pos = feats.to(DEV), feats_pos.to(DEV)
    cluster1, cluster2 = cluster1.to(DEV), cluster2.to(DEV)
    cfg = O.CorrCfg(feature_samples=p["S"], neg_samples=p["n_neg"])
    cfg.corr_precision = precision
    loss_fn = M.ContrastiveCorrelationLoss(cfg)
    opt = torch.optim.Adam(list(cluster1.parameters()) + list(cluster2.parameters()), lr=p["lr"])
    curve = []
    for t in range(p["steps"]):
        opt.zero_grad()
        code = cluster1(feats) + cluster2(feats)
        code_pos = cluster1(feats_pos) + cluster2(feats_pos)
        out = loss_fn.forward_explicit(feats, feats_pos, code, code_pos, _dev(g["coords1"][t]), _dev(g["coords2"][t]), _dev(g["perms"][t]))
        pil, pel, nl = out[0].mean(), out[2].mean(), out[4].mean()
        loss = p["w_inter"] * pel + p["w_intra"] * pil + p["w_neg"] * nl
        loss.backward()
        opt.step()
        curve.append([float(pil.detach()), float(pel.detach()), float(nl.detach()), float(loss.detach())])
    curve = np.array(curve)
    ref = g["curve"]
    err = np.abs(curve - ref)
    assert err.max() < 1e-4, (int(err.argmax()) // 4, err.max())                  # each logged term, absolute (terms are O(0.1))
    assert (err[:, 3] / np.abs(ref[:, 3])).max() < 1e-3                           # the total loss, relative: the north_star bar
    w1 = cluster1[0].weight.detach().reshape(p["K"], p["C"]).cpu().numpy()
    w2 = cluster2[2].weight.detach().reshape(p["K"], p["C"]).cpu().numpy()
    for got, want in ((w1, g["final_cluster1_w"]), (w2, g["final_cluster2_out_w"])):
        assert np.linalg.norm(got - want) / np.linalg.norm(want) < 1e-3


def test_backward_wide_map_fallback_is_order_nondeterministic_but_bounded():
    """Maps wider than 64 pixels take the backward's band fallback (corr_unsample_kernel): contributions are appended to
    per-row worklists in arrival order (an LDS atomic hands out the slots; LDS fp32 atomics only on overflow), so the fp32
    summation ORDER may differ between runs - documented in include/stego_corr.h.  Bound: repeated runs agree to fp32
    rounding of the sum (<= 1e-6 of the largest gradient), and each matches the fp64 oracle at the usual bar.  Every other
    kernel of the path (and W <= 64, every BASELINE config) is bitwise repeatable (test_forward_is_deterministic...)."""
    shape = dict(B=3, C=16, H=5, W=72, K=10, S=6, n_neg=2)
    d = O.synth_inputs(seed=15, **shape)
    cfg = O.CorrCfg(feature_samples=shape["S"], neg_samples=shape["n_neg"])
    inputs = {k: d[k] for k in ("feats", "feats_pos", "code", "code_pos", "coords1", "coords2")}
    runs = [_run(inputs, d["perms"], cfg) for _ in range(4)]
    numel = shape["B"] * shape["S"] ** 4
    g_nl = np.full((shape["n_neg"] * shape["B"],) + (shape["S"],) * 4, 0.63 / (shape["n_neg"] * numel))
    dc, dcp = O.corr_loss_backward(**inputs, perms=d["perms"], cfg=cfg, g_intra=0.67, g_inter=0.25, g_neg_loss=g_nl)
    scale = float(np.abs(dc).max())
    for r in runs:
        assert_close(r["d_code"], dc, rtol=1e-3, atol_frac=1e-3, what="d_code")
        assert_close(r["d_code_pos"], dcp, rtol=1e-3, atol_frac=1e-3, what="d_code_pos")
        assert float(np.abs(r["d_code"] - runs[0]["d_code"]).max()) <= 1e-6 * scale
        assert float(np.abs(r["d_code_pos"] - runs[0]["d_code_pos"]).max()) <= 1e-6 * scale


@pytest.mark.parametrize("K", [96, 100, 128, 101, 127])
@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_code_dimensions_above_72_forward_and_backward_against_fp64_oracle(K, precision):
    """cfg.dim beyond the 70 the reference ships (train_config.yml:39; its ViT-B models use ~100): the fused forward walks four
    code K-chunks, the backward tile kernel two groups of channel tiles, the unsample kernel 8 channel tiles.  Channels-last
    ViT-width maps (what DinoFeaturizer emits); other layouts / widths keep the K <= 72 limit and say so."""
    B, C, H, W, S, n_neg = 3, 384, 9, 10, 7, 2
    d = O.synth_inputs(B, C, H, W, K, S, n_neg, seed=40 + K)
    cfg = O.CorrCfg(feature_samples=S, neg_samples=n_neg)
    inputs = {k: d[k] for k in ("feats", "feats_pos", "code", "code_pos", "coords1", "coords2")}
    r = _run(inputs, d["perms"], cfg, layout="cl", precision=precision)
    ref = O.corr_loss_forward(**inputs, perms=d["perms"], cfg=cfg)
    la = 5e-4
    assert_close(r["out"][1], ref.pos_intra_cd, atol_frac=la, what="intra_cd")
    assert_close(r["out"][3], ref.pos_inter_cd, atol_frac=la, what="inter_cd")
    assert_close(r["out"][4], ref.neg_inter_loss, atol_frac=la, what="neg_loss")
    assert_close(r["out"][5], ref.neg_inter_cd, atol_frac=la, what="neg_cd")
    assert abs(float(r["out"][0]) - float(ref.pos_intra_loss)) < 1e-3 * max(abs(float(ref.pos_intra_loss)), 1e-3)
    numel = B * S ** 4
    g_nl = np.full(ref.neg_inter_loss.shape, 0.63 / (n_neg * numel))
    dc, dcp = O.corr_loss_backward(**inputs, perms=d["perms"], cfg=cfg, g_intra=0.67, g_inter=0.25, g_neg_loss=g_nl)
    assert_close(r["d_code"], dc, rtol=1e-3, atol_frac=1e-3, what="d_code")
    assert_close(r["d_code_pos"], dcp, rtol=1e-3, atol_frac=1e-3, what="d_code_pos")
    # a general (dense) upstream on every output goes through the same kernels
    rng = np.random.default_rng(K)
    ups = [rng.standard_normal(np.shape(o)).astype(np.float32) for o in r["out"]]

    def upstream(out):
        return sum((o * _dev(u)).sum() for o, u in zip(out, ups))
    r2 = _run(inputs, d["perms"], cfg, layout="cl", precision=precision, upstream=upstream)
    dc2, dcp2 = O.corr_loss_backward(**inputs, perms=d["perms"], cfg=cfg, g_intra=float(ups[0]), g_intra_cd=ups[1], g_inter=float(ups[2]),
                                     g_inter_cd=ups[3], g_neg_loss=ups[4], g_neg_cd=ups[5])
    assert_close(r2["d_code"], dc2, rtol=1e-3, atol_frac=1e-3, what="d_code (dense upstream)")
    assert_close(r2["d_code_pos"], dcp2, rtol=1e-3, atol_frac=1e-3, what="d_code_pos (dense upstream)")


def test_code_dimensions_above_72_need_the_fused_layout():
    """72 < K <= 128 exists on the fused kernel only (channels-last ViT widths): the C ABI refuses other maps with STEGO_ERR_UNSUPPORTED,
    the Python surface routes them to generic_forward (parity: test_feature_samples_above_11_and_wide_codes_run_on_the_generic_path)."""
    cfg = O.CorrCfg(feature_samples=3, neg_samples=1)
    f = torch.randn(2, 16, 8, 8, device=DEV)                # not a ViT width: three-launch path, K <= 72 only
    c = torch.randn(2, 96, 8, 8, device=DEV)
    assert not M.ContrastiveCorrelationLoss.fused_kernels_cover(2, 16, 96, 8, 8, 3)
    desc = capi.make_desc(2, 16, 96, 8, 8, 3, 1, cfg, (.18, .12, .46), capi.PREC_F16X3)
    co = torch.rand(2, 3, 3, 2, device=DEV) * 2 - 1
    perms = torch.tensor([[1, 0]], device=DEV)
    with pytest.raises(RuntimeError, match="unsupported"):
        capi.corr_fwd(desc, f, f, c, c, co, co, perms, False)
    out = M.ContrastiveCorrelationLoss(cfg)(f, f, None, None, c, c)
    assert all(torch.isfinite(t).all() for t in out)


def test_unprepared_entry_point_accepts_any_workspace_and_prepared_one_stays_clean():
    """stego_corr_fwd() zeroes the in-launch hand-off words itself: a workspace full of 0xFF bytes gives the same outputs as the
    prepared path (stego_corr_workspace_prepare once + stego_corr_fwd_prepared), bit for bit; and every launch leaves the words
    zero again, so the prepared workspace serves call after call."""
    import ctypes
    import bench
    dev = torch.device(DEV)
    C, H, W, K = bench.WORKLOADS["vits8_224"]
    B, S, n_neg = 8, 11, 5
    cfg = bench.Cfg()
    d = bench.make_inputs(B, C, H, W, K, S, n_neg, 77, dev)
    desc = capi.make_desc(B, C, K, H, W, S, n_neg, cfg, (.18, .12, .46), capi.PREC_F16X3)
    lib = capi.load()
    by = ctypes.byref
    assert capi.corr_fwd_launches(desc, d["feats"], d["feats_pos"], d["code"], d["code_pos"]) == 1
    nws = lib.stego_corr_workspace_bytes(by(desc))
    f32 = dict(dtype=torch.float32, device=dev)

    def outs():
        return [torch.empty(3, **f32), torch.empty(B, S ** 4, **f32), torch.empty(B, S ** 4, **f32), torch.empty(n_neg * B, S ** 4, **f32),
                torch.empty(n_neg * B, S ** 4, **f32), torch.empty(7 * B, S ** 4, **f32), torch.empty(7, **f32),
                torch.empty(lib.stego_corr_saved_ctx_bytes(by(desc)), dtype=torch.uint8, device=dev)]
    maps = [capi._map(d[k]) for k in ("feats", "feats_pos", "code", "code_pos")]
    stream = torch.cuda.current_stream().cuda_stream

    def call(fn, ws, o):
        rc = fn(by(desc), *[by(m) for m in maps], d["coords1"].data_ptr(), d["coords2"].data_ptr(), d["perms"].data_ptr(),
                *[t.data_ptr() for t in o], ws.data_ptr(), ws.numel(), stream)
        assert rc == 0, rc
        torch.cuda.synchronize()
    dirty = torch.full((nws,), 255, dtype=torch.uint8, device=dev)
    o1 = outs()
    call(lib.stego_corr_fwd, dirty, o1)
    clean = torch.full((nws,), 255, dtype=torch.uint8, device=dev)
    assert lib.stego_corr_workspace_prepare(by(desc), clean.data_ptr(), clean.numel(), stream) == 0
    for rep in range(3):                                   # the same prepared workspace, call after call
        o2 = outs()
        call(lib.stego_corr_fwd_prepared, clean, o2)
        for a, b in zip(o1[:7], o2[:7]):
            assert torch.equal(a, b)


@pytest.mark.parametrize("shape", [
    dict(B=1, C=384, H=4, W=4, K=4, S=1, n_neg=1),        # one sample point, one image (perm = [0])
    dict(B=2, C=384, H=3, W=5, K=6, S=2, n_neg=0),        # no negatives: no rendezvous, two pair-sets only
    dict(B=5, C=768, H=7, W=6, K=66, S=7, n_neg=2),       # ViT-B width, B not a multiple of 8 (uneven phase-1 shares)
    dict(B=9, C=384, H=12, W=12, K=70, S=11, n_neg=3),    # 45 tiles on 45 workgroups, anchors of 9 images over 8 XCDs
    dict(B=3, C=384, H=2, W=2, K=72, S=3, n_neg=1),       # 2x2 map: most taps clamp; K at the three-launch limit
    dict(B=36, C=384, H=8, W=8, K=10, S=4, n_neg=5),      # 252 tiles: the largest batch that is still one workgroup per CU
])
@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_fused_path_edge_shapes(shape, precision):
    """The single-launch forward (channels-last ViT-width maps) at the corners of its domain, forward + backward vs the oracle."""
    d = O.synth_inputs(seed=25, **shape)
    cfg = O.CorrCfg(feature_samples=shape["S"], neg_samples=shape["n_neg"])
    inputs = {k: d[k] for k in ("feats", "feats_pos", "code", "code_pos", "coords1", "coords2")}
    cl = {k: _channels_last(_dev(d[k])) for k in ("feats", "feats_pos", "code", "code_pos")}
    desc = capi.make_desc(shape["B"], shape["C"], shape["K"], shape["H"], shape["W"], shape["S"], shape["n_neg"], cfg, (.18, .12, .46))
    assert capi.corr_fwd_launches(desc, cl["feats"], cl["feats_pos"], cl["code"], cl["code_pos"]) == 1
    r = _run(inputs, d["perms"], cfg, layout="cl", precision=precision)
    ref = O.corr_loss_forward(**inputs, perms=d["perms"], cfg=cfg)
    la = 5e-4
    assert_close(r["out"][1], ref.pos_intra_cd, atol_frac=la, what="intra_cd")
    assert_close(r["out"][3], ref.pos_inter_cd, atol_frac=la, what="inter_cd")
    assert_close(r["out"][4], ref.neg_inter_loss, atol_frac=la, what="neg_loss")
    scale = max(abs(float(ref.pos_intra_loss)), abs(float(ref.pos_inter_loss)), 1e-3)
    assert abs(float(r["out"][0]) - float(ref.pos_intra_loss)) < 1e-3 * scale
    assert abs(float(r["out"][2]) - float(ref.pos_inter_loss)) < 1e-3 * scale
    numel = shape["B"] * shape["S"] ** 4
    g_nl = np.full(ref.neg_inter_loss.shape, 0.63 / (shape["n_neg"] * numel)) if shape["n_neg"] else None
    dc, dcp = O.corr_loss_backward(**inputs, perms=d["perms"], cfg=cfg, g_intra=0.67, g_inter=0.25, g_neg_loss=g_nl)
    assert_close(r["d_code"], dc, rtol=1e-3, atol_frac=1e-3, what="d_code")
    assert_close(r["d_code_pos"], dcp, rtol=1e-3, atol_frac=1e-3, what="d_code_pos")


@pytest.mark.parametrize("B,K,precision", [(40, 8, "f16x3"), (48, 70, "f16x3"), (64, 70, "f32"), (64, 100, "f16x3"), (100, 70, "f16x3")])
def test_more_tiles_than_compute_units_run_in_rounds_of_whole_pair_sets(B, K, precision):
    """(2 + 5) * B tiles > 256 compute units (the reference's batch_size is free, train_config.yml:11; B = 64 is an obvious way to use a
    288 GB GPU): the single-launch forward takes them in ROUNDS of whole pair-sets - 6 + 1 pair-sets at B = 40, 5 + 2 at 48, 4 + 3 at 64,
    2 + 2 + 2 + 1 at 100 - the later windows of the grid starting as compute units become free.  Forward and backward against the fp64
    oracle; K = 100 is a code dimension that only exists on this path."""
    S, n_neg, H, W = 5, 5, 9, 7
    d = O.synth_inputs(B, 384, H, W, K, S, n_neg, seed=260 + B, dino_like=True)
    cfg = O.CorrCfg(feature_samples=S, neg_samples=n_neg)
    inputs = {k: d[k] for k in ("feats", "feats_pos", "code", "code_pos", "coords1", "coords2")}
    cl = {k: _channels_last(_dev(d[k])) for k in ("feats", "feats_pos", "code", "code_pos")}
    desc = capi.make_desc(B, 384, K, H, W, S, n_neg, cfg, (.18, .12, .46))
    assert capi.corr_fwd_launches(desc, cl["feats"], cl["feats_pos"], cl["code"], cl["code_pos"]) == 1
    r = _run(inputs, d["perms"], cfg, layout="cl", precision=precision)
    ref = O.corr_loss_forward(**inputs, perms=d["perms"], cfg=cfg)
    assert_close(r["out"][1], ref.pos_intra_cd, atol_frac=5e-4, what="intra_cd")
    assert_close(r["out"][3], ref.pos_inter_cd, atol_frac=5e-4, what="inter_cd")
    assert_close(r["out"][4], ref.neg_inter_loss, atol_frac=5e-4, what="neg_loss")
    assert_close(r["out"][5], ref.neg_inter_cd, atol_frac=5e-4, what="neg_cd")
    scale = float(np.mean(np.abs(ref.neg_inter_loss)))
    assert abs(float(r["out"][0]) - float(ref.pos_intra_loss)) <= 1e-3 * scale + 1e-3 * abs(float(ref.pos_intra_loss))
    assert abs(float(r["out"][2]) - float(ref.pos_inter_loss)) <= 1e-3 * scale + 1e-3 * abs(float(ref.pos_inter_loss))
    numel = B * S ** 4
    g_nl = np.full((n_neg * B,) + (S,) * 4, 0.63 / (n_neg * numel))
    dc, dcp = O.corr_loss_backward(**inputs, perms=d["perms"], cfg=cfg, g_intra=0.67, g_inter=0.25, g_neg_loss=g_nl)
    assert_close(r["d_code"], dc, rtol=2e-3, atol_frac=1e-3, what="d_code")
    assert_close(r["d_code_pos"], dcp, rtol=2e-3, atol_frac=1e-3, what="d_code_pos")


def test_batch_64_at_full_size_single_launch_against_the_three_launch_path():
    """B = 64 at BASELINE config 2's map size: 448 tiles in two rounds (4 + 3 pair-sets), every output against the three-launch forward
    of the same library, on rotating inputs."""
    import bench
    dev = torch.device("cuda:0")
    B, C, H, W, K, S, n_neg = 64, 384, 28, 28, 70, 11, 5
    cfg = bench.Cfg()
    desc = capi.make_desc(B, C, K, H, W, S, n_neg, cfg, (.18, .12, .46), capi.PREC_F16X3)
    try:
        for seed in (6401, 6402):
            d = bench.make_inputs(B, C, H, W, K, S, n_neg, seed, dev)
            capi.debug_set("STEGO_FWD_VARIANT", 1)
            ref = _cfg2_capi_run(desc, d)
            capi.debug_set("STEGO_FWD_VARIANT", 0)
            assert capi.corr_fwd_launches(desc, d["feats"], d["feats_pos"], d["code"], d["code_pos"]) == 1
            for rep in range(2):
                got = _cfg2_capi_run(desc, d)
                for g, r in zip(got, ref):
                    assert not torch.isnan(g).any()
                    assert float((g - r).abs().max()) < 3e-6
    finally:
        capi.debug_set("STEGO_FWD_VARIANT", 0)


def test_split_fp16_backward_error_stays_in_the_fp32_class():
    """F16X3 also runs the backward's two code GEMMs as split-fp16 products (corr_bwd_tile_h_kernel); F32 keeps
    v_mfma_f32_16x16x4_f32.  Against the fp64 oracle the split gradients must be as good as the fp32 ones: same error class
    (<= 2x + rounding floor), on smooth DINO-like inputs and on codes with a 2^12 dynamic range inside a point (tiny channels
    fall into fp16-subnormal lo halves of the transposed operand images)."""
    B, C, H, W, K, S, n_neg = 4, 384, 12, 12, 70, 11, 2
    d = O.synth_inputs(B, C, H, W, K, S, n_neg, seed=505, dino_like=True)
    rng = np.random.default_rng(505)
    code = d["code"].copy()
    code_pos = d["code_pos"].copy()
    spread = np.exp2(rng.integers(-12, 1, size=(1, K, 1, 1))).astype(np.float32)
    code[1] *= spread[0]
    code_pos[2] *= spread[0]
    code[3] *= np.float32(1e-5)
    inputs = dict(feats=d["feats"], feats_pos=d["feats_pos"], code=code, code_pos=code_pos, coords1=d["coords1"], coords2=d["coords2"])
    cfg = O.CorrCfg(neg_samples=n_neg)
    numel = B * S ** 4
    g_nl = np.full((n_neg * B,) + (S,) * 4, 0.63 / (n_neg * numel))
    dc, dcp = O.corr_loss_backward(**inputs, perms=d["perms"], cfg=cfg, g_intra=0.67, g_inter=0.25, g_neg_loss=g_nl)
    err = {}
    for precision in ("f32", "f16x3"):
        r = _run(inputs, d["perms"], cfg, layout="cl", precision=precision)
        assert_close(r["d_code"], dc, rtol=1e-3, atol_frac=1e-3, what="d_code %s" % precision)
        assert_close(r["d_code_pos"], dcp, rtol=1e-3, atol_frac=1e-3, what="d_code_pos %s" % precision)
        err[precision] = max(np.abs(r["d_code"] - dc).max() / np.abs(dc).max(), np.abs(r["d_code_pos"] - dcp).max() / np.abs(dcp).max())
    assert err["f16x3"] <= 2.0 * err["f32"] + 1e-6, err
    assert err["f16x3"] < 2e-5, err


def _philox4x32_10(c, k):
    """Philox-4x32-10 (Salmon et al., SC'11) in plain Python: the generator of stego_fast_draws."""
    c = list(c)
    k0, k1 = k
    for _ in range(10):
        p0 = 0xD2511F53 * c[0]
        p1 = 0xCD9E8D57 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k0) & 0xffffffff, p1 & 0xffffffff, ((p0 >> 32) ^ c[3] ^ k1) & 0xffffffff, p0 & 0xffffffff]
        k0 = (k0 + 0x9E3779B9) & 0xffffffff
        k1 = (k1 + 0xBB67AE85) & 0xffffffff
    return c


def test_fast_draws_kernel_is_philox_with_the_reference_distributions():
    """cfg.fast_draws (opt-in): coords = 2 u - 1 with u on torch.rand's 2^-24 lattice from Philox-4x32-10 (known-answer vector
    of Random123 + the kernel's counter layout checked against the Python restatement), perms = uniformly random permutations
    with super_perm's fix-up (modules.py:307-311); deterministic in the seed; the loss accepts them like the torch draws."""
    assert _philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]          # Random123 kat_vectors
    assert _philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    dev = torch.device(DEV)
    B, S, n_neg = 32, 11, 5
    seed_val = 0x1234567890abcdef
    seed = torch.tensor([seed_val - (1 << 64) if seed_val >= (1 << 63) else seed_val], dtype=torch.int64, device=dev)
    c1, c2, perms = capi.fast_draws(seed, [B, S, S, 2], n_neg, B)
    c1b, c2b, permsb = capi.fast_draws(seed, [B, S, S, 2], n_neg, B)
    assert torch.equal(c1, c1b) and torch.equal(c2, c2b) and torch.equal(perms, permsb)
    k = (seed_val & 0xffffffff, seed_val >> 32)
    flat1, flat2 = c1.flatten().cpu().numpy(), c2.flatten().cpu().numpy()
    for i4 in (0, 1, 7, 1000, flat1.size // 4 - 1):
        a = _philox4x32_10([i4, 0, 0, 0], k)
        b = _philox4x32_10([i4, 0, 0, 1], k)
        for e in range(4):
            assert flat1[4 * i4 + e] == np.float32(np.float32((a[e] >> 8) * 2.0 ** -24) * 2 - 1)
            assert flat2[4 * i4 + e] == np.float32(np.float32((b[e] >> 8) * 2.0 ** -24) * 2 - 1)
    assert float(c1.min()) >= -1 and float(c1.max()) < 1 and abs(float(c1.mean())) < 0.02 and abs(float(c1.var()) - 1 / 3) < 0.02
    # permutations: rank of 64-bit keys, then perm[perm == arange] += 1; perm % B
    pn = perms.cpu().numpy()
    for n in range(n_neg):
        keys = []
        for i in range(B):
            c = _philox4x32_10([i, 0, 1 + n, 2], k)
            keys.append((c[0] << 32) | c[1])
        order = np.argsort(np.array(keys, dtype=np.uint64), kind="stable")          # order[rank] = element
        fixed = np.where(order == np.arange(B), order + 1, order) % B
        assert np.array_equal(pn[n], fixed)
        assert not (pn[n] == np.arange(B)).any()
    other = capi.fast_draws(seed + 1, [B, S, S, 2], n_neg, B)
    assert not torch.equal(other[0], c1) and not torch.equal(other[2], perms)
    # the module path: forward() with cfg.fast_draws draws these and runs the fused loss
    import bench
    cfg = bench.Cfg()
    cfg.fast_draws = True
    C, H, W, K = bench.WORKLOADS["vits8_224"]
    d = bench.make_inputs(8, C, H, W, K, S, n_neg, 3, dev)
    code = d["code"].detach().clone().requires_grad_(True)
    torch.manual_seed(11)
    out = M.ContrastiveCorrelationLoss(cfg)(d["feats"], d["feats_pos"], None, None, code, d["code_pos"])
    torch.manual_seed(11)
    out2 = M.ContrastiveCorrelationLoss(cfg)(d["feats"], d["feats_pos"], None, None, code, d["code_pos"])
    assert torch.equal(out[4], out2[4]) and torch.isfinite(out[4]).all()
    (out[0] + out[2] + out[4].mean()).backward()
    assert torch.isfinite(code.grad).all() and float(code.grad.abs().max()) > 0


@pytest.mark.parametrize("layout", ["cl", "nchw"])
def test_total_api_on_device_equals_forward_composition(layout):
    """loss_means[2] (the mean over all negative losses, from the forward launch itself: fused path and three-launch path) and the
    mean-upstream backward (g_neg_loss_stride = -1) against forward() + .mean() + dense/broadcast upstreams, same draws."""
    import bench
    dev = torch.device(DEV)
    cfg = bench.Cfg()
    C, H, W, K = bench.WORKLOADS["vits8_224"]
    d = bench.make_inputs(8, C, H, W, K, 11, 5, 9, dev, layout)
    loss_fn = M.ContrastiveCorrelationLoss(cfg)
    code = d["code"].detach().clone().requires_grad_(True)
    code_pos = d["code_pos"].detach().clone().requires_grad_(True)
    torch.manual_seed(21)
    out = loss_fn(d["feats"], d["feats_pos"], None, None, code, code_pos)
    ref = 0.67 * out[0] + 0.25 * out[2] + 0.63 * out[4].mean()
    ref.backward()
    g1, g2 = code.grad.clone(), code_pos.grad.clone()
    code.grad = None
    code_pos.grad = None
    torch.manual_seed(21)
    total, means, icd, ecd, ncd = loss_fn.total(d["feats"], d["feats_pos"], None, None, code, code_pos, (0.67, 0.25, 0.63))
    assert abs(float(means[2]) - float(out[4].mean())) < 1e-6 * max(abs(float(out[4].mean())), 1e-3) + 1e-9
    assert abs(float(total) - float(ref)) < 1e-5 * abs(float(ref)) + 1e-9
    total.backward()
    for got, want in ((code.grad, g1), (code_pos.grad, g2)):
        assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())


def test_randomised_shapes_against_the_oracle():
    """tools/fuzz_fused.py as a regression test: 16 random shapes / cfg switches / layouts / precisions inside and outside the
    fused path's domain, forward + backward against the fp64 oracle."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_fused.py"), "16", "7"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1] == "failures: 0", out.stdout[-3000:]


def _cfg2_capi_run(desc, d, debug=0, before=None):
    """One forward of BASELINE config 2 through the C ABI with the library's debug knob set; every output + the saved w / means."""
    capi.debug_set("STEGO_DEBUG", debug)
    try:
        if before is not None:
            before()
        o = capi.corr_fwd(desc, d["feats"], d["feats_pos"], d["code"], d["code_pos"], d["coords1"], d["coords2"], d["perms"], True)
        torch.cuda.synchronize()
    finally:
        capi.debug_set("STEGO_DEBUG", 0)
    return [t.clone() for t in (o[0],) + tuple(o[1:5]) + (o[5][0], o[5][1])]


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_fused_give_up_and_repair_paths_are_bitwise_the_normal_launch(precision):
    """The two fallback paths of the single-launch forward - what a shared or partitioned device takes - forced by the library's debug
    bits at BASELINE config 2's full size: bit 64 = no workgroup samples in phase 1, every tile gives up waiting for its anchor (after
    1 us) and samples it itself; bit 32 = no tile applies old_mean in its rendezvous, the last workgroup of the launch repairs all 160
    negative tiles.  Both must give the bytes of the normal launch (same arithmetic, other schedule), and those match the fp64 oracle."""
    import bench
    dev = torch.device("cuda:0")
    B, C, H, W, K, S, n_neg = 32, 384, 28, 28, 70, 11, 5
    cfg = bench.Cfg()
    prec = capi.PREC_F16X3 if precision == "f16x3" else capi.PREC_F32
    desc = capi.make_desc(B, C, K, H, W, S, n_neg, cfg, (.18, .12, .46), prec)
    for seed in (4101, 4102):                      # rotating inputs: stale bytes of the previous launch are not the right bytes
        d = bench.make_inputs(B, C, H, W, K, S, n_neg, seed, dev)
        assert capi.corr_fwd_launches(desc, d["feats"], d["feats_pos"], d["code"], d["code_pos"]) == 1
        normal = _cfg2_capi_run(desc, d)
        for bits in (64, 32, 96):
            alt = _cfg2_capi_run(desc, d, debug=bits)
            for x, y in zip(normal, alt):
                assert not torch.isnan(y).any()
                assert torch.equal(x, y), (bits, float((x - y).abs().max()))
    ref = O.corr_loss_forward(*[d[k].cpu().numpy() for k in ("feats", "feats_pos", "code", "code_pos", "coords1", "coords2", "perms")],
                              O.CorrCfg(neg_samples=n_neg))
    alt = _cfg2_capi_run(desc, d, debug=96)
    assert_close(alt[1].cpu().numpy().reshape(-1), ref.pos_intra_cd.reshape(-1), what="intra cd (fallback paths)")
    assert_close(alt[3].cpu().numpy().reshape(-1), ref.neg_inter_loss.reshape(-1), atol_frac=5e-4, what="neg loss (fallback paths)")
    assert abs(float(alt[0][2]) - float(ref.neg_inter_loss.mean())) < 1e-3 * abs(float(ref.neg_inter_loss.mean()))


@pytest.mark.parametrize("shared", [False, True])
def test_fused_forward_beside_a_foreign_kernel_that_holds_compute_units(shared):
    """A stand-in collective (stego_debug_occupy: 40 workgroups with 64 KB of LDS each spinning 120 us on another stream) owns compute
    units when the forward is launched.  With the per-call flag STEGO_FLAG_SHARED_DEVICE the forward needs only its tiles' workgroups;
    without it the workgroups that cannot be placed come late - the tiles whose anchors they own give up after their bounded spin and
    sample themselves.  Either way: the bytes of the undisturbed launch, no NaN, no hang."""
    import bench
    dev = torch.device("cuda:0")
    B, C, H, W, K, S, n_neg = 32, 384, 28, 28, 70, 11, 5
    cfg = bench.Cfg()
    desc = capi.make_desc(B, C, K, H, W, S, n_neg, cfg, (.18, .12, .46), capi.PREC_F16X3, shared_device=shared)
    side = torch.cuda.Stream()
    for seed in (4201, 4202, 4203):
        d = bench.make_inputs(B, C, H, W, K, S, n_neg, seed, dev)
        quiet = _cfg2_capi_run(desc, d)

        def hold():
            capi.occupy(40, 64 * 1024, 120, stream=side)
            torch.cuda._sleep(30000)               # let the stand-in get onto its compute units first
        busy = _cfg2_capi_run(desc, d, before=hold)
        for x, y in zip(quiet, busy):
            assert not torch.isnan(y).any()
            assert torch.equal(x, y), float((x - y).abs().max())


def test_fused_forward_shared_device_mode_matches():
    """STEGO_FLAG_SHARED_DEVICE (per call; cfg.shared_device / capi.set_shared_device): the fused forward launches one
    workgroup per tile instead of one per compute unit, a third of them take a second phase-1 pass.  Same bytes out."""
    c = GoldenCase("cfg1_B4_vits8_dinolike")
    half = _run(c.inputs, c.perms, c.cfg, layout="cl", grad=False, precision="f16x3")["out"]       # B = 4: the column-half launch (round 6)
    capi.debug_set("STEGO_DEBUG", 16384)       # the full-tile launch of the same library: the launch shape the shared mode varies
    try:
        base = _run(c.inputs, c.perms, c.cfg, layout="cl", grad=False, precision="f16x3")["out"]
    finally:
        capi.debug_set("STEGO_DEBUG", 0)
    for x, y in zip(base, half):               # the two launch kinds add a row's two halves in different orders: last bits, not more
        np.testing.assert_allclose(x, y, rtol=2e-5, atol=3e-5 * float(np.abs(y).mean()))
    capi.set_shared_device(True)
    try:
        alt = _run(c.inputs, c.perms, c.cfg, layout="cl", grad=False, precision="f16x3")["out"]
        B, C, H, W, K, S, n_neg = 32, 384, 28, 28, 70, 11, 5            # the full-size launch: 224 tiles, 18.3 anchor rows per workgroup
        d = O.synth_inputs(B, C, H, W, K, S, n_neg, seed=77, dino_like=True)
        inputs = {k: d[k] for k in ("feats", "feats_pos", "code", "code_pos", "coords1", "coords2")}
        cfg = O.CorrCfg(neg_samples=n_neg)
        big_alt = _run(inputs, d["perms"], cfg, layout="cl", grad=False, precision="f16x3")["out"]
    finally:
        capi.set_shared_device(False)
    big = _run(inputs, d["perms"], cfg, layout="cl", grad=False, precision="f16x3")["out"]
    for x, y in list(zip(base, alt)) + list(zip(big, big_alt)):
        np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("shared", [False, True])
def test_fused_forward_at_the_largest_batch_of_a_launch(shared):
    """B = 36 with 5 negatives: 252 tiles on 256 CUs, five anchors on XCDs 0-3, i.e. 20 anchor rows per workgroup when every CU has one
    (19 and 23 per tile workgroup in the one-per-tile launch, which then needs a second phase-1 pass): the gather waves that take the
    overflow rows of phase 1 are all busy.  Fused forward against the three-launch path of the same library."""
    import bench
    dev = torch.device("cuda:0")
    B, C, H, W, K, S, n_neg = 36, 384, 14, 14, 70, 11, 5
    cfg = bench.Cfg()
    d = bench.make_inputs(B, C, H, W, K, S, n_neg, 5151, dev)
    desc = capi.make_desc(B, C, K, H, W, S, n_neg, cfg, (.18, .12, .46), capi.PREC_F16X3, shared_device=shared)

    def run():
        o = capi.corr_fwd(desc, d["feats"], d["feats_pos"], d["code"], d["code_pos"], d["coords1"], d["coords2"], d["perms"], True)
        torch.cuda.synchronize()
        return [t.clone() for t in (o[0],) + tuple(o[1:5]) + (o[5][0],)]

    try:
        capi.debug_set("STEGO_FWD_VARIANT", 1)
        ref = run()
        capi.debug_set("STEGO_FWD_VARIANT", 0)
        assert capi.corr_fwd_launches(desc, d["feats"], d["feats_pos"], d["code"], d["code_pos"]) == 1
        for rep in range(5):
            for g, r in zip(run(), ref):
                assert not torch.isnan(g).any()
                assert float((g - r).abs().max()) < 2e-6
    finally:
        capi.debug_set("STEGO_FWD_VARIANT", 0)



def test_nccl_world_size_one_dress_rehearsal_of_the_data_parallel_step():
    """RCCL has never run this code with more than one rank where the builder can see it (1-GPU boxes): at least the whole N > 1 path -
    process group on backend "nccl", flat gradient bucket, asynchronous ReduceOp.AVG launched by manual_backward, wait_gradients as a
    stream dependency, the all-reduce captured in bench.py's step graphs - runs here on a group of ONE, and must give exactly the
    numbers of the run without a collective (the mean over one rank is the identity)."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_PORT="29547", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "nccl_world1_worker.py")], capture_output=True, text=True,
                         timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    last_json = lambda text: json.loads([l for l in text.strip().splitlines() if l.startswith("{")][-1])     # (RCCL prints its banner to stdout)
    rec = last_json(out.stdout)
    # (not bitwise: the torch ops of the probes' backward are not run-to-run deterministic; the collective itself is the identity)
    assert np.allclose(rec["loss_collective"], rec["loss_plain"], rtol=1e-6, atol=0) and rec["head_equal"], rec
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "nccl_world1.json"), "w") as f:
        json.dump(rec, f)
    # the bench's own N > 1 protocol with the all-reduce captured in the step graphs, on a group of one
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "40", "--warmup", "10", "--force-collective",
           "--launch", "graph", "--no-cpu-baseline", "--no-alt"]
    env2 = {k: v for k, v in env.items() if k not in ("RANK", "WORLD_SIZE")}
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env2, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    b = last_json(out.stdout)
    assert b["config"]["collective"] and b["config"]["launch"].startswith("hipgraph") and "all-reduce captured" in b["config"]["launch"] and b["value"] > 0, b["config"]
    with open(os.path.join(root, "gpurun_out", "nccl_world1_bench.json"), "w") as f:
        json.dump(b, f)


def test_loss_kernels_beside_an_rccl_kernel_give_the_same_bits():
    """STEGO_FLAG_SHARED_DEVICE next to a real RCCL device kernel (tests/rccl_beside_worker.py: a one-rank group's ReduceOp.AVG, the
    collective that launches a kernel at world size 1), eagerly on two streams and as two branches of one captured graph: forward
    outputs and both gradients bit for bit those of the quiet device."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import socket
    with socket.socket() as sk:                       # a free rendezvous port (a fixed one may still be held by an earlier test's group)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(root, "tests", "rccl_beside_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    if out.returncode != 0:       # the process group did not come up (rendezvous / RCCL init): once more - a MISMATCH is a zero exit code
        first = out.stderr[-1500:]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(env, MASTER_PORT=str(port + 1)))
        assert out.returncode == 0, (first, out.stdout[-1500:], out.stderr[-3000:])
    rec = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert rec["eager_equal"] and rec["graph_equal"] and rec["bucket_is_ones"] and rec["eager_rounds"] == 12 and rec["graph_rounds"] == 12, rec
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "rccl_beside.json"), "w") as f:
        json.dump(rec, f)


@pytest.mark.parametrize("B,S,n_neg", [(4, 11, 5), (16, 11, 5), (32, 11, 5), (36, 11, 5), (32, 7, 3), (300, 5, 2)])
def test_one_launch_draws_are_the_torch_generator_s_rand_and_randperm(B, S, n_neg):
    """stego_ref_draws against the seven torch calls of the reference (modules.py:366-367 torch.rand x 2, :383 super_perm = torch.randperm
    x neg_samples) from the same generator state, over hundreds of seeds per size: every float and every index equal, the generator left
    at the same offset.  13 key bits at B = 32 make ~6 % of the permutations hit the duplicate-key reshuffle; B = 300 (18 bits of 300
    keys) exercises the sort."""
    dev = torch.device("cuda:0")
    shape = [B, S, S, 2]
    v = M.ref_draw_variant(shape, n_neg, B, dev)
    assert v >= 0, "no variant of stego_ref_draws reproduces this torch build: the product would fall back to the torch calls"
    gen = M._device_generator(dev)
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    dup = 0
    n_seeds = 2000 if B <= 36 else 60
    for seed in range(n_seeds):
        torch.manual_seed(1000 + seed)
        if seed % 3 == 1:
            torch.rand(5, device=dev)                       # (an offset that is not 0)
        st = gen.get_state()
        ref = M._torch_draws(shape, n_neg, B, dev)
        off = gen.get_offset()
        gen.set_state(st)
        c1, c2, perms = capi.ref_draws(gen, shape, n_neg, B, v, dev)
        assert gen.get_offset() == off
        bad += (c1 != ref[0] * 2 - 1).sum() + (c2 != ref[1] * 2 - 1).sum() + (perms != M._unfix(torch.stack(ref[2:]))).sum()
    assert int(bad) == 0


@pytest.mark.parametrize("B", [4, 32])
def test_one_launch_draws_captured_in_a_hip_graph_follow_the_generator_replay_after_replay(B):
    """stego_ref_draws_indirect: the launch captured ONCE reads the generator's state where CUDAGraph.replay refreshes it
    (at::PhiloxCudaState in its captured form, handed over by the in-tree torch extension), so replay k draws what the seven torch
    calls would draw k-th from the generator's state at the first replay - whatever was seeded or consumed in between - and every
    replay advances the generator by the seven calls' amount."""
    dev = torch.device("cuda:0")
    S, n_neg = 11, 5
    shape = [B, S, S, 2]
    assert capi.torchglue() is not None, "stego_amd/lib/_stego_torchglue.so not built (__graft_entry__.build())"
    v = M.ref_draw_variant(shape, n_neg, B, dev)
    assert v >= 0
    gen = M._device_generator(dev)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        c1, c2, perms = capi.ref_draws(gen, shape, n_neg, B, v, dev)
        c1b, c2b, permsb = capi.ref_draws(gen, shape, n_neg, B, v, dev)          # a second forward inside the same graph
    for seed in range(40):
        torch.manual_seed(500 + seed)
        if seed % 2:
            torch.rand(7, device=dev)
        st = gen.get_state()
        ref = [M._torch_draws(shape, n_neg, B, dev) for _ in range(4)]
        off = gen.get_offset()
        gen.set_state(st)
        for k in range(2):
            g.replay()
            for got, want in (((c1, c2, perms), ref[2 * k]), ((c1b, c2b, permsb), ref[2 * k + 1])):
                assert torch.equal(got[0], want[0] * 2 - 1) and torch.equal(got[1], want[1] * 2 - 1)
                assert torch.equal(got[2], M._unfix(torch.stack(want[2:])))
        assert gen.get_offset() == off


def test_captured_product_step_draws_like_the_eager_one():
    """ContrastiveCorrelationLoss.forward captured in a HIP graph (after one eager call, which runs the draws' self-check) = the eager
    forward from the same generator state, replay after replay; the capture holds ONE draw kernel instead of the ~30 of the torch calls."""
    import bench
    dev = torch.device("cuda:0")
    B, C, H, W, K, S, n_neg = 8, 384, 28, 28, 70, 11, 5
    d = bench.make_inputs(B, C, H, W, K, S, n_neg, 98, dev)
    cfg = bench.Cfg()
    loss_fn = M.ContrastiveCorrelationLoss(cfg)
    args = (d["feats"], d["feats_pos"], None, None, d["code"], d["code_pos"])
    with torch.no_grad():
        loss_fn(*args)                                                            # eager warm-up: workspaces, self-check
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = loss_fn(*args)
        assert M._usable_ref_draw_variant([B, S, S, 2], n_neg, B, dev) >= 0
        torch.manual_seed(11)
        want = [[t.clone() for t in loss_fn(*args)] for _ in range(3)]
        nxt = torch.rand(3, device=dev)
        torch.manual_seed(11)
        for k in range(3):
            g.replay()
            for a, b in zip(out, want[k]):
                assert torch.equal(a, b)
        assert torch.equal(torch.rand(3, device=dev), nxt)


def test_forward_with_one_launch_draws_equals_forward_with_the_torch_calls():
    """The product path: ContrastiveCorrelationLoss.forward seeded alike with cfg.one_launch_draws on (default) and off."""
    import bench
    dev = torch.device("cuda:0")
    B, C, H, W, K, S, n_neg = 8, 384, 28, 28, 70, 11, 5
    d = bench.make_inputs(B, C, H, W, K, S, n_neg, 99, dev)
    outs = []
    for one in (True, False):
        cfg = bench.Cfg()
        cfg.one_launch_draws = one
        torch.manual_seed(7)
        o = M.ContrastiveCorrelationLoss(cfg)(d["feats"], d["feats_pos"], None, None, d["code"], d["code_pos"])
        outs.append([t.detach().clone() for t in o] + [torch.rand(3, device=dev)])          # + the generator's next numbers
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_cpp_autograd_function_is_the_python_one_bit_for_bit():
    """csrc/torch_glue_ext.cpp::CorrLoss (the default on a HIP device) against modules._CorrLossFunction / _CorrLossMeansFunction
    (cfg.native_autograd = False): the same C ABI calls, so every output and every gradient is bitwise equal - with the training
    upstream (three scalars), with dense upstreams on every output, with an expanded-scalar upstream on the negative loss map, and
    through means() / total(); a second differentiation is refused."""
    import bench
    dev = torch.device("cuda:0")
    B, C, H, W, K, S, n_neg = 8, 384, 28, 28, 70, 11, 5
    d = bench.make_inputs(B, C, H, W, K, S, n_neg, 97, dev)
    assert capi.torchglue() is not None, "stego_amd/lib/_stego_torchglue.so not built (__graft_entry__.build())"
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    ups = None
    res = {}
    for native in (True, False):
        cfg = bench.Cfg()
        cfg.native_autograd = native
        loss_fn = M.ContrastiveCorrelationLoss(cfg)
        assert (M._native_autograd(cfg) is not None) == native
        rec = []
        for case in ("train", "dense", "expanded", "total"):
            c = d["code"].detach().clone().requires_grad_(True)
            cp = d["code_pos"].detach().clone().requires_grad_(True)
            if case == "total":
                torch.manual_seed(3)                                   # (total() makes its own draws)
                out = loss_fn.total(d["feats"], d["feats_pos"], None, None, c, cp, (0.67, 0.25, 0.63))
                out[0].backward()
                rec += [t.detach().clone() for t in out] + [c.grad.clone(), cp.grad.clone()]
                continue
            o = loss_fn.forward_explicit(d["feats"], d["feats_pos"], c, cp, d["coords1"], d["coords2"], d["perms"])
            rec += [t.detach().clone() for t in o]
            if case == "train":
                (0.67 * o[0] + 0.25 * o[2] + 0.63 * o[4].mean()).backward()
            elif case == "dense":
                if ups is None:
                    ups = [torch.randn(t.shape, device=dev, generator=gen) for t in o]
                torch.autograd.backward(list(o), ups)
            else:
                (o[0] * 2 + o[4].sum() * 0.5 + o[4].mean()).backward()
            rec += [c.grad.clone(), cp.grad.clone()]
        res[native] = rec
    assert len(res[True]) == len(res[False])
    for a, b in zip(res[True], res[False]):
        assert a.shape == b.shape and torch.equal(a, b)
    # once differentiable
    cfg = bench.Cfg()
    c = d["code"].detach().clone().requires_grad_(True)
    o = M.ContrastiveCorrelationLoss(cfg).forward_explicit(d["feats"], d["feats_pos"], c, d["code_pos"], d["coords1"], d["coords2"], d["perms"])
    up = torch.ones((), device=dev, requires_grad=True)
    with pytest.raises(RuntimeError, match="differentiable once"):
        torch.autograd.grad(o[0], c, grad_outputs=up, create_graph=True)


def test_without_the_torch_glue_extension_the_product_is_the_same(monkeypatch):
    """A host where stego_amd/lib/_stego_torchglue.so is missing (or does not load against its torch): the Python autograd.Function
    runs, eager draws come from Generator.get_offset / set_offset, captured steps keep the torch draw calls - outputs, gradients and
    generator consumption equal the run with the extension, eagerly and replayed from a graph."""
    import bench
    dev = torch.device("cuda:0")
    B, C, H, W, K, S, n_neg = 8, 384, 28, 28, 70, 11, 5
    d = bench.make_inputs(B, C, H, W, K, S, n_neg, 96, dev)
    assert capi.torchglue() is not None

    def run(captured):
        cfg = bench.Cfg()
        loss_fn = M.ContrastiveCorrelationLoss(cfg)
        c = d["code"].detach().clone().requires_grad_(True)
        cp = d["code_pos"].detach().clone().requires_grad_(True)

        def step():
            c.grad = None
            cp.grad = None
            o = loss_fn(d["feats"], d["feats_pos"], None, None, c, cp)
            (0.67 * o[0] + 0.25 * o[2] + 0.63 * o[4].mean()).backward()
            return o
        step()                                          # warm-up (workspaces, draw self-checks)
        torch.cuda.synchronize()
        if captured:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                o = step()
            torch.manual_seed(21)
            g.replay()
        else:
            torch.manual_seed(21)
            o = step()
        return [t.detach().clone() for t in o] + [c.grad.clone(), cp.grad.clone(), torch.rand(3, device=dev)]

    with_ext = [run(False), run(True)]
    monkeypatch.setattr(capi, "_torchglue_mod", False)
    assert capi.torchglue() is None and M._native_autograd(bench.Cfg()) is None
    without = [run(False), run(True)]
    for a_run, b_run in zip(with_ext, without):
        for a, b in zip(a_run, b_run):
            assert torch.equal(a, b)


def test_lazy_loss_sum_gives_the_plain_gradients_eagerly_and_inside_a_captured_graph():
    """The reference's weighted sum of the three loss scalars (train_segmentation.py:178-181) written on the outputs of forward():
    with cfg.lazy_loss_sums = True (opt-in since round 6) the coefficients reach the loss op's backward as its upstream gradients without
    a kernel; values and gradients equal the plain tensor expression's (the default), eagerly and replayed from a HIP graph."""
    import copy
    B, C, H, W, K, S, n_neg = 8, 384, 28, 28, 70, 11, 5
    d = O.synth_inputs(B, C, H, W, K, S, n_neg, seed=4)
    t = {k: _dev(v) for k, v in d.items()}
    f, fp = _channels_last(t["feats"]), _channels_last(t["feats_pos"])
    cfg = O.CorrCfg()
    cfg.corr_precision = "f16x3"
    plain_cfg = copy.copy(cfg)                 # the default: three plain 0-dim tensors
    assert not getattr(plain_cfg, "lazy_loss_sums", False)
    cfg.lazy_loss_sums = True

    def run(cfg_, c, cp):
        out = M.ContrastiveCorrelationLoss(cfg_).forward_explicit(f, fp, c, cp, t["coords1"], t["coords2"], t["perms"])
        loss = 0
        loss += (0.25 * out[2].mean() + 0.67 * out[0].mean() + 0.63 * out[4].mean()) * 1.5
        return out, loss

    res = {}
    for name, cfg_ in (("lazy", cfg), ("plain", plain_cfg)):
        c = _channels_last(t["code"]).detach().requires_grad_(True)
        cp = _channels_last(t["code_pos"]).detach().requires_grad_(True)
        out, loss = run(cfg_, c, cp)
        assert isinstance(loss, M._LazyLoss) == (name == "lazy")
        assert isinstance(loss, torch.Tensor) == (name == "plain")
        val = float(loss)                      # (evaluates a copy of the sum; the lazy object itself stays lazy for backward())
        out, loss = run(cfg_, c, cp)
        loss.backward()
        res[name] = (val, c.grad.clone(), cp.grad.clone())
    assert abs(res["lazy"][0] - res["plain"][0]) <= 1e-6 * abs(res["plain"][0])
    for a, b in zip(res["lazy"][1:], res["plain"][1:]):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6 * float(b.abs().max()))
    # captured: the device constants exist (made by the eager run above), so the lazy backward is capturable
    c = _channels_last(t["code"]).detach().requires_grad_(True)
    cp = _channels_last(t["code_pos"]).detach().requires_grad_(True)
    for _ in range(2):
        c.grad = cp.grad = None
        run(cfg, c, cp)[1].backward()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    c.grad = cp.grad = None
    with torch.cuda.graph(g, capture_error_mode="thread_local"):      # (the backward runs on the autograd engine's device thread)
        run(cfg, c, cp)[1].backward()
    g.replay()
    torch.cuda.synchronize()
    assert torch.allclose(c.grad, res["plain"][1], rtol=1e-5, atol=1e-6 * float(res["plain"][1].abs().max()))
    assert torch.allclose(cp.grad, res["plain"][2], rtol=1e-5, atol=1e-6 * float(res["plain"][2].abs().max()))


def test_event_counters_report_the_give_up_and_repair_paths():
    """stego_corr_event_counters (ABI 5): zero in normal operation; every tile that gives up on its anchor (debug 64 forces all of
    them) and every negative tile whose old_mean rendezvous does not happen (debug 32) is counted - so a host on a shared or
    partitioned device can tell why a launch was slow."""
    import bench
    dev = torch.device("cuda:0")
    C, H, W, K = bench.WORKLOADS["vits8_224"]
    B, S, n_neg = 32, 11, 5
    cfg = bench.Cfg()
    d = bench.make_inputs(B, C, H, W, K, S, n_neg, 77, dev)
    desc = capi.make_desc(B, C, K, H, W, S, n_neg, cfg, (.18, .12, .46), capi.PREC_F16X3)
    lib = capi.load()

    def run():
        capi.corr_fwd(desc, d["feats"], d["feats_pos"], d["code"], d["code_pos"], d["coords1"], d["coords2"], d["perms"], True)
        torch.cuda.synchronize()
        return capi.event_counters(desc, capi._prepared_ws(lib, desc, dev))

    try:
        # (cumulative since the workspace was prepared, and the workspace cache hands out the one an earlier test with this descriptor
        # used: deltas)
        base = run()
        assert run() == base, "normal launches add nothing"
        n_tiles = (2 + n_neg) * B
        capi.debug_set("STEGO_DEBUG", 64)
        a = run()
        assert (a[0] - base[0], a[1] - base[1]) == (n_tiles, 0), (base, a)
        capi.debug_set("STEGO_DEBUG", 32)
        b = run()
        assert (b[0] - a[0], b[1] - a[1]) == (0, n_neg * B), (a, b)
        capi.debug_set("STEGO_DEBUG", 0)
        assert run() == b
        # what the trainer reads every few hundred steps (round 5): the sum over every kept workspace of this process - the Python wrapper's
        # and the C++ autograd function's - and its one-time warning
        tot = capi.event_counters_total()
        assert tot[0] >= b[0] and tot[1] >= b[1], (tot, b)
        import warnings
        from stego_amd.train_segmentation import Trainer
        tr = Trainer.__new__(Trainer)
        tr.device, tr.rank, tr._events_warned = dev, 0, False
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            tr._check_loss_events(200)
            tr._check_loss_events(400)
        assert len([x for x in w if "sampled their anchor themselves" in str(x.message)]) == 1 and tr._events_warned
    finally:
        capi.debug_set("STEGO_DEBUG", 0)


@pytest.mark.parametrize("shape", [(8, 14, 14, 70, 11, 5), (3, 6, 9, 101, 7, 2), (33, 14, 14, 24, 7, 5)])
@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_vit_tiny_width_takes_the_single_launch_forward(shape, precision):
    """C = 192 (vit_tiny, src/dino/vision_transformer.py:259-263): six feature stages of 32 channels - one and a half of the 128-channel
    groups a half-wave samples at a time - on the single-launch kernel (round 4; the three-launch kernels before), forward and
    backward against the fp64 oracle, incl. a code dimension above 72 and more tiles than compute units (33 x 7 = 231 ... one round)."""
    B, H, W, K, S, n_neg = shape
    C = 192
    # (a seed whose cd has no element within fp32 rounding of the clamp bound: there the pass mask of ANY fp32 evaluation may differ from
    # the fp64 oracle's, one whole term of a gradient sum - see tests/test_bwd_fused.py)
    d = O.synth_inputs(B, C, H, W, K, S, n_neg, seed=500 + K + B)
    d["coords1"][B - 1, 0, 0] = [1.0, 1.0]                   # the last pixel of the last image: channels 192.. would be past the tensor
    d["coords2"][B - 1, 0, 0] = [1.0, 1.0]
    cfg = O.CorrCfg(feature_samples=S, neg_samples=n_neg)
    inputs = {k: d[k] for k in ("feats", "feats_pos", "code", "code_pos", "coords1", "coords2")}
    t = {k: _dev(v) for k, v in inputs.items()}
    desc = capi.make_desc(B, C, K, H, W, S, n_neg, cfg, (.18, .12, .46), capi.PREC_F16X3 if precision == "f16x3" else capi.PREC_F32)
    cl = [_channels_last(t[k]) for k in ("feats", "feats_pos", "code", "code_pos")]
    assert capi.corr_fwd_launches(desc, *cl) == 1
    r = _run(inputs, d["perms"], cfg, layout="cl", precision=precision)
    ref = O.corr_loss_forward(**inputs, perms=d["perms"], cfg=cfg)
    la = 5e-4
    assert_close(r["out"][1], ref.pos_intra_cd, atol_frac=la, what="intra_cd")
    assert_close(r["out"][3], ref.pos_inter_cd, atol_frac=la, what="inter_cd")
    assert_close(r["out"][4], ref.neg_inter_loss, atol_frac=la, what="neg_loss")
    assert_close(r["out"][5], ref.neg_inter_cd, atol_frac=la, what="neg_cd")
    scale = float(np.mean(np.abs(ref.neg_inter_loss)))
    assert abs(float(r["out"][0]) - float(ref.pos_intra_loss)) <= 1e-3 * scale + 1e-3 * abs(float(ref.pos_intra_loss))
    assert abs(float(r["out"][2]) - float(ref.pos_inter_loss)) <= 1e-3 * scale + 1e-3 * abs(float(ref.pos_inter_loss))
    g_nl = np.full(ref.neg_inter_loss.shape, 0.63 / (n_neg * B * S ** 4))
    dc, dcp = O.corr_loss_backward(**inputs, perms=d["perms"], cfg=cfg, g_intra=0.67, g_inter=0.25, g_neg_loss=g_nl)
    assert_close(r["d_code"], dc, rtol=1e-3, atol_frac=1e-3, what="d_code")
    assert_close(r["d_code_pos"], dcp, rtol=1e-3, atol_frac=1e-3, what="d_code_pos")


@pytest.mark.parametrize("K", [3, 69, 71])
@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_odd_code_dimensions_take_the_single_launch_forward(K, precision):
    """cfg.dim is free in the reference (train_config.yml:39): odd code dimensions - pixels that are only 4-byte aligned, a last
    channel pair that straddles the pixel's end - run on the single-launch kernel too (round 4), forward and backward against the
    fp64 oracle; the last image's last pixel is the case where a careless pair load would leave the tensor."""
    B, C, H, W, S, n_neg = 3, 384, 9, 10, 7, 2
    d = O.synth_inputs(B, C, H, W, K, S, n_neg, seed=50 + K)
    d["coords1"][B - 1, 0, 0] = [1.0, 1.0]                   # the last pixel of the last image, by both sides
    d["coords2"][B - 1, 0, 0] = [1.0, 1.0]
    cfg = O.CorrCfg(feature_samples=S, neg_samples=n_neg)
    inputs = {k: d[k] for k in ("feats", "feats_pos", "code", "code_pos", "coords1", "coords2")}
    t = {k: _dev(v) for k, v in inputs.items()}
    desc = capi.make_desc(B, C, K, H, W, S, n_neg, cfg, (.18, .12, .46), capi.PREC_F16X3 if precision == "f16x3" else capi.PREC_F32)
    cl = [_channels_last(t[k]) for k in ("feats", "feats_pos", "code", "code_pos")]
    assert capi.corr_fwd_launches(desc, *cl) == 1
    r = _run(inputs, d["perms"], cfg, layout="cl", precision=precision)
    ref = O.corr_loss_forward(**inputs, perms=d["perms"], cfg=cfg)
    la = 5e-4
    assert_close(r["out"][1], ref.pos_intra_cd, atol_frac=la, what="intra_cd")
    assert_close(r["out"][3], ref.pos_inter_cd, atol_frac=la, what="inter_cd")
    assert_close(r["out"][4], ref.neg_inter_loss, atol_frac=la, what="neg_loss")
    assert_close(r["out"][5], ref.neg_inter_cd, atol_frac=la, what="neg_cd")
    g_nl = np.full(ref.neg_inter_loss.shape, 0.63 / (n_neg * B * S ** 4))
    dc, dcp = O.corr_loss_backward(**inputs, perms=d["perms"], cfg=cfg, g_intra=0.67, g_inter=0.25, g_neg_loss=g_nl)
    assert_close(r["d_code"], dc, rtol=1e-3, atol_frac=1e-3, what="d_code")
    assert_close(r["d_code_pos"], dcp, rtol=1e-3, atol_frac=1e-3, what="d_code_pos")


@pytest.mark.parametrize("layout", ["cl", "nchw"])
def test_native_sample_with_index_matches_the_reference_expression(layout):
    """stego_sample / stego_sample_bwd (ABI 7) against the reference's own statement sample(t[perm], coords.repeat(..)) (modules.py:287-288,
    :384-385) on torch's grid_sample: forward values and the gradient that flows back into `t` through the duplicated rows of `perm`, for the
    channels-last views the featurizer emits and for a contiguous NCHW map, S = 13 (not a shape of the fused kernels), borders included."""
    from stego_amd.modules import sample, sample_indexed
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(5)
    M, C, H, W, S, n_rep = 6, 70, 9, 14, 13, 3
    t = torch.randn(M, C, H, W, generator=g).to(dev)
    if layout == "cl":
        t = t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    coords = (torch.rand(M, S, S, 2, generator=g) * 2.4 - 1.2).to(dev)          # beyond [-1, 1]: border padding
    idx = torch.tensor([3, 0, 3, 5, 1, 1] * n_rep, device=dev)                  # duplicates: gradients accumulate
    t_ref = t.detach().clone().requires_grad_(True)
    t_nat = t.detach().clone().requires_grad_(True)
    ref = sample(t_ref[idx], coords.repeat(n_rep, 1, 1, 1))
    got = sample_indexed(t_nat, coords, idx)
    assert got.shape == ref.shape
    assert float((got - ref).abs().max()) < 2e-6
    up = torch.randn(ref.shape, generator=g).to(dev)
    (ref * up).sum().backward()
    (got * up).sum().backward()
    assert float((t_nat.grad - t_ref.grad).abs().max()) < 2e-5 * float(t_ref.grad.abs().max())
    # no index: the plain sample()
    assert float((sample_indexed(t, coords) - sample(t, coords)).abs().max()) < 2e-6


@pytest.mark.parametrize("variant", ["default", "noclamp_stab", "plain"])
def test_native_pointwise_loss_matches_the_reference_statements(variant):
    """stego_rowsum + stego_loss_pointwise_fwd / _bwd (ABI 7) against helper()'s own statements (modules.py:330-345) in torch, per pair-set:
    the negative losses, every set's loss sum, and the gradient to cd for dense + per-set upstreams; zero_clamp / stabalize / pointwise variants."""
    from stego_amd.modules import _PointwiseLossFunction
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(9)
    n_sets, B, P = 4, 3, 37
    fd = (torch.randn(n_sets, B, P, P, generator=g) * 0.3).to(dev)
    cd0 = (torch.randn(n_sets, B, P, P, generator=g) * 0.6).to(dev)
    shifts = (0.18, 0.12, 0.46)
    zero_clamp, stab, pointwise = {"default": (True, False, True), "noclamp_stab": (False, True, True), "plain": (True, False, False)}[variant]
    cmin, cmax = (0.0 if zero_clamp else -9999.0), (0.8 if stab else 3.0e38)
    cd_ref = cd0.clone().requires_grad_(True)
    losses = []
    for s_ in range(n_sets):                                   # the reference, one helper() call per pair-set
        f = fd[s_].clone()
        if pointwise:
            old = f.mean()
            f = f - f.mean(dim=2, keepdim=True)                # fd.mean([3, 4]) of [n, h, w, i, j]: over the second set of points
            f = f - f.mean() + old
        c = cd_ref[s_]
        cl = c.clamp(cmin, 0.8) if stab else c.clamp(cmin)
        losses.append(-cl * (f - shifts[min(s_, 2)]))
    ref_neg = torch.stack(losses[2:])
    ref_sums = torch.stack([l.sum() for l in losses])
    cd_nat = cd0.clone().requires_grad_(True)
    neg, sums = _PointwiseLossFunction.apply(fd, cd_nat, shifts, cmin, cmax, pointwise)
    scale = float(ref_neg.abs().mean())
    assert float((neg - ref_neg).abs().max()) < 2e-6 * max(1.0, scale) + 2e-6
    assert float((sums - ref_sums).abs().max()) < 1e-4 * float(ref_sums.abs().max()) + 1e-4
    up_neg = torch.randn(ref_neg.shape, generator=g).to(dev)
    up_sums = torch.randn(n_sets, generator=g).to(dev)
    ((ref_neg * up_neg).sum() + (ref_sums * up_sums).sum()).backward()
    ((neg * up_neg).sum() + (sums * up_sums).sum()).backward()
    assert float((cd_nat.grad - cd_ref.grad).abs().max()) < 1e-5 * float(cd_ref.grad.abs().max()) + 1e-6
    # an expanded scalar upstream (the backward of .mean()): no dense copy is made
    cd_b = cd0.clone().requires_grad_(True)
    neg_b, _ = _PointwiseLossFunction.apply(fd, cd_b, shifts, cmin, cmax, pointwise)
    neg_b.mean().backward()
    cd_r = cd0.clone().requires_grad_(True)
    ls = []
    for s_ in range(2, n_sets):
        f = fd[s_].clone()
        if pointwise:
            old = f.mean()
            f = f - f.mean(dim=2, keepdim=True)
            f = f - f.mean() + old
        c = cd_r[s_]
        ls.append(-(c.clamp(cmin, 0.8) if stab else c.clamp(cmin)) * (f - shifts[2]))
    torch.stack(ls).mean().backward()
    assert float((cd_b.grad - cd_r.grad).abs().max()) < 1e-5 * float(cd_r.grad.abs().max()) + 1e-9


@pytest.mark.parametrize("S,variant", [(12, "default"), (16, "noclamp_stab"), (13, "plain")])
def test_wide_path_dense_upstreams_and_cfg_variants(S, variant):
    """feature_samples 12 .. 16 on csrc/corr_wide.hip with DENSE upstreams on every output (neg_inter_loss and the three cd tensors: the
    backward then scales G per pair by a first pass over it) and the zero_clamp / stabalize / pointwise variants: forward outputs and
    both code gradients against the fp64 oracle."""
    B, C, H, W, K, n_neg = 2, 64, 9, 11, 70, 2
    zero_clamp, stab, pointwise = {"default": (True, False, True), "noclamp_stab": (False, True, True), "plain": (True, False, False)}[variant]
    d = O.synth_inputs(B, C, H, W, K, S, n_neg, seed=77 + S, dino_like=True)
    cfg = O.CorrCfg(feature_samples=S, neg_samples=n_neg, zero_clamp=zero_clamp, stabalize=stab, pointwise=pointwise)
    assert M.ContrastiveCorrelationLoss.fused_kernels_cover(B, C, K, H, W, S)
    inputs = {k: d[k] for k in ("feats", "feats_pos", "code", "code_pos", "coords1", "coords2")}
    rng = np.random.default_rng(5)
    shp = (S,) * 4
    g_nl = rng.standard_normal((n_neg * B,) + shp) * 1e-3
    g_icd, g_ecd, g_ncd = (rng.standard_normal((m,) + shp) * 1e-3 for m in (B, B, n_neg * B))

    def upstream(out):
        t = lambda a: torch.from_numpy(a).float().to(DEV)                        # noqa: E731
        return 0.67 * out[0] + 0.25 * out[2] + (out[4] * t(g_nl)).sum() + (out[1] * t(g_icd)).sum() + (out[3] * t(g_ecd)).sum() + \
            (out[5] * t(g_ncd)).sum()

    r = _run(inputs, d["perms"], cfg, layout="cl", precision="f16x3", upstream=upstream)
    ref = O.corr_loss_forward(**inputs, perms=d["perms"], cfg=cfg)
    assert_close(r["out"][1], ref.pos_intra_cd, atol_frac=5e-4, what="intra_cd")
    assert_close(r["out"][4], ref.neg_inter_loss, atol_frac=5e-4, what="neg_loss")
    assert_close(r["out"][5], ref.neg_inter_cd, atol_frac=5e-4, what="neg_cd")
    scale = float(np.mean(np.abs(ref.neg_inter_loss)))
    assert abs(float(r["out"][0]) - float(ref.pos_intra_loss)) <= 1e-3 * scale + 1e-3 * abs(float(ref.pos_intra_loss))
    assert abs(float(r["out"][2]) - float(ref.pos_inter_loss)) <= 1e-3 * scale + 1e-3 * abs(float(ref.pos_inter_loss))
    dc, dcp = O.corr_loss_backward(**inputs, perms=d["perms"], cfg=cfg, g_intra=0.67, g_inter=0.25, g_neg_loss=g_nl,
                                   g_intra_cd=g_icd, g_inter_cd=g_ecd, g_neg_cd=g_ncd)
    assert_close(r["d_code"], dc, rtol=2e-3, atol_frac=1e-3, what="d_code")
    assert_close(r["d_code_pos"], dcp, rtol=2e-3, atol_frac=1e-3, what="d_code_pos")


@pytest.mark.parametrize("B,S,C,HW", [(32, 16, 384, 28), (16, 13, 384, 28), (5, 12, 768, 40)])
def test_wide_path_full_size_matches_the_composed_path(B, S, C, HW):
    """feature_samples 12 .. 16 at BASELINE sizes (cfg-2: B = 32, 28 x 28 x 384; cfg-4's map) - too large for the fp64 oracle in a test: the
    multi-launch kernels behind stego_corr_fwd / _bwd (csrc/corr_wide.hip) against generic_forward, the same arithmetic composed in Python
    from the native samplers / correlation kernels / batched GEMMs (itself pinned to the oracle at small sizes).  Every output and both code
    gradients; the two paths share the split-fp16 products, so they agree far inside the oracle bars."""
    import copy
    K, n_neg = 70, 5
    dev = DEV
    g = torch.Generator(device="cpu").manual_seed(100 + B + S)
    mk = lambda *shape: torch.randn(*shape, generator=g).to(dev)                  # noqa: E731
    f, fp = (mk(B, HW, HW, C).permute(0, 3, 1, 2) for _ in range(2))
    c0, cp0 = (mk(B, HW, HW, K).permute(0, 3, 1, 2) for _ in range(2))
    coords1 = (torch.rand(B, S, S, 2, generator=g) * 2 - 1).to(dev)
    coords2 = (torch.rand(B, S, S, 2, generator=g) * 2 - 1).to(dev)
    perms = torch.stack([M.super_perm(B, dev) for _ in range(n_neg)])
    cfg = O.CorrCfg(feature_samples=S, neg_samples=n_neg)
    cfg.corr_precision = "f16x3"
    loss = M.ContrastiveCorrelationLoss(cfg)
    assert loss.fused_kernels_cover(B, C, K, HW, HW, S)
    res = []
    for native in (True, False):
        c, cp = c0.detach().clone().requires_grad_(True), cp0.detach().clone().requires_grad_(True)
        out = loss.forward_explicit(f, fp, c, cp, coords1, coords2, perms) if native else loss.generic_forward(f, fp, c, cp, coords1, coords2, perms)
        (0.67 * out[0] + 0.25 * out[2] + 0.63 * out[4].mean()).backward()
        res.append(([o.detach().float() for o in out], c.grad.clone(), cp.grad.clone()))
    (on, gcn, gpn), (og, gcg, gpg) = res
    for i, what in enumerate(("intra mean", "intra cd", "inter mean", "inter cd", "neg loss", "neg cd")):
        a, b = on[i].reshape(-1), og[i].reshape(-1)
        assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(b.abs().max())) + 2e-7, what
    for a, b, what in ((gcn, gcg, "d_code"), (gpn, gpg, "d_code_pos")):
        assert float((a - b).abs().max()) <= 2e-4 * float(b.abs().max()), what          # (mask flips at cd == bound +- rounding aside: none expected, same products)
        assert float((a - b).norm()) <= 2e-5 * float(b.norm()), what


def test_wide_path_forward_is_repeatable_and_capturable():
    """csrc/corr_wide.hip: every kernel sums in a fixed order - the backward's pixel gather sorts its entries by key - so two calls agree bit for
    bit, outputs AND gradients (a pixel with more than 64 sample taps would be the exception: ~14 per pixel here), and the 8 + 4 launches hold no
    host synchronisation: a captured step replays with the same bits."""
    B, C, HW, K, S, n_neg = 8, 384, 14, 70, 13, 3
    g = torch.Generator(device="cpu").manual_seed(4)
    mk = lambda *shape: torch.randn(*shape, generator=g).to(DEV)                  # noqa: E731
    f, fp = (mk(B, HW, HW, C).permute(0, 3, 1, 2) for _ in range(2))
    c, cp = (mk(B, HW, HW, K).permute(0, 3, 1, 2).requires_grad_(True) for _ in range(2))
    coords1 = (torch.rand(B, S, S, 2, generator=g) * 2 - 1).to(DEV)
    coords2 = (torch.rand(B, S, S, 2, generator=g) * 2 - 1).to(DEV)
    perms = torch.stack([M.super_perm(B, DEV) for _ in range(n_neg)])
    cfg = O.CorrCfg(feature_samples=S, neg_samples=n_neg)
    cfg.corr_precision = "f16x3"
    loss = M.ContrastiveCorrelationLoss(cfg)

    def step():
        c.grad = None
        cp.grad = None
        out = loss.forward_explicit(f, fp, c, cp, coords1, coords2, perms)
        (0.67 * out[0] + 0.25 * out[2] + 0.63 * out[4].mean()).backward()
        return [o.detach().float().clone() for o in out], c.grad.clone(), cp.grad.clone()

    a, b = step(), step()
    for x, y in zip(a[0], b[0]):
        assert torch.equal(x, y)
    for x, y in zip(a[1:], b[1:]):
        assert torch.equal(x, y)
    with torch.no_grad():                        # nothing kept for a backward: fd and the context live in the workspace
        ng = loss.forward_explicit(f, fp, c.detach(), cp.detach(), coords1, coords2, perms)
    for x, y in zip(a[0], ng):
        assert torch.equal(x, y.detach().float())
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        got = step()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    for x, y in zip(a[0], got[0]):
        assert torch.equal(x, y)
    for x, y in zip(a[1:], got[1:]):
        assert torch.equal(x, y)


def test_wide_path_expanded_scalar_upstream():
    """feature_samples 13 with `neg_inter_loss.sum()` as the caller's reduction: autograd hands the backward ONE expanded scalar for the whole
    tensor (g_neg_loss_stride = 0 of the C ABI) - the third form of that upstream next to dense and "the mean's" (the other wide-path tests)."""
    B, C, H, W, K, S, n_neg = 2, 64, 7, 9, 70, 13, 3
    d = O.synth_inputs(B, C, H, W, K, S, n_neg, seed=313, dino_like=True)
    cfg = O.CorrCfg(feature_samples=S, neg_samples=n_neg)
    inputs = {k: d[k] for k in ("feats", "feats_pos", "code", "code_pos", "coords1", "coords2")}
    r = _run(inputs, d["perms"], cfg, layout="cl", precision="f16x3", upstream=lambda out: 0.4 * out[0] + 3e-4 * out[4].sum())
    g_nl = np.full((n_neg * B,) + (S,) * 4, 3e-4)
    dc, dcp = O.corr_loss_backward(**inputs, perms=d["perms"], cfg=cfg, g_intra=0.4, g_inter=0.0, g_neg_loss=g_nl)
    assert_close(r["d_code"], dc, rtol=2e-3, atol_frac=1e-3, what="d_code")
    assert_close(r["d_code_pos"], dcp, rtol=2e-3, atol_frac=1e-3, what="d_code_pos")


def test_wide_path_backward_on_maps_beyond_4096_pixels_and_crowded_pixels():
    """The pixel gather of csrc/corr_wide.hip serves maps up to 4096 pixels; a 70 x 60 map takes the scatter with fp32 atomics (same gradients to
    rounding).  And a map of 2 x 2 pixels puts far more than 64 taps on every pixel (the gather's chunked sums): both against the fp64 oracle."""
    for (H, W, S) in ((70, 60, 12), (2, 2, 13)):
        B, C, K, n_neg = 2, 16, 24, 2
        d = O.synth_inputs(B, C, H, W, K, S, n_neg, seed=55 + H, dino_like=False)
        cfg = O.CorrCfg(feature_samples=S, neg_samples=n_neg)
        inputs = {k: d[k] for k in ("feats", "feats_pos", "code", "code_pos", "coords1", "coords2")}
        r = _run(inputs, d["perms"], cfg, layout="cl", precision="f16x3")
        numel = B * S ** 4
        g_nl = np.full((n_neg * B,) + (S,) * 4, 0.63 / (n_neg * numel))
        dc, dcp = O.corr_loss_backward(**inputs, perms=d["perms"], cfg=cfg, g_intra=0.67, g_inter=0.25, g_neg_loss=g_nl)
        assert_close(r["d_code"], dc, rtol=2e-3, atol_frac=1e-3, what="d_code %dx%d" % (H, W))
        assert_close(r["d_code_pos"], dcp, rtol=2e-3, atol_frac=1e-3, what="d_code_pos %dx%d" % (H, W))


# ---------------------------------------------------------------------------------------------------------------------------------------
# round 6: the column-half launch of small batches (csrc/corr_fused_half.hip; taken when 16 B <= compute units)
def _half_launch_ran(desc, maps, d, extra_debug=0):
    """Runs stego_corr_fwd with the stamps on (STEGO_DEBUG 256) on a zeroed workspace and says whether a workgroup BEYOND the tiles left a
    start stamp: in the full-tile launch those are phase-1 helpers, which never stamp; in the column-half launch every workgroup does."""
    from ctypes import byref
    lib = capi.load()
    B, S, n_neg = desc.B, desc.S, desc.n_neg
    nt = (2 + n_neg) * B
    f32 = dict(dtype=torch.float32, device=DEV)
    outs = [torch.empty(3, **f32), torch.empty(B, S ** 4, **f32), torch.empty(B, S ** 4, **f32), torch.empty(max(1, n_neg * B), S ** 4, **f32),
            torch.empty(max(1, n_neg * B), S ** 4, **f32), torch.empty((2 + n_neg) * B, S ** 4, **f32), torch.empty(2 + n_neg, **f32)]
    ctx = torch.empty(lib.stego_corr_saved_ctx_bytes(byref(desc)), dtype=torch.uint8, device=DEV)
    ws = torch.zeros(lib.stego_corr_workspace_bytes(byref(desc)), dtype=torch.uint8, device=DEV)
    capi.debug_set("STEGO_DEBUG", 256 | extra_debug)
    try:
        rc = lib.stego_corr_fwd(byref(desc), *[byref(m) for m in maps], d["coords1"].data_ptr(), d["coords2"].data_ptr(),
                                d["perms"].data_ptr() if n_neg else None, *[o.data_ptr() for o in outs], ctx.data_ptr(), ws.data_ptr(), ws.numel(),
                                torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    finally:
        capi.debug_set("STEGO_DEBUG", 0)
    assert rc == 0, rc
    cus = torch.cuda.get_device_properties(0).multi_processor_count & ~7
    ts = ws[nt * 16 + 1024: nt * 16 + 1024 + cus * 128].view(torch.int64).view(cus, 16)
    return bool((ts[nt:, 0] != 0).any().item())


@pytest.mark.parametrize("shape", [
    dict(B=16, C=384, H=28, W=28, K=70, S=11, n_neg=5),      # the reference's batch size (train_config.yml:11): 224 items + 32 anchor workgroups
    dict(B=8, C=384, H=28, W=28, K=70, S=11, n_neg=5),
    dict(B=5, C=384, H=14, W=14, K=70, S=11, n_neg=5),       # not a multiple of 8: XCDs with one / no anchor
    dict(B=1, C=384, H=28, W=28, K=8, S=11, n_neg=0),        # one image, no negatives, one code chunk
    dict(B=3, C=384, H=9, W=11, K=128, S=10, n_neg=2),       # four code chunks, 100 points: the second half holds 36 columns
    dict(B=2, C=384, H=28, W=28, K=70, S=9, n_neg=5),        # 81 points: 17 columns in the second half
    dict(B=4, C=768, H=40, W=40, K=70, S=11, n_neg=5),       # ViT-B width (BASELINE config 4): two phase-1 passes per sampler
])
@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_column_half_launch_of_small_batches_equals_the_full_tile_launch_and_the_oracle(shape, precision):
    """16 B <= compute units: every column half of every tile runs on a compute unit of its own (corr_fused_half.hip).  The launch is
    taken (stamps of workgroups beyond the tiles), repeats bit for bit, gives the full-tile launch's self-correlation cd bit for bit and its
    other outputs to the last bits (a row mean is the sum of two halves; the two kernels' bilinear blends are contracted differently), and meets the fp64 oracle's bars forward and backward
    (modules.py:349-398, :325-347)."""
    # (seed 11 puts one code correlation of the B = 8 case 2e-8 from the clamp bound: its mask - hence 8 pixels of the gradient - differs
    # between fp32 and the fp64 oracle in BOTH launch kinds; tools/exp/r6_dbg_b8.py)
    d = O.synth_inputs(seed=12, dino_like=True, **shape)
    cfg = O.CorrCfg(feature_samples=shape["S"], neg_samples=shape["n_neg"])
    inputs = {k: d[k] for k in ("feats", "feats_pos", "code", "code_pos", "coords1", "coords2")}
    t = {k: _dev(v) for k, v in d.items() if k != "perms"}
    t["perms"] = _dev(d["perms"]) if shape["n_neg"] else None
    keep = [_channels_last(t[k]) for k in ("feats", "feats_pos", "code", "code_pos")]
    maps = [capi._map(x) for x in keep]
    desc = capi.make_desc(shape["B"], shape["C"], shape["K"], shape["H"], shape["W"], shape["S"], shape["n_neg"], cfg,
                          (cfg.pos_intra_shift, cfg.pos_inter_shift, cfg.neg_inter_shift), capi.PREC_F32 if precision == "f32" else capi.PREC_F16X3)
    assert _half_launch_ran(desc, maps, t)
    assert not _half_launch_ran(desc, maps, t, extra_debug=16384)       # (bit 16384: the full-tile launch of the same library)
    capi.debug_set("STEGO_DEBUG", 16384)
    try:
        full = _run(inputs, d["perms"], cfg, layout="cl", precision=precision)
    finally:
        capi.debug_set("STEGO_DEBUG", 0)
    r = _run(inputs, d["perms"], cfg, layout="cl", precision=precision)
    r2 = _run(inputs, d["perms"], cfg, layout="cl", precision=precision)
    for a, b in zip(r["out"], r2["out"]):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(r["d_code"], r2["d_code"])
    np.testing.assert_array_equal(r["out"][1], full["out"][1])          # intra cd: the anchors' operands are the same bytes, the products the same order
    for a, b in zip(r["out"], full["out"]):
        if b.size:
            np.testing.assert_allclose(a, b, rtol=2e-5, atol=3e-5 * float(np.abs(b).mean()))   # (fp32 rounding of a row mean: ~1e-7 absolute)
    ref = O.corr_loss_forward(**inputs, perms=d["perms"], cfg=cfg)
    la = 5e-4
    assert_close(r["out"][1], ref.pos_intra_cd, atol_frac=la, what="intra_cd")
    assert_close(r["out"][3], ref.pos_inter_cd, atol_frac=la, what="inter_cd")
    if shape["n_neg"]:
        assert_close(r["out"][4], ref.neg_inter_loss, atol_frac=la, what="neg_loss")
        assert_close(r["out"][5], ref.neg_inter_cd, atol_frac=la, what="neg_cd")
    assert abs(float(r["out"][0]) - float(ref.pos_intra_loss)) <= 1e-3 * abs(float(ref.pos_intra_loss)) + 1e-3 * float(np.abs(ref.neg_inter_loss).mean() if shape["n_neg"] else 1e-3)
    numel = shape["B"] * shape["S"] ** 4
    g_nl = np.full(ref.neg_inter_loss.shape, 0.63 / (shape["n_neg"] * numel)) if shape["n_neg"] else None
    dc, dcp = O.corr_loss_backward(**inputs, perms=d["perms"], cfg=cfg, g_intra=0.67, g_inter=0.25, g_neg_loss=g_nl)
    assert_close(r["d_code"], dc, rtol=1e-3, atol_frac=1e-3, what="d_code")
    assert_close(r["d_code_pos"], dcp, rtol=1e-3, atol_frac=1e-3, what="d_code_pos")


@pytest.mark.parametrize("variant", ["nopointwise", "stab", "noclamp"])
def test_column_half_launch_cfg_variants_against_the_oracle(variant):
    """pointwise off (no row means: the halves exchange nothing), stabalize (clamp at 0.8), zero_clamp off - modules.py:330-345."""
    kw = dict(nopointwise=dict(pointwise=False), stab=dict(stabalize=True), noclamp=dict(zero_clamp=False, stabalize=True))[variant]
    shape = dict(B=6, C=384, H=12, W=10, K=24, S=11, n_neg=3)
    d = O.synth_inputs(seed=3, dino_like=True, **shape)
    cfg = O.CorrCfg(feature_samples=11, neg_samples=3, **kw)
    inputs = {k: d[k] for k in ("feats", "feats_pos", "code", "code_pos", "coords1", "coords2")}
    for precision in ("f16x3", "f32"):
        r = _run(inputs, d["perms"], cfg, layout="cl", precision=precision)
        ref = O.corr_loss_forward(**inputs, perms=d["perms"], cfg=cfg)
        assert_close(r["out"][1], ref.pos_intra_cd, atol_frac=5e-4, what="intra_cd")
        assert_close(r["out"][4], ref.neg_inter_loss, atol_frac=5e-4, what="neg_loss")
        for i, want in ((0, ref.pos_intra_loss), (2, ref.pos_inter_loss)):
            assert abs(float(r["out"][i]) - float(want)) <= 1e-3 * abs(float(want)) + 1e-3 * float(np.abs(ref.neg_inter_loss).mean())
        numel = shape["B"] * 11 ** 4
        dc, dcp = O.corr_loss_backward(**inputs, perms=d["perms"], cfg=cfg, g_intra=0.67, g_inter=0.25,
                                       g_neg_loss=np.full(ref.neg_inter_loss.shape, 0.63 / (3 * numel)))
        assert_close(r["d_code"], dc, rtol=1e-3, atol_frac=1e-3, what="d_code")
        assert_close(r["d_code_pos"], dcp, rtol=1e-3, atol_frac=1e-3, what="d_code_pos")
