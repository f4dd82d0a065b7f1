"""The lists-first backward (corr_bwd_tile_build_kernel: builders beside the tiles; corr_unsample_list_kernel: one wave per
destination unit, csrc/corr_bwd.hip) against the fp64 oracle's autograd restatement (reference: autograd through
src/modules.py:325-347, 369-391) and against the tile + row kernels it replaces, through the C ABI.  Needs the MI355X."""
import os

import numpy as np
import pytest
import torch

from conftest import assert_close
from oracle import corr_oracle as O
from stego_amd import capi

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
BASE = int(os.environ.get("STEGO_DEBUG_BWD", "0"))     # (tools: variants of the kernels under test)
TWO_LAUNCHES = 1024          # STEGO_DEBUG_BWD bit: take the plain tile kernel + the row kernel (three dependent rounds per unit)


def _dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def _cl(t):
    return t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)


class _Case:
    """Forward once through the C ABI (saved context kept), then as many backwards as a test wants."""

    def __init__(self, d, cfg, precision=capi.PREC_F16X3):
        self.d, self.cfg = d, cfg
        B, C, H, W = d["feats"].shape
        K = d["code"].shape[1]
        S = d["coords1"].shape[1]
        n_neg = d["perms"].shape[0]
        self.dims = (B, C, H, W, K, S, n_neg)
        self.desc = capi.make_desc(B, C, K, H, W, S, n_neg, cfg, (cfg.pos_intra_shift, cfg.pos_inter_shift, cfg.neg_inter_shift), precision)
        t = {k: _dev(v) for k, v in d.items()}
        self.t = t
        self.f, self.fp, self.c, self.cp = (_cl(t[k]) for k in ("feats", "feats_pos", "code", "code_pos"))
        out = capi.corr_fwd(self.desc, self.f, self.fp, self.c, self.cp, t["coords1"], t["coords2"], t["perms"] if n_neg else None, True)
        self.lm, self.icd, self.ecd, self.nl, self.ncd, self.saved = out
        torch.cuda.synchronize()

    def backward(self, g_intra=0.67, g_inter=0.25, g_neg=0.63, two_launches=False):
        B, C, H, W, K, S, n_neg = self.dims
        t = self.t
        capi.debug_set("STEGO_DEBUG_BWD", TWO_LAUNCHES if two_launches else BASE)
        try:
            gi = torch.tensor(g_intra, device=DEV)
            ge = torch.tensor(g_inter, device=DEV)
            gn = torch.full((1,), g_neg / max(n_neg * B * S ** 4, 1), device=DEV) if n_neg else None
            gnl = gn.expand(n_neg * B, S, S, S, S) if n_neg else None
            dc, dcp = capi.corr_bwd(self.desc, self.c, self.cp, t["coords1"], t["coords2"], t["perms"] if n_neg else None, self.saved,
                                    self.icd, self.ecd, self.ncd, gi, ge, gnl, None, None, None)
            torch.cuda.synchronize()
        finally:
            capi.debug_set("STEGO_DEBUG_BWD", BASE)
        return dc.contiguous().cpu().numpy(), dcp.contiguous().cpu().numpy()

    def oracle(self, g_intra=0.67, g_inter=0.25, g_neg=0.63):
        B, C, H, W, K, S, n_neg = self.dims
        d = self.d
        g_nl = np.full((n_neg * B, S, S, S, S), g_neg / max(n_neg * B * S ** 4, 1))
        return O.corr_loss_backward(d["feats"], d["feats_pos"], d["code"], d["code_pos"], d["coords1"], d["coords2"], d["perms"],
                                    self.cfg, g_intra, g_inter, g_nl)


def test_cfg2_full_size_lists_first_equals_the_row_kernel_and_is_repeatable():
    """BASELINE config 2 at its real size.  Same tile arithmetic in both paths; the unsample sums the same rows in another (fixed)
    order: agreement to fp32 rounding of a ~12-term sum, and bit for bit from run to run and on a re-used saved context."""
    B, C, H, W, K, S, n_neg = 32, 384, 28, 28, 70, 11, 5
    d = O.synth_inputs(B, C, H, W, K, S, n_neg, seed=2024, dino_like=True)
    case = _Case(d, O.CorrCfg())
    a1 = case.backward()
    a2 = case.backward()                     # same saved context again (retain_graph)
    b = case.backward(two_launches=True)
    for x, y, z, name in ((a1[0], a2[0], b[0], "d_code"), (a1[1], a2[1], b[1], "d_code_pos")):
        assert np.array_equal(x, y), name + ": not repeatable"
        scale = float(np.abs(z).max())
        assert np.abs(x - z).max() <= 2e-6 * scale, (name, np.abs(x - z).max(), scale)
    dc, dcp = case.oracle()
    assert_close(a1[0], dc, rtol=1e-3, atol_frac=1e-3, what="d_code")
    assert_close(a1[1], dcp, rtol=1e-3, atol_frac=1e-3, what="d_code_pos")


@pytest.mark.parametrize("shape", [
    (1, 32, 8, 8, 8, 5, 0),          # one image, no negatives: builders / tiles / units at their minimum
    (3, 32, 6, 9, 8, 5, 2),          # H != W, a partial 16-pixel bin
    (4, 64, 40, 40, 70, 11, 5),      # cfg-4's map: three bins per row
    (2, 32, 64, 64, 16, 7, 3),       # the largest map of the lists-first path: four bins, 64 rows
    (2, 384, 12, 12, 100, 6, 2),     # K = 100: seven channel tiles, two operand groups in the tile role
    (5, 32, 16, 16, 70, 11, 5),      # S = 11 on a small map: long lists per unit (more than one step of 64 entries)
    (40, 32, 8, 8, 24, 5, 5),        # more builders' images than builder workgroups (32)
])
def test_lists_first_backward_edge_shapes_against_the_oracle(shape):
    B, C, H, W, K, S, n_neg = shape
    d = O.synth_inputs(B, C, H, W, K, S, n_neg, seed=77 + B + W)
    cfg = O.CorrCfg(feature_samples=S, neg_samples=n_neg)
    case = _Case(d, cfg)
    a = case.backward()
    b = case.backward(two_launches=True)
    dc, dcp = case.oracle()
    assert_close(a[0], dc, rtol=1e-3, atol_frac=1e-3, what="d_code")
    assert_close(a[1], dcp, rtol=1e-3, atol_frac=1e-3, what="d_code_pos")
    for x, z in zip(a, b):
        assert np.abs(x - z).max() <= 4e-6 * float(np.abs(z).max()) + 1e-30


def test_every_negative_from_one_image_and_every_point_on_one_row():
    """The worst lists: all perms pick image 0 (7 + n_neg B items on one destination image: lists far beyond their slots, served from the overflow pool), and the
    points of some images all sample the border row (coords beyond [-1, 1]: one pixel row collects everything)."""
    B, C, H, W, K, S, n_neg = 16, 32, 12, 20, 24, 7, 5
    d = O.synth_inputs(B, C, H, W, K, S, n_neg, seed=5)
    perms = np.zeros((n_neg, B), np.int64)
    perms[:, 0] = 1                                       # (no fixed points: super_perm never maps b to b)
    d["perms"] = perms
    d["coords2"][3] = 1.7                                 # clipped to the last row / column
    d["coords1"][5, :, :, 1] = -2.0                       # the first row
    d["coords2"][7, :, :, 0] = -1.0                       # exactly the first column
    cfg = O.CorrCfg(feature_samples=S, neg_samples=n_neg)
    case = _Case(d, cfg)
    a = case.backward()
    b = case.backward(two_launches=True)
    dc, dcp = case.oracle()
    assert np.isfinite(a[0]).all() and np.isfinite(a[1]).all()
    assert_close(a[0], dc, rtol=1e-3, atol_frac=1e-3, what="d_code")
    assert_close(a[1], dcp, rtol=1e-3, atol_frac=1e-3, what="d_code_pos")
    for x, z in zip(a, b):
        assert np.abs(x - z).max() <= 4e-6 * float(np.abs(z).max())


def test_lists_first_backward_is_linear_in_the_upstreams_and_zero_without_them():
    B, C, H, W, K, S, n_neg = 4, 32, 8, 8, 16, 5, 2
    d = O.synth_inputs(B, C, H, W, K, S, n_neg, seed=33)
    case = _Case(d, O.CorrCfg(feature_samples=S, neg_samples=n_neg))
    z = case.backward(0.0, 0.0, 0.0)
    assert not z[0].any() and not z[1].any()
    g1 = case.backward(1.0, 0.0, 0.0)
    g2 = case.backward(0.0, 1.0, 1.0)
    g3 = case.backward(2.0, -3.0, -3.0)
    np.testing.assert_allclose(g3[0], 2.0 * g1[0] - 3.0 * g2[0], rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(g3[1], 2.0 * g1[1] - 3.0 * g2[1], rtol=2e-4, atol=1e-7)


def test_back_to_back_backwards_on_rotating_contexts_never_see_stale_rows():
    """Launch after launch on workspaces the allocator recycles: a unit that read a DT row or a list of an EARLIER launch would
    differ from the same backward run alone."""
    B, C, H, W, K, S, n_neg = 32, 384, 28, 28, 70, 11, 5
    cases = [_Case(O.synth_inputs(B, C, H, W, K, S, n_neg, seed=100 + i), O.CorrCfg()) for i in range(3)]
    alone = [c.backward() for c in cases]
    torch.cuda.synchronize()
    for rep in range(15):
        outs = []
        for c in cases:                      # no synchronisation in between: launches queue back to back
            t = c.t
            gi = torch.tensor(0.67, device=DEV); ge = torch.tensor(0.25, device=DEV)
            gn = torch.full((1,), 0.63 / (n_neg * B * S ** 4), device=DEV).expand(n_neg * B, S, S, S, S)
            outs.append(capi.corr_bwd(c.desc, c.c, c.cp, t["coords1"], t["coords2"], t["perms"], c.saved, c.icd, c.ecd, c.ncd,
                                      gi, ge, gn, None, None, None))
        torch.cuda.synchronize()
        for (dc, dcp), (rc, rcp) in zip(outs, alone):
            assert np.array_equal(dc.contiguous().cpu().numpy(), rc), rep
            assert np.array_equal(dcp.contiguous().cpu().numpy(), rcp), rep


@pytest.mark.parametrize("shape", [(32, 384, 28, 28, 70, 11, 5), (3, 32, 6, 9, 8, 5, 2), (4, 64, 40, 40, 70, 11, 5), (40, 32, 8, 8, 24, 5, 5)])
def test_lists_first_backward_in_exact_fp32_mode(shape):
    """F32 mode: the exact-fp32 tile kernel with four-wave builders beside it, the same unsample kernel."""
    B, C, H, W, K, S, n_neg = shape
    # (seeds whose cd has no element within fp32 rounding of the clamp bound: there the pass mask of any fp32 evaluation - the reference's
    # included - may differ from the fp64 oracle's, one whole term of a gradient sum)
    d = O.synth_inputs(B, C, H, W, K, S, n_neg, seed=2024 if C == 384 else 11 + B + W, dino_like=C == 384)
    cfg = O.CorrCfg(feature_samples=S, neg_samples=n_neg)
    case = _Case(d, cfg, precision=capi.PREC_F32)
    a = case.backward()
    a2 = case.backward()
    b = case.backward(two_launches=True)
    dc, dcp = case.oracle()
    assert_close(a[0], dc, rtol=1e-3, atol_frac=1e-3, what="d_code")
    assert_close(a[1], dcp, rtol=1e-3, atol_frac=1e-3, what="d_code_pos")
    for x, y, z in zip(a, a2, b):
        assert np.array_equal(x, y)
        assert np.abs(x - z).max() <= 4e-6 * float(np.abs(z).max()) + 1e-30
