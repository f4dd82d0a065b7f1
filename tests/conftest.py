import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


class GoldenCase:
    """A golden fixture (made by oracle/make_golden.py from the reference) turned back
    into inputs + cfg.  Inputs are stored, or regenerated from the seed and checked
    against the stored checksums."""

    def __init__(self, name):
        from oracle.corr_oracle import CorrCfg, synth_inputs
        g = load_golden(name)
        self.g = g
        self.name = name
        B, C, H, W, K, S, n_neg, seed, dino_like, subsample = [int(v) for v in g["meta"]]
        self.B, self.C, self.H, self.W, self.K, self.S, self.n_neg = B, C, H, W, K, S, n_neg
        self.subsample = subsample
        pw, zc, st = [bool(v) for v in g["cfg_flags"]]
        sh = [float(v) for v in g["shifts"]]
        self.cfg = CorrCfg(pointwise=pw, zero_clamp=zc, stabalize=st, feature_samples=S, neg_samples=n_neg,
                           pos_intra_shift=sh[0], pos_inter_shift=sh[1], neg_inter_shift=sh[2])
        if "in_feats" in g:
            self.inputs = {k: g["in_" + k] for k in ("feats", "feats_pos", "code", "code_pos", "coords1", "coords2")}
        else:
            d = synth_inputs(B, C, H, W, K, S, n_neg, seed, dino_like=bool(dino_like))
            chk = np.array([float(np.abs(d[k].astype(np.float64)).sum()) for k in
                            ("feats", "feats_pos", "code", "code_pos", "coords1", "coords2")])
            np.testing.assert_allclose(chk, g["input_checksum"], rtol=1e-12,
                                       err_msg="seeded input regeneration drifted (numpy RNG change?)")
            assert np.array_equal(d["perms"], g["perms"])
            self.inputs = {k: d[k] for k in ("feats", "feats_pos", "code", "code_pos", "coords1", "coords2")}
        self.perms = g["perms"]

    def sub(self, x):
        return np.asarray(x).reshape(-1)[::self.subsample]


ALL_CASES = ["small_default", "small_nopointwise", "small_noclamp_stab", "small_stab",
             "small_dinolike_S11", "small_noneg", "cfg1_B4_vits8", "cfg1_B4_vits8_dinolike", "cfg4_B2_vitb8",
             "wide_S12_small", "wide_S16_vits8"]        # (round 5: feature_samples 12 / 16, the shapes of csrc/corr_wide.hip)
SMALL_CASES = [c for c in ALL_CASES if c.startswith("small")]


def assert_close(actual, expected, rtol=1e-3, atol_frac=1e-4, what=""):
    """Elementwise |a-e| <= atol + rtol*|e| with atol = atol_frac * mean|e|
    (north_star: 1e-3 fp32 relative; the small atol covers cancellation near zero)."""
    a = np.asarray(actual, dtype=np.float64).reshape(-1)
    e = np.asarray(expected, dtype=np.float64).reshape(-1)
    assert a.shape == e.shape, (what, a.shape, e.shape)
    if e.size == 0:
        return
    atol = atol_frac * float(np.mean(np.abs(e))) + 1e-12
    bad = np.abs(a - e) > atol + rtol * np.abs(e)
    if bad.any():
        i = int(np.argmax(np.abs(a - e) - rtol * np.abs(e)))
        raise AssertionError("%s: %d/%d mismatches (atol=%.3e rtol=%.1e); worst idx %d actual=%.9g expected=%.9g"
                             % (what, int(bad.sum()), e.size, atol, rtol, i, a[i], e[i]))
