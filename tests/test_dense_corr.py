"""Dense feature correspondence (tensor_correlation, reference modules.py:283-284) on the HIP
kernel through the C ABI, against the reference's golden vector and a float64 einsum."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(__file__), "golden", "primitives.npz")


def _ref(a, b, normalize):
    a = a.astype(np.float64)
    b = b.astype(np.float64)
    if normalize:                                     # norm(): F.normalize(dim=1, eps=1e-10), modules.py:275-276
        a = a / np.maximum(np.sqrt((a * a).sum(1, keepdims=True)), 1e-10)
        b = b / np.maximum(np.sqrt((b * b).sum(1, keepdims=True)), 1e-10)
    return np.einsum("nchw,ncij->nhwij", a, b)


def test_reference_golden_tensor_correlation():
    from stego_amd import modules as M
    g = np.load(GOLD)
    out = M.tensor_correlation(torch.from_numpy(g["a"]).to(DEV), torch.from_numpy(g["b"]).to(DEV))
    assert tuple(out.shape) == g["corr"].shape
    np.testing.assert_allclose(out.cpu().numpy(), g["corr"], rtol=1e-5, atol=1e-6)
    # with autograd the reference einsum is used, same numbers
    a = torch.from_numpy(g["a"]).to(DEV).requires_grad_(True)
    np.testing.assert_allclose(M.tensor_correlation(a, torch.from_numpy(g["b"]).to(DEV)).detach().cpu().numpy(), g["corr"],
                               rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("shape", [
    (2, 384, 28, 28, 28, 28, True),      # ViT-S/8 at 224^2: the [B, 784, 784] correspondence, cosine
    (1, 768, 40, 40, 40, 40, True),      # ViT-B/8 at 320^2
    (3, 70, 5, 9, 11, 3, False),         # unequal maps, C straddling the 64-wide chunk, raw dot products
    (2, 5, 1, 1, 2, 200, False),         # 1-pixel map against a wide one
])
@pytest.mark.parametrize("layout", ["cl", "nchw"])
def test_dense_corr_matches_float64_einsum(shape, layout):
    from stego_amd import capi
    B, C, H1, W1, H2, W2, normalize = shape
    rng = np.random.default_rng(B * C + H1)
    a = rng.standard_normal((B, C, H1, W1)).astype(np.float32)
    b = rng.standard_normal((B, C, H2, W2)).astype(np.float32)
    ta, tb = torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV)
    if layout == "cl":
        ta = ta.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        tb = tb.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    out = capi.dense_corr(ta, tb, normalize=normalize).cpu().numpy()
    ref = _ref(a, b, normalize)
    scale = np.abs(ref).mean()
    np.testing.assert_allclose(out, ref, rtol=1e-3, atol=2e-5 * max(scale, 1e-3) + 1e-6)
    if normalize and (H1, W1) == (H2, W2):
        self_sim = capi.dense_corr(ta, ta, normalize=True).cpu().numpy().reshape(B, H1 * W1, H1 * W1)
        np.testing.assert_allclose(np.diagonal(self_sim, axis1=1, axis2=2), 1.0, atol=2e-6)      # unit self-similarity
        np.testing.assert_allclose(self_sim, self_sim.transpose(0, 2, 1), atol=1e-6)              # symmetric


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [
    (3, 192, 14, 14, 14, 14, True),      # three chunks (an odd count), one whole block of B + a tail of 68 pixels
    (2, 136, 9, 13, 7, 11, False),       # the last chunk holds 8 channels; 117 rows (a partial row block), 77 columns: a tail only, N % 4 = 1 (scalar stores)
    (9, 328, 12, 12, 20, 20, True),      # two groups of images; 144 rows (a block + 16), 400 columns (three blocks + 16), five chunks + 8 channels
    (1, 72, 16, 16, 16, 16, True),       # two chunks, exactly two blocks, no tail
    (2, 256, 3, 50, 50, 3, False),       # 150 x 150, N % 4 = 2
])
def test_dense_stream_kernel_shapes(shape):
    """csrc/dense_stream.hip (channels-last maps, 64 < C <= 384, C % 8 == 0) on the shapes its special cases exist for, against fp64."""
    from stego_amd import capi
    B, C, H1, W1, H2, W2, normalize = shape
    rng = np.random.default_rng(B * C + H1 + 7)
    a = (rng.standard_normal((B, C, H1, W1)) * rng.uniform(0.2, 5.0, (B, 1, H1, W1))).astype(np.float32)      # pixel norms vary
    b = (rng.standard_normal((B, C, H2, W2)) * rng.uniform(0.2, 5.0, (B, 1, H2, W2))).astype(np.float32)
    ta = torch.from_numpy(a).to(DEV).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    tb = torch.from_numpy(b).to(DEV).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    out = capi.dense_corr(ta, tb, normalize=normalize).cpu().numpy()
    ref = _ref(a, b, normalize)
    scale = np.abs(ref).mean()
    np.testing.assert_allclose(out, ref, rtol=1e-3, atol=2e-5 * max(scale, 1e-3) + 1e-6)
    err = np.abs(out - ref).max() / max(np.abs(ref).max(), 1e-30)
    assert err < 2e-6, err                                                                      # fp32 class, not just the north-star bar


@pytest.mark.gpu
@pytest.mark.parametrize("scale", [1e-7, 1.0, 1e6])
def test_dense_stream_raw_products_any_magnitude(scale):
    """The streaming kernel's fp16 staging (a power of two per pixel from dense_stats_kernel) on tiny and huge raw maps."""
    from stego_amd import capi
    rng = np.random.default_rng(5)
    a = (rng.standard_normal((2, 96, 6, 7)) * scale).astype(np.float32)
    b = (rng.standard_normal((2, 96, 5, 4)) * scale).astype(np.float32)
    ta = torch.from_numpy(a).to(DEV).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    tb = torch.from_numpy(b).to(DEV).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    out = capi.dense_corr(ta, tb).cpu().numpy()
    ref = _ref(a, b, False)
    np.testing.assert_allclose(out, ref, rtol=1e-4, atol=1e-5 * np.abs(ref).mean())


@pytest.mark.parametrize("scale", [1e-7, 1.0, 1e6])
def test_dense_corr_raw_products_any_magnitude(scale):
    """Without norm() the products are raw: the fp16 staging must not lose tiny maps or overflow on huge ones."""
    from stego_amd import capi
    rng = np.random.default_rng(3)
    a = (rng.standard_normal((2, 96, 6, 7)) * scale).astype(np.float32)
    b = (rng.standard_normal((2, 96, 5, 4)) * scale).astype(np.float32)
    out = capi.dense_corr(torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV)).cpu().numpy()
    ref = _ref(a, b, False)
    np.testing.assert_allclose(out, ref, rtol=1e-4, atol=1e-5 * np.abs(ref).mean())


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 64, 7, 9, 5, 6), (3, 384, 28, 28, 28, 28), (1, 70, 11, 11, 20, 20)])
def test_tensor_correlation_gradients_run_on_the_native_kernel(shape):
    """modules.tensor_correlation with autograd (reference modules.py:283-284: the einsum is differentiable there): both
    adjoints are the dense-correspondence kernel on strided views; compared with autograd through the reference's einsum."""
    import torch
    from stego_amd import modules as M
    N, C, H1, W1, H2, W2 = shape
    g = torch.Generator(device="cuda").manual_seed(5)
    a = torch.randn(N, C, H1, W1, device="cuda", generator=g)
    b = torch.randn(N, H2, W2, C, device="cuda", generator=g).permute(0, 3, 1, 2)       # a channels-last view, as DINO emits
    up = torch.randn(N, H1, W1, H2, W2, device="cuda", generator=g)
    a1, b1 = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    out = M.tensor_correlation(a1, b1)
    assert out.grad_fn is not None and "DenseCorr" in type(out.grad_fn).__name__
    (out * up).sum().backward()
    a2, b2 = a.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = torch.einsum("nchw,ncij->nhwij", a2, b2)
    (ref * up.double()).sum().backward()
    for got, want in ((out, ref), (a1.grad, a2.grad), (b1.grad, b2.grad)):
        err = (got.double() - want).abs().max() / want.abs().max()
        assert float(err) < 2e-6, float(err)
