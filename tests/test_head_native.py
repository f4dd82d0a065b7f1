"""the segmentation head (modules.py:108-116) on the hand-written kernels of include/stego_head.h against the
reference's convolution head with the same Dropout2d draws, forward and all six parameter gradients."""
import types

import numpy as np
import pytest
import torch

from conftest import assert_close
from stego_amd import capi
from stego_amd.featurizers import DinoFeaturizer
from stego_amd.train_segmentation import load_config

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _featurizer(arch, dim, proj, dropout, seed=0):
    cfg = load_config(overrides=["model_type=%s" % arch, "dino_patch_size=8", "dim=%d" % dim, "projection_type=%s" % proj,
                                 "dropout=%s" % dropout, "native_backbone=False"])
    torch.manual_seed(seed)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        net = DinoFeaturizer(dim, cfg).to(DEV)
    return net, cfg


def _reference_head_fp64(net, image_feat, masks):
    """modules.py:108-116 in double precision with explicit channel masks: conv1x1(x * m) = (x * m) W^T + b over the tokens.
    Also returns, for a nonlinear head, which hidden channels have a pre-activation within fp32 rounding of zero somewhere: the sign
    of such a value - hence relu' - is not defined at fp32 accuracy (fp32 torch and the kernels flip the same few), and one flipped
    (token, channel) moves that channel's row of cluster2[0]'s weight gradient by far more than any rounding error."""
    x = image_feat.double()
    m1, m2, m3 = [m.double().view(m.shape[0], -1, 1, 1) for m in masks]
    c1 = net.cluster1[0]
    conv = lambda c, t: torch.nn.functional.conv2d(t, c.weight.double(), c.bias.double())
    code = conv(c1, x * m1)
    ambiguous = None
    if net.proj_type == "nonlinear":
        pre = conv(net.cluster2[0], x * m2)
        ambiguous = (pre.detach().abs() < 1e-5).permute(1, 0, 2, 3).flatten(1).any(1)
        code = code + conv(net.cluster2[2], torch.relu(pre))
    return x * m3, code, ambiguous


@pytest.mark.parametrize("arch,B,hw,dim,proj", [("vit_small", 4, (14, 14), 70, "nonlinear"), ("vit_small", 3, (9, 13), 70, "linear"),
                                                 ("vit_base", 2, (10, 10), 100, "nonlinear"), ("vit_small", 32, (28, 28), 70, "nonlinear")])
def test_native_head_matches_the_reference_head_with_the_same_dropout_draws(arch, B, hw, dim, proj):
    net, cfg = _featurizer(arch, dim, proj, True)
    net.train()
    C = net.n_feats
    g = torch.Generator(device=DEV).manual_seed(5)
    tokens = torch.randn(B, 1 + hw[0] * hw[1], C, device=DEV, generator=g)
    image_feat = tokens[:, 1:, :].reshape(B, hw[0], hw[1], C).permute(0, 3, 1, 2)       # what forward() hands the head (modules.py:97)
    up = torch.randn(B, dim, hw[0], hw[1], device=DEV, generator=g) / (hw[0] * hw[1])
    assert net._native_head_ok(image_feat)
    params = [p for n, p in net.named_parameters() if n.startswith("cluster")]

    def run(native):
        cfg.native_head = native
        for p in params:
            p.grad = None
        torch.manual_seed(77)
        if native:
            feats, code = net._head_native(image_feat)
        else:
            code = net._head(image_feat)
            feats = net.dropout(image_feat)
        (code * up).sum().backward()
        return feats.detach().clone(), code.detach().clone(), [p.grad.detach().clone() for p in params], torch.rand(2, device=DEV)

    f_n, c_n, g_n, r_n = run(True)
    f_t, c_t, g_t, r_t = run(False)
    assert torch.equal(r_n, r_t)                       # both consumed the generator alike
    assert torch.equal(f_n, f_t)                       # the returned dropout(image_feat): same mask, same product
    # masks as drawn (same seed, same calls), for the fp64 reference
    torch.manual_seed(77)
    masks = [net._feature_noise(image_feat).view(B, C) for _ in range(3 if proj == "nonlinear" else 1)]
    if proj != "nonlinear":
        masks = [masks[0], masks[0], net._feature_noise(image_feat).view(B, C)]
    x64 = image_feat.double().requires_grad_(False)
    for p in params:
        p.grad = None
    f64, c64, ambiguous = _reference_head_fp64(net, x64, masks)
    net64_grads = torch.autograd.grad((c64 * up.double()).sum(), params, allow_unused=True)
    assert_close(c_n.cpu().numpy(), c64.detach().cpu().numpy(), rtol=1e-3, atol_frac=1e-4, what="code")
    err_n = float((c_n.double() - c64).abs().max())
    err_t = float((c_t.double() - c64).abs().max())
    assert err_n <= 2.0 * err_t + 1e-6, (err_n, err_t)           # the split-fp16 head is in the fp32 class of the torch head
    for (name, p), gn, gt, g64 in zip([(n, p) for n, p in net.named_parameters() if n.startswith("cluster")], g_n, g_t, net64_grads):
        if name.startswith("cluster2.0.") and ambiguous is not None and bool(ambiguous.any()):
            keep = ~ambiguous                     # rows of dW21 / entries of db21 whose relu' is well defined
            assert int(keep.sum()) >= 0.5 * keep.numel()
            gn, gt, g64 = gn[keep], gt[keep], g64[keep]
        assert_close(gn.cpu().numpy(), g64.detach().cpu().numpy(), rtol=1e-3, atol_frac=2e-4, what="grad %s" % (tuple(p.shape),))
        en, et = float((gn.double() - g64).abs().max()), float((gt.double() - g64).abs().max())
        assert en <= 3.0 * et + 1e-7 * float(g64.abs().max()) + 1e-12, (tuple(p.shape), en, et)


def test_native_head_in_eval_mode_and_bitwise_repeatable():
    net, cfg = _featurizer("vit_small", 70, "nonlinear", True)
    net.eval()
    B, hw, C = 5, (12, 12), 384
    tokens = torch.randn(B, 1 + 144, C, device=DEV)
    image_feat = tokens[:, 1:, :].reshape(B, 12, 12, C).permute(0, 3, 1, 2)
    with torch.no_grad():
        f1, c1 = net._head_native(image_feat)
        f2, c2 = net._head_native(image_feat)
        cfg.native_head = False
        ct = net._head(image_feat)
    assert f1.data_ptr() == image_feat.data_ptr()          # eval: nn.Dropout2d is the identity, the map itself is returned
    assert torch.equal(c1, c2)
    assert_close(c1.cpu().numpy(), ct.cpu().numpy(), rtol=1e-3, atol_frac=1e-4, what="code (eval)")


def test_head_argument_checks_need_no_launch():
    import ctypes
    lib = capi.load()
    d = capi.StegoHeadDesc(2, 49, 100, 70, 1, 100, 50 * 100)               # C not a multiple of 32
    assert lib.stego_head_fwd_workspace_bytes(ctypes.byref(d)) == 0
    d = capi.StegoHeadDesc(2, 49, 384, 70, 1, 384, 50 * 384)
    assert lib.stego_head_fwd_workspace_bytes(ctypes.byref(d)) >= 2 * 49 * 384 * 4
    assert lib.stego_head_fwd(ctypes.byref(d), *([None] * 14), 0, None) == 1   # STEGO_ERR_NULL


@pytest.mark.parametrize("B,C,n", [(64, 384, 3), (32, 384, 3), (4, 768, 2), (3, 96, 1)])
def test_one_launch_dropout_masks_are_the_torch_generator_s_bernoulli_draws(B, C, n):
    """stego_ref_dropout_masks against nn.Dropout2d's own draws (x.new_empty(B, C, 1, 1).bernoulli_(1 - p).div_(1 - p), n calls) from the
    same generator state over many seeds: every mask value equal, the generator left at the same offset - eagerly and replayed from a
    HIP graph (the captured launch reads the generator's graph-safe state)."""
    from stego_amd import capi, featurizers as FZ
    from stego_amd.modules import _device_generator
    net, cfg = _featurizer("vit_small", 70, "nonlinear", True)
    dev = torch.device(DEV)
    x = torch.zeros(B, C, 2, 2, device=dev)
    q = 1.0 - float(net.dropout.p)
    assert capi.torchglue() is not None
    v = FZ._mask_variant(net, x, n, q)
    assert v >= 0, "no variant of stego_ref_dropout_masks reproduces this torch build: DinoFeaturizer would keep the torch calls"
    gen = _device_generator(dev)
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    for seed in range(300):
        torch.manual_seed(4000 + seed)
        if seed % 3 == 1:
            torch.rand(5, device=dev)
        st = gen.get_state()
        want = torch.stack([net._feature_noise(x).view(-1) for _ in range(n)])
        off = gen.get_offset()
        gen.set_state(st)
        got = capi.ref_dropout_masks(gen, n, B * C, q, v, dev)
        assert gen.get_offset() == off
        bad += (got != want).sum()
    assert int(bad) == 0
    assert 0.05 < float((want == 0).float().mean()) < 0.16          # (p = 0.1 of the channels dropped)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        got = capi.ref_dropout_masks(gen, n, B * C, q, v, dev)
    for seed in range(10):
        torch.manual_seed(7000 + seed)
        st = gen.get_state()
        want = [torch.stack([net._feature_noise(x).view(-1) for _ in range(n)]) for _ in range(2)]
        off = gen.get_offset()
        gen.set_state(st)
        for k in range(2):
            g.replay()
            assert torch.equal(got, want[k])
        assert gen.get_offset() == off


def test_cached_tokens_arrive_converted_with_their_magnitude_word():
    """stego_tokens_from_cache (TokenCache.fetch on a HIP device): rows of the fp16 table as fp32 + the largest magnitude of the patch
    tokens (class token skipped) in one pass; the head run with that word (StegoHeadDesc.tokens_amax) is bitwise the head that scans
    the tokens itself."""
    from stego_amd import capi
    dev = torch.device(DEV)
    g = torch.Generator(device=dev).manual_seed(3)
    table = (torch.randn(10, 1 + 49, 384, device=dev, generator=g) * 3).half()
    table[:, 0, :] *= 50                                   # a class token larger than every patch token
    index = torch.tensor([7, 2, 2, 9], device=dev)
    out, amax = capi.tokens_from_cache(table, index, skip_rows=1)
    assert torch.equal(out, table[index].float())
    assert float(amax.view(torch.float32)) == float(table[index][:, 1:, :].float().abs().max())
    net, cfg = _featurizer("vit_small", 70, "nonlinear", True)
    net.eval()
    image_feat = out[:, 1:, :].reshape(4, 7, 7, 384).permute(0, 3, 1, 2)
    with torch.no_grad():
        f0, c0 = net._head_native(image_feat)
        f1, c1 = net._head_native(image_feat, amax)
    assert torch.equal(c0, c1) and torch.equal(f0, f1)


def test_cpp_head_function_is_the_python_one_bit_for_bit():
    """csrc/torch_glue_ext.cpp::HeadFn (default) against featurizers._NativeHeadFunction (cfg.native_autograd = False): same C ABI calls,
    so feats, code and the six parameter gradients are bitwise equal - training mode (masks, saved H) and eval mode."""
    from stego_amd import capi
    assert capi.torchglue() is not None and hasattr(capi.torchglue(), "head")
    net, cfg = _featurizer("vit_small", 70, "nonlinear", True)
    g = torch.Generator(device=DEV).manual_seed(9)
    tokens = torch.randn(6, 1 + 14 * 14, 384, device=DEV, generator=g)
    image_feat = tokens[:, 1:, :].reshape(6, 14, 14, 384).permute(0, 3, 1, 2)
    up = torch.randn(6, 70, 14, 14, device=DEV, generator=g) / 196
    params = [p for n, p in net.named_parameters() if n.startswith("cluster")]
    res = {}
    for native_autograd in (True, False):
        cfg.native_autograd = native_autograd
        rec = []
        for train in (True, False):
            net.train(train)
            for p in params:
                p.grad = None
            torch.manual_seed(5)
            feats, code = net._head_native(image_feat)
            (code * up).sum().backward()
            rec += [feats.detach().clone(), code.detach().clone()] + [p.grad.detach().clone() for p in params]
        res[native_autograd] = rec
    for a, b in zip(res[True], res[False]):
        assert torch.equal(a, b)
