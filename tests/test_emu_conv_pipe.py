"""The pipelined convolution kernel (csrc/conv_pipe.h: direct-to-LDS operand ring, 256-row tiles, swapped-role MFMA store pass) against
conv_igemm_kernel on the same operands: outputs BIT-identical (same K order, same rounding points), fused BatchNorm sums equal up to
fp32 summation order.  LP_CONV_PIPE is read per call, so both kernels run in one process; LP_CONV_MAX_WGS forces long persistent walks
(the loader crosses tile boundaries two K steps ahead of the MFMAs; the per-thread BatchNorm sums flush when the column block changes)."""

import numpy as np
import pytest
import torch

from tests.hipemu import emu

pytestmark = pytest.mark.usefixtures("kernel_backend")

CASES = [
    # B, Hi, Wi, Ci, Co, R, stride, pad
    (2, 8, 8, 64, 64, 1, 1, 0),       # 1x1, one K step per tile (the loader is two TILES ahead), M = 128 < one tile
    (3, 16, 16, 64, 128, 3, 1, 1),    # 3x3 "same", M = 768 = 3 tiles, BN = 128
    (1, 19, 15, 128, 64, 3, 1, 1),    # M = 285: a ragged second tile; two K steps per tap; BN = 64
    (2, 18, 18, 64, 256, 3, 2, 1),    # 3x3 stride 2 (4 parity-class launches in the data gradient), two column tiles
    (4, 16, 16, 256, 64, 1, 1, 0),    # 1x1 reduce (K = 256), 4 tiles
    (5, 8, 8, 128, 128, 1, 2, 0),     # 1x1 stride 2 (downsample): data gradient has three empty parity classes (old kernel) + one pipelined
]


def _both(monkeypatch, fn):
    monkeypatch.setenv("LP_CONV_PIPE", "0")
    ref = fn()
    monkeypatch.setenv("LP_CONV_PIPE", "1")
    return ref, fn()


@pytest.mark.parametrize("wgs", ["0", "1", "2"])
@pytest.mark.parametrize("case", CASES)
def test_pipe_equals_igemm(case, wgs, monkeypatch):
    monkeypatch.setenv("LP_CONV_HALO", "0")   # the per-tap ring (the HALO form of the 3x3 layers has its own test below)
    if wgs != "0":
        monkeypatch.setenv("LP_CONV_MAX_WGS", wgs)
    B, Hi, Wi, Ci, Co, R, st, pad = case
    gen = torch.Generator().manual_seed(11 + sum(case))
    g = emu.geom(B, Hi, Wi, Ci, Co, R, R, st, pad)
    x = emu.to_bf16_bits(torch.randn(B, Hi, Wi, Ci, generator=gen))
    w = torch.randn(Co, R, R, Ci, generator=gen) / (Ci * R * R) ** 0.5
    wg, wd = emu.to_bf16_bits(w), emu.to_bf16_bits(w.permute(3, 1, 2, 0))
    # forward, plain and with the fused BatchNorm sums
    (z0, _), (z1, _) = _both(monkeypatch, lambda: emu.conv_fwd(x, wg, g))
    assert np.array_equal(z0, z1) and emu.lib().lp_conv_last_kernel() == 1   # LP_CONV_KERNEL_PIPE
    (zb0, s0), (zb1, s1) = _both(monkeypatch, lambda: emu.conv_fwd_bn(x, wg, g))
    assert np.array_equal(zb0, z0) and np.array_equal(zb1, z0)
    np.testing.assert_allclose(s1, s0, rtol=2e-5, atol=2e-4)
    # data gradient: addend + each ReLU-mask source, plain and with the BatchNorm-backward sums
    Mi = B * Hi * Wi
    zin_bits = emu.to_bf16_bits(torch.randn(Mi, Ci, generator=gen))
    gamma, beta = torch.rand(Ci, generator=gen) + 0.5, torch.randn(Ci, generator=gen) * 0.3
    a_bits, mean, invstd, relu_bits = emu.bn_forward(zin_bits, Mi, Ci, gamma.numpy(), beta.numpy(), relu=True, want_bits=True)
    dy = emu.to_bf16_bits(torch.randn(B * g.Ho * g.Wo, Co, generator=gen))
    add = emu.to_bf16_bits(torch.randn(Mi, Ci, generator=gen))
    (d0, _), (d1, _) = _both(monkeypatch, lambda: emu.conv_dgrad(dy, wd, g, addend_bits=add, mask_bits=a_bits))
    assert np.array_equal(d0, d1)
    (e0, _), (e1, _) = _both(monkeypatch, lambda: emu.conv_dgrad(dy, wd, g))
    assert np.array_equal(e0, e1)
    # ... and the mask at 1 bit per element (lp_conv_dgrad_bits: kEkPB on conv_pipe_kernel, the relu_bits branch of conv_igemm_kernel): the
    # same gradient as with the bf16 activation as mask; then the projection shortcut's form - accumulated in place into an existing
    # gradient, only the pixels a filter tap reaches touched
    b0, b1 = _both(monkeypatch, lambda: emu.conv_dgrad_bits(dy, wd, g, relu_bits, addend_bits=add))
    assert np.array_equal(b0, d0) and np.array_equal(b1, d0)
    assert st != 1 or emu.lib().lp_conv_last_kernel() == 1   # LP_CONV_KERNEL_PIPE (a strided launch ends on its last parity class's kernel)
    (m0, _), _ = _both(monkeypatch, lambda: emu.conv_dgrad(dy, wd, g, mask_bits=a_bits))
    n0, n1 = _both(monkeypatch, lambda: emu.conv_dgrad_bits(dy, wd, g, relu_bits))
    assert np.array_equal(n0, m0) and np.array_equal(n1, m0)
    acc0, acc1 = _both(monkeypatch, lambda: emu.conv_dgrad_bits(dy, wd, g, relu_bits, into=d0, skip=1))
    (ref0, _), (ref1, _) = _both(monkeypatch, lambda: emu.conv_dgrad(dy, wd, g, mask_bits=a_bits, into=d0))
    assert np.array_equal(ref0, ref1) and np.array_equal(acc0, ref0) and np.array_equal(acc1, ref0)
    assert not np.array_equal(ref0, d0)   # (something was accumulated)
    for mask, bits in ((a_bits, None), (None, None), (None, relu_bits)):
        r0, r1 = _both(monkeypatch, lambda: emu.conv_dgrad_bn(dy, wd, g, zin_bits, mean, invstd, gamma.numpy(), beta.numpy(), addend_bits=add,
                                                           mask_bits=mask, relu_bits=bits))
        assert np.array_equal(r0[0], r1[0]) and np.array_equal(r0[0], d0)
        for a, b in zip(r0[1:], r1[1:]):
            np.testing.assert_allclose(b, a, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("B,H,W,Ci,Co,seg", [(2, 16, 16, 128, 64, 0), (3, 16, 16, 256, 128, 1), (1, 15, 17, 128, 64, 0)])
def test_dgrad_takes_the_shortcut_gradient_from_its_own_grid(B, H, W, Ci, Co, seg, monkeypatch):
    """lp_bn_fuse.addend_half: the addend of a 1x1 data gradient with fused BatchNorm sums (store-pass kind kEkAZB) is a tensor on the
    half-resolution grid - the gradient of the block's stride-2 projection shortcut - added at the even pixels only.  Equal (as numbers) to the
    same launch with that tensor scattered into a dense addend of zeros, sums included; odd sizes (the shortcut's grid is ceil(H / 2) x
    ceil(W / 2)); two BatchNorm segments; both kernels."""
    gen = torch.Generator().manual_seed(B + H + Ci)
    g = emu.geom(B, H, W, Ci, Co, 1, 1, 1, 0)
    M = B * H * W
    Hh, Wh = (H + 1) // 2, (W + 1) // 2
    dy = emu.to_bf16_bits(torch.randn(M, Co, generator=gen))
    wd = emu.to_bf16_bits((torch.randn(Co, 1, 1, Ci, generator=gen) / Co ** 0.5).permute(3, 1, 2, 0))
    half = emu.from_bf16_bits(emu.to_bf16_bits(torch.randn(B, Hh, Wh, Ci, generator=gen))).reshape(B, Hh, Wh, Ci)
    dense = torch.zeros(B, H, W, Ci)
    dense[:, ::2, ::2] = half
    zin = emu.to_bf16_bits(torch.randn(M, Ci, generator=gen))
    gamma, beta = torch.rand(Ci, generator=gen) + 0.5, torch.randn(Ci, generator=gen) * 0.3
    if seg:
        if (seg * H * W) % 256:
            pytest.skip("segment boundary off the tile grid")
        mean, invstd = np.random.default_rng(1).normal(size=(2, Ci)).astype(np.float32), (np.random.default_rng(2).random((2, Ci)) + 0.5).astype(np.float32)
        _, _, _, relu_bits = emu.bn_forward(zin, M, Ci, gamma.numpy(), beta.numpy(), relu=True, want_bits=True)
    else:
        _, mean, invstd, relu_bits = emu.bn_forward(zin, M, Ci, gamma.numpy(), beta.numpy(), relu=True, want_bits=True)
    monkeypatch.setenv("LP_CONV_PIPE", "1")
    ref = emu.conv_dgrad_bn(dy, wd, g, zin, mean, invstd, addend_bits=emu.to_bf16_bits(dense.reshape(M, Ci)), relu_bits=relu_bits, seg=seg)
    got = emu.conv_dgrad_bn(dy, wd, g, zin, mean, invstd, addend_bits=emu.to_bf16_bits(half.reshape(-1, Ci)), relu_bits=relu_bits, seg=seg,
                            addend_half=True)
    assert emu.lib().lp_conv_last_kernel() == 1
    assert torch.equal(emu.from_bf16_bits(got[0]), emu.from_bf16_bits(ref[0]))
    np.testing.assert_allclose(got[1], ref[1], rtol=1e-6, atol=1e-6)
    assert emu.from_bf16_bits(got[0]).abs().max() > 0
    monkeypatch.setenv("LP_CONV_PIPE", "0")   # the register-staged kernel (what shapes conv_pipe_kernel declines run on)
    old = emu.conv_dgrad_bn(dy, wd, g, zin, mean, invstd, addend_bits=emu.to_bf16_bits(half.reshape(-1, Ci)), relu_bits=relu_bits, seg=seg,
                            addend_half=True)
    assert emu.lib().lp_conv_last_kernel() == 0
    assert torch.equal(emu.from_bf16_bits(old[0]), emu.from_bf16_bits(ref[0]))
    np.testing.assert_allclose(old[1], ref[1], rtol=1e-4, atol=1e-3)


def test_dgrad_bits_refuses_what_it_cannot_do():
    """lp_conv_dgrad_bits: a null mask is an argument error (lp_conv_dgrad is the call without a mask); the bits are indexed per 8 output
    channels, so a channel count that is not a multiple of 8 is refused before anything is enqueued"""
    import ctypes as C
    g = emu.geom(1, 8, 8, 64, 64, 1, 1, 1, 0)
    dy, wd = emu.Buf(emu.to_bf16_bits(torch.randn(64, 64))), emu.Buf(emu.to_bf16_bits(torch.randn(64, 64)))
    out, bits = emu.Z((64, 64), np.uint16), emu.Z((64, 8), np.uint8)
    lib = emu.lib()
    assert lib.lp_conv_dgrad_bits(dy.p, wd.p, C.byref(g), None, None, out.p, 0, emu.stream()) == -1     # LP_ERR_ARGUMENT
    assert lib.lp_conv_dgrad_bits(dy.p, wd.p, C.byref(g), None, bits.p, None, 0, emu.stream()) == -1
    g4 = emu.geom(1, 8, 8, 4, 64, 1, 1, 1, 0)                                                                # 4 "input" channels
    assert lib.lp_conv_dgrad_bits(dy.p, wd.p, C.byref(g4), None, bits.p, out.p, 0, emu.stream()) == -2  # LP_ERR_UNSUPPORTED


def test_pipe_two_batchnorm_segments(monkeypatch):
    """Joint labeled + unlabeled pass: images [0, seg) and [seg, B) keep their own sums; the boundary sits on a 256-row tile."""
    gen = torch.Generator().manual_seed(5)
    B, H, Ci, Co, seg = 3, 16, 64, 128, 1      # 256 rows per image
    g = emu.geom(B, H, H, Ci, Co, 1, 1, 1, 0)
    x = emu.to_bf16_bits(torch.randn(B, H, H, Ci, generator=gen))
    w = emu.to_bf16_bits(torch.randn(Co, 1, 1, Ci, generator=gen) / 8)
    (z0, s0), (z1, s1) = _both(monkeypatch, lambda: emu.conv_fwd_bn(x, w, g, seg=seg))
    assert np.array_equal(z0, z1) and s0.shape == (2, 2, Co)
    np.testing.assert_allclose(s1, s0, rtol=2e-5, atol=2e-4)
    zf = emu.from_bf16_bits(z1).double().reshape(B, -1, Co)
    np.testing.assert_allclose(s1[0, 0], zf[:seg].sum((0, 1)).numpy(), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(s1[1, 1], (zf[seg:] ** 2).sum((0, 1)).numpy(), rtol=1e-5, atol=1e-4)
    # backward: per-segment mean / invstd select by row, sums per segment
    M = B * H * H
    zin = emu.to_bf16_bits(torch.randn(M, Ci, generator=gen))
    mean, invstd = torch.randn(2, Ci, generator=gen).numpy() * 0.1, (torch.rand(2, Ci, generator=gen) + 0.5).numpy()
    gamma, beta = (torch.rand(Ci, generator=gen) + 0.5).numpy(), (torch.randn(Ci, generator=gen) * 0.3).numpy()
    dy = emu.to_bf16_bits(torch.randn(M, Co, generator=gen))
    wd = emu.to_bf16_bits(torch.randn(Ci, 1, 1, Co, generator=gen) / 11)
    r0, r1 = _both(monkeypatch, lambda: emu.conv_dgrad_bn(dy, wd, g, zin, mean, invstd, gamma, beta, seg=seg))
    assert np.array_equal(r0[0], r1[0])
    for a, b in zip(r0[1:], r1[1:]):
        np.testing.assert_allclose(b, a, rtol=1e-4, atol=1e-3)


WG_CASES = [
    (3, 16, 16, 64, 128, 3, 1, 1),    # 3x3, Ka = 576 (a ragged third a-tile), BN = 128
    (1, 19, 15, 128, 64, 3, 1, 1),    # M = 285 (a ragged last K step), Ka = 1152 (5 a-tiles), BN = 64
    (4, 18, 18, 64, 256, 3, 2, 1),    # 3x3 stride 2: gathered rows jump by two pixels
    (4, 16, 16, 64, 256, 1, 1, 0),    # 1x1 with 64 input channels: transposed (a = co, b = ci)
    (4, 16, 16, 256, 64, 1, 1, 0),    # 1x1 reduce, plain addressing, BN = 64
    (6, 16, 16, 256, 128, 1, 2, 0),   # 1x1 stride 2 (projection shortcut)
    (2, 12, 12, 128, 128, 3, 1, 1),   # 3x3 "same" with 144-pixel images: the tracked (row, column) walk crosses images within a K step
]


@pytest.mark.parametrize("split", [0, 3])
@pytest.mark.parametrize("case", WG_CASES)
def test_pipelined_weight_gradient(case, split, monkeypatch):
    """conv_wgrad_pipe_kernel (+ its reduction) vs torch's weight gradient of the same bf16-rounded operands, and vs conv_wgrad_kernel"""
    import torch.nn.functional as F

    B, Hi, Wi, Ci, Co, R, st, pad = case
    gen = torch.Generator().manual_seed(3 + sum(case))
    bf = lambda t: t.to(torch.bfloat16).float()  # noqa: E731
    x = bf(torch.randn(B, Ci, Hi, Wi, generator=gen))
    w = bf(torch.randn(Co, Ci, R, R, generator=gen) / (Ci * R * R) ** 0.5).requires_grad_(True)
    y = F.conv2d(x, w, stride=st, padding=pad)
    dy = bf(torch.randn(y.shape, generator=gen))
    y.backward(dy)
    g = emu.geom(B, Hi, Wi, Ci, Co, R, R, st, pad)
    xb, dyb = emu.to_bf16_bits(x.permute(0, 2, 3, 1).contiguous()), emu.to_bf16_bits(dy.permute(0, 2, 3, 1).contiguous())
    want = w.grad.permute(0, 2, 3, 1).reshape(Co, -1)
    monkeypatch.setenv("LP_CONV_PIPE", "0")
    old = emu.conv_wgrad(xb, dyb, g, split=split)
    monkeypatch.setenv("LP_CONV_PIPE", "1")
    monkeypatch.setenv("LP_WGRAD_PIPE", "2")      # wherever the kernel can run (the default leaves HBM-bound shapes to conv_wgrad_kernel)
    new = emu.conv_wgrad(xb, dyb, g, split=split)
    assert emu.lib().lp_conv_last_kernel() == 3   # LP_CONV_KERNEL_WGRAD_PIPE
    torch.testing.assert_close(torch.from_numpy(new), want, atol=2e-3, rtol=2e-3)
    torch.testing.assert_close(torch.from_numpy(new), torch.from_numpy(old), atol=1e-3, rtol=1e-3)


HALO_CASES = [
    # B, Hi, Wi, Ci, Co   (3x3, stride 1, pad 1)
    (3, 16, 16, 64, 64),      # one slice per tile, BN = 64, tiles = whole images (256 pixels)
    (2, 16, 24, 64, 128),     # 384-pixel images: tiles start mid-row and cross images; BN = 128
    (1, 19, 15, 128, 64),     # M = 285: a ragged second tile; two slices per tile (K order differs from conv_igemm_kernel's)
    (4, 16, 8, 256, 128),     # 128-pixel images: every tile spans two images (a zero border between them); four slices
    (2, 8, 32, 64, 256),      # wide rows, two column tiles (the halo image is staged once per column tile)
]


@pytest.mark.parametrize("wgs", ["0", "1", "3"])
@pytest.mark.parametrize("case", HALO_CASES)
def test_halo_form_equals_the_per_tap_ring(case, wgs, monkeypatch):
    """conv_pipe_kernel<..., HALO> (the tile's padded-raster neighbourhood staged once per 64-channel slice, 9 taps read from it) against the
    per-tap ring and conv_igemm_kernel: bit-identical with 64 channels (same K order), equal to fp32 reassociation otherwise; forward and
    the data gradient with the fused BatchNorm-backward sums (the only store-pass form a 3x3 layer of the trunk uses)."""
    monkeypatch.setenv("LP_CONV_RES2D", "0")   # (64 -> 64 channels on 16-aligned images would go to conv_res2d_kernel: its own test below)
    if wgs != "0":
        monkeypatch.setenv("LP_CONV_MAX_WGS", wgs)
    B, Hi, Wi, Ci, Co = case
    gen = torch.Generator().manual_seed(23 + sum(case))
    g = emu.geom(B, Hi, Wi, Ci, Co, 3, 3, 1, 1)
    x = emu.to_bf16_bits(torch.randn(B, Hi, Wi, Ci, generator=gen))
    w = torch.randn(Co, 3, 3, Ci, generator=gen) / (Ci * 9) ** 0.5
    wg, wd = emu.to_bf16_bits(w), emu.to_bf16_bits(w.permute(3, 1, 2, 0))

    def both(fn):
        monkeypatch.setenv("LP_CONV_HALO", "0")
        ref = fn()
        assert emu.lib().lp_conv_last_kernel() == 1      # LP_CONV_KERNEL_PIPE
        monkeypatch.setenv("LP_CONV_HALO", "1")
        out = fn()
        assert emu.lib().lp_conv_last_kernel() == 4      # LP_CONV_KERNEL_PIPE_HALO
        return ref, out

    def same(a, b, exact):
        if exact:
            assert np.array_equal(a, b)
        else:   # bf16 results of two fp32 summation orders: at most one unit in the last place, on a small fraction of the elements
            fa, fb = emu.from_bf16_bits(a).float(), emu.from_bf16_bits(b).float()
            torch.testing.assert_close(fb, fa, atol=2e-2, rtol=1e-2)
            assert float((fa != fb).float().mean()) < 0.2

    (z0, _), (z1, _) = both(lambda: emu.conv_fwd(x, wg, g))
    same(z0, z1, Ci == 64)
    (zb0, s0), (zb1, s1) = both(lambda: emu.conv_fwd_bn(x, wg, g))
    assert np.array_equal(zb1, z1)
    np.testing.assert_allclose(s1, s0, rtol=1e-3, atol=2e-2)
    # forward against fp32 convolution of the same bf16 operands
    import torch.nn.functional as F
    xf = emu.from_bf16_bits(x).float().reshape(B, Hi, Wi, Ci).permute(0, 3, 1, 2)
    wf = emu.from_bf16_bits(wg).float().reshape(Co, 3, 3, Ci).permute(0, 3, 1, 2)
    want = F.conv2d(xf, wf, padding=1).permute(0, 2, 3, 1).reshape(-1, Co)
    torch.testing.assert_close(emu.from_bf16_bits(z1).float().reshape(-1, Co), want, atol=2e-2, rtol=1e-2)
    # data gradient: mask recomputed from z, BatchNorm-backward sums fused
    Mi = B * Hi * Wi
    zin_bits = emu.to_bf16_bits(torch.randn(Mi, Ci, generator=gen))
    gamma, beta = torch.rand(Ci, generator=gen) + 0.5, torch.randn(Ci, generator=gen) * 0.3
    _, mean, invstd, _ = emu.bn_forward(zin_bits, Mi, Ci, gamma.numpy(), beta.numpy(), relu=True, want_bits=True)
    dy = emu.to_bf16_bits(torch.randn(Mi, Co, generator=gen))
    r0, r1 = both(lambda: emu.conv_dgrad_bn(dy, wd, g, zin_bits, mean, invstd, gamma.numpy(), beta.numpy()))
    same(r0[0], r1[0], Co == 64)
    for a, b in zip(r0[1:], r1[1:]):
        np.testing.assert_allclose(b, a, rtol=2e-3, atol=3e-2)


RES2D_CASES = [
    # B, H, W, seg   (3x3, stride 1, pad 1, 64 -> 64 channels; H and W multiples of 16)
    (3, 16, 16, 0),       # one tile per image
    (2, 32, 48, 0),       # 6 tiles per image: interior / edge / corner neighbourhoods
    (3, 16, 32, 1),       # two BatchNorm segments (image 0 | images 1 - 2)
]


@pytest.mark.parametrize("wgs", ["0", "1", "2"])
@pytest.mark.parametrize("case", RES2D_CASES)
def test_res2d_kernel_equals_igemm(case, wgs, monkeypatch):
    """conv_res2d_kernel (16 x 16 tiles, the 3 x 3 x 64 x 64 filter resident in LDS, no barrier inside a tile) against conv_igemm_kernel: same K
    order and rounding points, so outputs are BIT-identical; the fused BatchNorm sums agree to fp32 summation order.  Forward (+ sums) and the
    data gradient with the ReLU mask recomputed from z and the BatchNorm-backward sums."""
    if wgs != "0":
        monkeypatch.setenv("LP_CONV_MAX_WGS", wgs)
    B, H, W, seg = case
    Ci = Co = 64
    gen = torch.Generator().manual_seed(31 + sum(case))
    g = emu.geom(B, H, W, Ci, Co, 3, 3, 1, 1)
    x = emu.to_bf16_bits(torch.randn(B, H, W, Ci, generator=gen))
    w = torch.randn(Co, 3, 3, Ci, generator=gen) / (Ci * 9) ** 0.5
    wg, wd = emu.to_bf16_bits(w), emu.to_bf16_bits(w.permute(3, 1, 2, 0))

    def both(fn):
        monkeypatch.setenv("LP_CONV_PIPE", "0")
        ref = fn()
        monkeypatch.setenv("LP_CONV_PIPE", "1")
        out = fn()
        assert emu.lib().lp_conv_last_kernel() == 5      # LP_CONV_KERNEL_RES2D
        monkeypatch.setenv("LP_CONV_RES2D", "0")
        fn()
        assert emu.lib().lp_conv_last_kernel() == 4      # ... and without it the HALO form of conv_pipe_kernel takes the layer
        monkeypatch.delenv("LP_CONV_RES2D")
        return ref, out

    (z0, _), (z1, _) = both(lambda: emu.conv_fwd(x, wg, g))
    assert np.array_equal(z0, z1)
    (zb0, s0), (zb1, s1) = both(lambda: emu.conv_fwd_bn(x, wg, g, seg=seg))
    assert np.array_equal(zb1, z0) and s1.shape == s0.shape
    np.testing.assert_allclose(s1, s0, rtol=2e-5, atol=2e-4)
    M = B * H * W
    zin = emu.to_bf16_bits(torch.randn(M, Ci, generator=gen))
    nseg = 2 if seg else 1
    mean, invstd = torch.randn(nseg, Ci, generator=gen).numpy() * 0.1, (torch.rand(nseg, Ci, generator=gen) + 0.5).numpy()
    gamma, beta = (torch.rand(Ci, generator=gen) + 0.5).numpy(), (torch.randn(Ci, generator=gen) * 0.3).numpy()
    dy = emu.to_bf16_bits(torch.randn(M, Co, generator=gen))
    r0, r1 = both(lambda: emu.conv_dgrad_bn(dy, wd, g, zin, mean if seg else mean[0], invstd if seg else invstd[0], gamma, beta, seg=seg))
    assert np.array_equal(r0[0], r1[0])
    for a, b in zip(r0[1:], r1[1:]):
        np.testing.assert_allclose(b, a, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("wgs", ["0", "1", "3"])
@pytest.mark.parametrize("B,H,W,seg", [(2, 32, 32, 0), (3, 64, 32, 1), (1, 32, 96, 0)])
def test_stem_2d_kernel_equals_igemm(B, H, W, seg, wgs, monkeypatch):
    """conv_stem2d_kernel (16 x 16 output tiles, the [64][256] filter resident in LDS, the input neighbourhood staged by direct loads, fragments
    read from the staged image) against conv_igemm_kernel<64, stem>: same k-slices in the same order, so outputs are BIT-identical; the fused
    BatchNorm sums (atomics here, the per-tile workspace there) agree to fp32 summation order."""
    if wgs != "0":
        monkeypatch.setenv("LP_CONV_MAX_WGS", wgs)
    gen = torch.Generator().manual_seed(41 + B + H + W)
    x4 = torch.zeros(B, H, W, 4)
    x4[..., :3] = torch.randn(B, H, W, 3, generator=gen)
    wp = torch.zeros(64, 8, 8, 4)
    wp[:, :7, :7, :3] = torch.randn(64, 7, 7, 3, generator=gen) / 12
    g = emu.geom(B, H, W, 4, 64, 7, 7, 2, 3)
    xb, wb = emu.to_bf16_bits(x4), emu.to_bf16_bits(wp)
    monkeypatch.setenv("LP_STEM_2D", "0")
    ref = emu.stem_fwd(xb, wb, g)
    ref_bn, ref_s = emu.stem_fwd_bn(xb, wb, g, seg=seg)
    assert emu.lib().lp_conv_last_kernel() == 0      # LP_CONV_KERNEL_IGEMM
    monkeypatch.setenv("LP_STEM_2D", "1")
    got = emu.stem_fwd(xb, wb, g)
    assert emu.lib().lp_conv_last_kernel() == 5      # LP_CONV_KERNEL_RES2D
    got_bn, got_s = emu.stem_fwd_bn(xb, wb, g, seg=seg)
    assert np.array_equal(ref, got) and np.array_equal(ref_bn, got_bn) and np.array_equal(ref, ref_bn)
    np.testing.assert_allclose(got_s, ref_s, rtol=2e-5, atol=2e-4)
