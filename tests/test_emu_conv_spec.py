"""conv_spec_kernel (csrc/conv_spec.h: the forward convolution on conv_pipe_kernel's operand ring with producer / consumer wave roles, the
finished tile staged in dead LDS and stored - with its fused BatchNorm sums - by the producer waves over the next tile's first K steps)
against conv_pipe_kernel on the same operands: outputs BIT-identical in the same form (ring vs ring, HALO vs
HALO: same K order, same rounding points), fused sums equal up to fp32 summation order.  LP_CONV_SPEC selects the kernel
(lp_config_reload_env, tests/conftest.py); LP_CONV_MAX_WGS forces long persistent walks: the staged tile of one tile is drained under the
next one's K loop, the per-thread sums flush when the column block or the BatchNorm segment changes, the last tile drains after the walk."""

import numpy as np
import pytest
import torch

from tests.hipemu import emu

pytestmark = pytest.mark.usefixtures("kernel_backend")

SPEC, SPEC_HALO, PIPE, PIPE_HALO = 6, 7, 1, 4   # lp_hip.h: LP_CONV_KERNEL_*


def _both(monkeypatch, fn, want_kernel):
    monkeypatch.setenv("LP_CONV_SPEC", "0")
    ref = fn()
    assert emu.lib().lp_conv_last_kernel() in (PIPE, PIPE_HALO)
    monkeypatch.setenv("LP_CONV_SPEC", "1")
    out = fn()
    assert emu.lib().lp_conv_last_kernel() == want_kernel
    return ref, out


FWD_CASES = [
    # B, Hi, Wi, Ci, Co, R, stride, pad
    (4, 16, 16, 256, 128, 1, 1, 0),    # 1x1, K = 256: exactly the 4 K steps the handed-over store pass needs; 4 tiles
    (3, 16, 16, 64, 128, 3, 1, 1),     # 3x3 "same", 9 K steps, 3 tiles
    (1, 19, 15, 128, 128, 3, 1, 1),    # M = 285: a ragged second tile (rows past M are staged but not stored)
    (2, 18, 18, 64, 256, 3, 2, 1),     # 3x3 stride 2, two column tiles: the column block changes inside a workgroup's walk (sums flush)
    (2, 8, 8, 320, 384, 1, 1, 0),      # M = 128 < one tile, three column tiles, 5 K steps
]


@pytest.mark.parametrize("wgs", ["0", "1", "2"])
@pytest.mark.parametrize("case", FWD_CASES)
def test_spec_forward_equals_pipe(case, wgs, monkeypatch):
    monkeypatch.setenv("LP_CONV_HALO", "0")   # the per-tap ring form (the HALO form: below)
    if wgs != "0":
        monkeypatch.setenv("LP_CONV_MAX_WGS", wgs)
    B, Hi, Wi, Ci, Co, R, st, pad = case
    gen = torch.Generator().manual_seed(17 + sum(case))
    g = emu.geom(B, Hi, Wi, Ci, Co, R, R, st, pad)
    x = emu.to_bf16_bits(torch.randn(B, Hi, Wi, Ci, generator=gen))
    wg = emu.to_bf16_bits(torch.randn(Co, R, R, Ci, generator=gen) / (Ci * R * R) ** 0.5)
    (z0, _), (z1, _) = _both(monkeypatch, lambda: emu.conv_fwd(x, wg, g), SPEC)
    assert np.array_equal(z0, z1)
    (zb0, s0), (zb1, s1) = _both(monkeypatch, lambda: emu.conv_fwd_bn(x, wg, g), SPEC)
    assert np.array_equal(zb0, z0) and np.array_equal(zb1, z0)
    np.testing.assert_allclose(s1, s0, rtol=2e-5, atol=2e-4)


HALO_CASES = [
    # B, H, W, Ci, Co   (3x3, stride 1, pad 1)
    (3, 16, 16, 64, 128),      # one channel slice per tile (9 K steps): the two halo images alternate per TILE, each staged over in turn
    (2, 16, 16, 128, 128),     # two slices
    (1, 19, 15, 128, 256),     # ragged M, two column tiles
    (2, 24, 24, 64, 128),      # ResNet's layer3 grid: a tile spans 10.7 image rows
]


@pytest.mark.parametrize("wgs", ["0", "1", "3"])
@pytest.mark.parametrize("case", HALO_CASES)
def test_spec_halo_form_equals_pipe_halo_form(case, wgs, monkeypatch):
    if wgs != "0":
        monkeypatch.setenv("LP_CONV_MAX_WGS", wgs)
    B, Hi, Wi, Ci, Co = case
    gen = torch.Generator().manual_seed(41 + sum(case))
    g = emu.geom(B, Hi, Wi, Ci, Co, 3, 3, 1, 1)
    x = emu.to_bf16_bits(torch.randn(B, Hi, Wi, Ci, generator=gen))
    wg = emu.to_bf16_bits(torch.randn(Co, 3, 3, Ci, generator=gen) / (Ci * 9) ** 0.5)
    (z0, _), (z1, _) = _both(monkeypatch, lambda: emu.conv_fwd(x, wg, g), SPEC_HALO)
    assert np.array_equal(z0, z1)
    (zb0, s0), (zb1, s1) = _both(monkeypatch, lambda: emu.conv_fwd_bn(x, wg, g), SPEC_HALO)
    assert np.array_equal(zb1, z0)
    np.testing.assert_allclose(s1, s0, rtol=2e-5, atol=2e-4)


def test_spec_two_batchnorm_segments(monkeypatch):
    """Joint labeled + unlabeled pass: the segment boundary sits on a 256-row tile; a workgroup that walks across it flushes its sums"""
    monkeypatch.setenv("LP_CONV_MAX_WGS", "1")
    gen = torch.Generator().manual_seed(6)
    B, H, Ci, Co, seg = 3, 16, 256, 128, 1      # 256 rows per image
    g = emu.geom(B, H, H, Ci, Co, 1, 1, 1, 0)
    x = emu.to_bf16_bits(torch.randn(B, H, H, Ci, generator=gen))
    w = emu.to_bf16_bits(torch.randn(Co, 1, 1, Ci, generator=gen) / 16)
    (z0, s0), (z1, s1) = _both(monkeypatch, lambda: emu.conv_fwd_bn(x, w, g, seg=seg), SPEC)
    assert np.array_equal(z0, z1) and s0.shape == (2, 2, Co)
    np.testing.assert_allclose(s1, s0, rtol=2e-5, atol=2e-4)
    zf = emu.from_bf16_bits(z1).double().reshape(B, -1, Co)
    np.testing.assert_allclose(s1[0, 0], zf[:seg].sum((0, 1)).numpy(), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(s1[1, 1], (zf[seg:] ** 2).sum((0, 1)).numpy(), rtol=1e-5, atol=1e-4)


def test_short_tiles_stay_on_conv_pipe_kernel(monkeypatch):
    """fewer K steps per tile than the handed-over store pass needs (4), 64-channel column blocks, a bias: conv_pipe_kernel"""
    monkeypatch.setenv("LP_CONV_SPEC", "1")
    gen = torch.Generator().manual_seed(2)
    for (Ci, Co) in ((128, 128), (256, 64)):
        g = emu.geom(2, 16, 16, Ci, Co, 1, 1, 1, 0)
        x = emu.to_bf16_bits(torch.randn(2, 16, 16, Ci, generator=gen))
        w = emu.to_bf16_bits(torch.randn(Co, 1, 1, Ci, generator=gen) / 16)
        emu.conv_fwd(x, w, g)
        assert emu.lib().lp_conv_last_kernel() == PIPE
