"""Inference kernels (written after round 1's last device run, hence in a late-sorting file): lp_bn_fold + lp_conv_fwd_act (conv + folded
BatchNorm + residual + ReLU in one launch) and lp_attn_fwd with p = NULL (no T x T write)."""

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.hipemu import emu

pytestmark = pytest.mark.usefixtures("kernel_backend")

bf = lambda t: t.to(torch.bfloat16).float()  # noqa: E731


def nhwc(t):  # (B,C,H,W) -> (B,H,W,C)
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("case", [(2, 8, 8, 64, 64, 1, 1, 0), (1, 9, 7, 64, 128, 3, 1, 1), (2, 10, 10, 128, 192, 3, 2, 1), (3, 8, 8, 128, 64, 1, 2, 0)])
@pytest.mark.parametrize("residual,relu", [(False, True), (True, True), (False, False), (True, False)])
def test_conv_fwd_act_folded_batchnorm(case, residual, relu):
    """inference form of conv -> BatchNorm(running statistics) [-> + identity] [-> ReLU]: lp_bn_fold + ONE lp_conv_fwd_act launch
    vs torch's eval-mode sequence on the same bf16-rounded folded operands"""
    B, Hi, Wi, Ci, Co, R, st, pad = case
    gen = torch.Generator().manual_seed(sum(case) + 7)
    x = bf(torch.randn(B, Ci, Hi, Wi, generator=gen))
    w = torch.randn(Co, Ci, R, R, generator=gen) / (Ci * R * R) ** 0.5
    gamma, beta = 0.5 + torch.rand(Co, generator=gen), torch.randn(Co, generator=gen) * 0.3
    rmean, rvar = torch.randn(Co, generator=gen) * 0.2, 0.3 + torch.rand(Co, generator=gen)
    g = emu.geom(B, Hi, Wi, Ci, Co, R, R, st, pad)
    wf_bits, bias = emu.bn_fold(w.permute(0, 2, 3, 1).reshape(Co, -1).numpy(), gamma.numpy(), beta.numpy(), rmean.numpy(), rvar.numpy())
    a = gamma / torch.sqrt(rvar + 1e-5)
    torch.testing.assert_close(torch.from_numpy(bias), beta - rmean * a, atol=1e-6, rtol=1e-6)
    wf = emu.from_bf16_bits(wf_bits).reshape(Co, R, R, Ci)
    # <= 1 bf16 ulp: the kernel's fp32 scale factor may differ from torch's by a rounding of the divide / sqrt on some hosts
    torch.testing.assert_close(wf, bf(w.permute(0, 2, 3, 1) * a.view(-1, 1, 1, 1)), atol=0, rtol=2.0 ** -7)
    res = bf(torch.randn(B * g.Ho * g.Wo, Co, generator=gen)) if residual else None
    got = emu.from_bf16_bits(emu.conv_fwd_act(emu.to_bf16_bits(nhwc(x)), wf_bits, g, bias=bias,
                                              residual_bits=emu.to_bf16_bits(res) if residual else None, relu=relu))
    want = nhwc(F.conv2d(x, wf.permute(0, 3, 1, 2), stride=st, padding=pad)).reshape(-1, Co) + torch.from_numpy(bias)
    if residual:
        want = want + res
    if relu:
        want = torch.relu(want)
    torch.testing.assert_close(got, bf(want), atol=2e-2, rtol=1e-2)
    # and against the un-folded definition in fp32: eval-mode BatchNorm of the convolution with the original weights
    ref = F.batch_norm(F.conv2d(x, bf(w), stride=st, padding=pad), rmean, rvar, gamma, beta, training=False, eps=1e-5)
    ref = nhwc(ref).reshape(-1, Co) + (res if residual else 0)
    ref = torch.relu(ref) if relu else ref
    assert float((got - ref).abs().max()) < 0.06 * float(ref.abs().max())


@pytest.mark.parametrize("case,kernel", [((2, 16, 16, 64, 128, 1, 1, 0), 1), ((1, 20, 12, 128, 128, 3, 1, 1), 4), ((2, 17, 16, 128, 256, 1, 2, 0), 1),
                                         ((2, 12, 12, 64, 64, 3, 2, 1), 1), ((1, 9, 9, 64, 192, 1, 1, 0), 0)])
def test_conv_fwd_act_on_the_pipelined_kernel(case, kernel, monkeypatch):
    """round 3: lp_conv_fwd_act takes conv_pipe_kernel (plain / HALO form) where the shape allows, the residual and the ReLU applied in its store
    pass to the bf16 value of accumulator + bias - at most one bf16 rounding away from conv_igemm_kernel<infer> (LP_INFER_PIPE=0), which adds
    in fp32 before its only rounding"""
    B, Hi, Wi, Ci, Co, R, st, pad = case
    gen = torch.Generator().manual_seed(sum(case) + 11)
    x = emu.to_bf16_bits(nhwc(torch.randn(B, Ci, Hi, Wi, generator=gen)))
    w = emu.to_bf16_bits(torch.randn(Co, R * R * Ci, generator=gen) / (Ci * R * R) ** 0.5)
    bias = torch.randn(Co, generator=gen).numpy()
    g = emu.geom(B, Hi, Wi, Ci, Co, R, R, st, pad)
    res = emu.to_bf16_bits(torch.randn(B * g.Ho * g.Wo, Co, generator=gen))
    for residual, relu in ((True, True), (False, True), (True, False)):
        got = emu.from_bf16_bits(emu.conv_fwd_act(x, w, g, bias=bias, residual_bits=res if residual else None, relu=relu))
        assert emu.lib().lp_conv_last_kernel() == kernel
        monkeypatch.setenv("LP_INFER_PIPE", "0")
        want = emu.from_bf16_bits(emu.conv_fwd_act(x, w, g, bias=bias, residual_bits=res if residual else None, relu=relu))
        assert emu.lib().lp_conv_last_kernel() == 0
        monkeypatch.delenv("LP_INFER_PIPE")
        if not residual and Ci * R * R <= 64:
            assert torch.equal(got, want)        # same K order, no second rounding
        # two roundings of values up to |a| + |r|: 2^-8 relative of the larger of the sum and its terms
        scale = torch.maximum(want.abs(), emu.from_bf16_bits(res).abs() if residual else want.abs())
        assert float(((got - want).abs() / (scale + 1e-3)).max()) <= 2.0 ** -7


def test_conv_fwd_act_of_layer1s_3x3_layers_on_the_resident_filter_kernel(monkeypatch):
    """round 4: the 64 -> 64 channel 3x3 layers (conv2 of layer1's blocks: no residual branch) take conv_res2d_kernel's inference store pass -
    bias and ReLU on the accumulators, one rounding: bit-identical to conv_igemm_kernel<infer>; with a residual the launch stays on
    conv_pipe_kernel"""
    B, Hi, Wi = 2, 32, 16
    gen = torch.Generator().manual_seed(41)
    x = emu.to_bf16_bits(nhwc(torch.randn(B, 64, Hi, Wi, generator=gen)))
    w = emu.to_bf16_bits(torch.randn(64, 9 * 64, generator=gen) / 24)
    bias = torch.randn(64, generator=gen).numpy()
    g = emu.geom(B, Hi, Wi, 64, 64, 3, 3, 1, 1)
    for relu in (True, False):
        got = emu.conv_fwd_act(x, w, g, bias=bias, relu=relu)
        assert emu.lib().lp_conv_last_kernel() == 5          # LP_CONV_KERNEL_RES2D
        monkeypatch.setenv("LP_INFER_PIPE", "0")
        want = emu.conv_fwd_act(x, w, g, bias=bias, relu=relu)
        assert emu.lib().lp_conv_last_kernel() == 0
        monkeypatch.delenv("LP_INFER_PIPE")
        np.testing.assert_array_equal(got, want)
        if relu:
            assert float(emu.from_bf16_bits(got).min()) == 0.0
    res = emu.to_bf16_bits(torch.randn(B * Hi * Wi, 64, generator=gen))
    emu.conv_fwd_act(x, w, g, bias=bias, residual_bits=res, relu=True)
    assert emu.lib().lp_conv_last_kernel() in (1, 4)       # conv_pipe_kernel, plain or HALO form


@pytest.mark.parametrize("nb,nh,T,ldp", [(2, 2, 77, 80), (1, 3, 130, 192), (1, 1, 64, 64), (1, 2, 5, 8)])
def test_attention_forward_without_probabilities(nb, nh, T, ldp):
    """inference form (p = NULL): the same O as the training form bit for bit, nothing of size T x T written"""
    gen = torch.Generator().manual_seed(nb * 100 + T)
    d, scale = 64, 0.125
    D = nh * d
    ld = 3 * D + 8
    qkv = emu.to_bf16_bits(torch.randn(nb * T, ld, generator=gen) * 1.5)
    _p, obits = emu.attn_fwd(qkv.reshape(-1), ld, D, 2 * D, nb, nh, T, scale, ldp, D)
    none, obits_inf = emu.attn_fwd(qkv.reshape(-1), ld, D, 2 * D, nb, nh, T, scale, ldp, D, write_p=False)
    assert none is None
    np.testing.assert_array_equal(obits_inf, obits)
