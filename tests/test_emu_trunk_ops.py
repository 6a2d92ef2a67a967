"""Kernel-logic tests of csrc/bn.hip and csrc/optim.hip on the CPU emulator vs torch fp32 references on the same
bf16-rounded inputs."""

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.hipemu import emu

pytestmark = pytest.mark.usefixtures("kernel_backend")

bf = lambda t: t.to(torch.bfloat16).float()  # noqa: E731


@pytest.mark.parametrize("M,C,relu,res", [(300, 64, True, False), (70, 256, True, True), (513, 24, False, False)])
def test_batchnorm_fwd_bwd(M, C, relu, res):
    gen = torch.Generator().manual_seed(M + C)
    x = bf(torch.randn(M, C, generator=gen) * 2 + 0.5).requires_grad_(True)
    gamma = (torch.rand(C, generator=gen) + 0.5).requires_grad_(True)
    beta = torch.randn(C, generator=gen).requires_grad_(True)
    r = bf(torch.randn(M, C, generator=gen)).requires_grad_(True) if res else None
    rm, rv = torch.zeros(C), torch.ones(C)
    # torch reference: BN over rows (N=M, C) in training mode
    yt = F.batch_norm(x, rm, rv, gamma, beta, training=True, momentum=0.1, eps=1e-5)
    if res:
        yt = yt + r
    if relu:
        yt = F.relu(yt)
    run = (np.zeros(C, np.float32), np.ones(C, np.float32))
    y, mean, invstd = emu.bn_forward(emu.to_bf16_bits(x.detach()), M, C, gamma.detach().numpy(), beta.detach().numpy(),
                                     emu.to_bf16_bits(r.detach()) if res else None, relu, running=run)
    torch.testing.assert_close(emu.from_bf16_bits(y), bf(yt.detach()), atol=2e-2, rtol=2e-2)
    np.testing.assert_allclose(run[0], rm.numpy(), atol=1e-5, rtol=1e-4)
    np.testing.assert_allclose(run[1], rv.numpy(), atol=1e-5, rtol=1e-4)
    # backward: use the kernel's own bf16 output as the relu mask, compare with autograd on the fp32 graph
    dy = bf(torch.randn(M, C, generator=gen))
    yt.backward(dy)
    dx, dres, dgamma, dbeta = emu.bn_backward(emu.to_bf16_bits(dy), y if relu else None, emu.to_bf16_bits(x.detach()), mean, invstd,
                                              gamma.detach().numpy(), M, C, want_dres=res)
    mask_mismatch = ((emu.from_bf16_bits(y) > 0) != (yt.detach() > 0)).float().mean() if relu else 0.0
    assert mask_mismatch < 0.01
    tol = dict(atol=3e-2, rtol=3e-2)
    torch.testing.assert_close(emu.from_bf16_bits(dx), x.grad, **tol)
    torch.testing.assert_close(torch.from_numpy(dgamma), gamma.grad, atol=5e-2, rtol=2e-2)
    torch.testing.assert_close(torch.from_numpy(dbeta), beta.grad, atol=5e-2, rtol=2e-2)
    if res:
        torch.testing.assert_close(emu.from_bf16_bits(dres), r.grad, **tol)


def test_maxpool_fwd_bwd_with_ties():
    gen = torch.Generator().manual_seed(1)
    B, H, W, C = 2, 9, 10, 16
    x = bf(F.relu(torch.randn(B, C, H, W, generator=gen)))  # many exact zeros -> ties
    x.requires_grad_(True)
    y = F.max_pool2d(x, 3, 2, 1)
    dy = bf(torch.randn(y.shape, generator=gen))
    y.backward(dy)
    xb = emu.to_bf16_bits(x.detach().permute(0, 2, 3, 1))
    yb, arg = emu.maxpool(xb, B, H, W, C)
    torch.testing.assert_close(emu.from_bf16_bits(yb), y.detach().permute(0, 2, 3, 1))
    dx = emu.maxpool_bwd(arg, emu.to_bf16_bits(dy.permute(0, 2, 3, 1)), B, H, W, C)
    torch.testing.assert_close(emu.from_bf16_bits(dx), bf(x.grad.permute(0, 2, 3, 1)), atol=2e-2, rtol=2e-2)


def test_stem_bn_relu_maxpool_fused_equals_the_unfused_chain():
    """lp_bn_relu_maxpool_fwd == lp_bn_apply(relu) -> lp_maxpool_fwd bit for bit (values AND arg-max bytes, ties included), and the
    fused backward (gradient of the never-stored activation rebuilt from pooled gradient + arg-max + z) == lp_maxpool_bwd ->
    lp_bn_bwd_reduce / lp_bn_bwd_apply, and both == autograd of BatchNorm -> ReLU -> max_pool2d."""
    gen = torch.Generator().manual_seed(4)
    B, H, W, C = 2, 10, 12, 64
    M = B * H * W
    z = bf(torch.randn(B, C, H, W, generator=gen) * 2 + 0.3).requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(C, generator=gen)).requires_grad_(True)
    beta = (0.1 * torch.randn(C, generator=gen) - 0.4).requires_grad_(True)  # plenty of negative pre-activations -> zeros -> ties
    zb = emu.to_bf16_bits(z.detach().permute(0, 2, 3, 1)).reshape(M, C)
    a_bits, mean, invstd = emu.bn_forward(zb, M, C, gamma.detach().numpy(), beta.detach().numpy(), relu=True)
    y0, arg0 = emu.maxpool(a_bits, B, H, W, C)
    y1, arg1 = emu.bn_relu_maxpool(zb, mean, invstd, gamma.detach().numpy(), beta.detach().numpy(), B, H, W, C)
    assert np.array_equal(y0, y1) and np.array_equal(arg0, arg1)
    # backward
    y = F.max_pool2d(F.relu(F.batch_norm(z, None, None, gamma, beta, training=True, eps=1e-5)), 3, 2, 1)
    dy = bf(torch.randn(y.shape, generator=gen))
    y.backward(dy)
    dyb = emu.to_bf16_bits(dy.permute(0, 2, 3, 1))
    da = emu.maxpool_bwd(arg0, dyb, B, H, W, C).reshape(M, C)
    dz0, _, dgamma0, dbeta0 = emu.bn_backward(da, a_bits, zb, mean, invstd, gamma.detach().numpy(), M, C)
    dz1, dgamma1, dbeta1, _ = emu.bn_pool_backward(arg1, dyb, zb, mean, invstd, gamma.detach().numpy(), beta.detach().numpy(), B, H, W, C)
    # the unfused chain rounds the gathered gradient to bf16 before reducing it; the fused one keeps it in fp32
    torch.testing.assert_close(torch.from_numpy(dgamma1), torch.from_numpy(dgamma0), atol=3e-2, rtol=5e-3)
    torch.testing.assert_close(torch.from_numpy(dbeta1), torch.from_numpy(dbeta0), atol=3e-2, rtol=5e-3)
    torch.testing.assert_close(emu.from_bf16_bits(dz1), emu.from_bf16_bits(dz0), atol=2e-2, rtol=2e-2)
    torch.testing.assert_close(torch.from_numpy(dgamma1), gamma.grad, atol=5e-2, rtol=1e-2)
    torch.testing.assert_close(torch.from_numpy(dbeta1), beta.grad, atol=5e-2, rtol=1e-2)
    torch.testing.assert_close(emu.from_bf16_bits(dz1).reshape(B, H, W, C), z.grad.permute(0, 2, 3, 1), atol=2e-2, rtol=2e-2)


def test_images_and_pixel_shuffle():
    gen = torch.Generator().manual_seed(2)
    img = torch.randn(2, 3, 6, 5, generator=gen)
    out = emu.from_bf16_bits(emu.images_to_nhwc4(img.numpy()))
    torch.testing.assert_close(out[..., :3], bf(img.permute(0, 2, 3, 1)))
    assert not out[..., 3].any()
    x = bf(torch.randn(2, 32, 3, 4, generator=gen))  # (B, 4*c_out, h, w)
    want = F.pixel_shuffle(x, 2).permute(0, 2, 3, 1)
    got = emu.pixel_shuffle(emu.to_bf16_bits(x.permute(0, 2, 3, 1)), 2, 3, 4, 8)
    torch.testing.assert_close(emu.from_bf16_bits(got), want)
    back = emu.pixel_shuffle(got, 2, 3, 4, 8, inverse=True)
    torch.testing.assert_close(emu.from_bf16_bits(back), x.permute(0, 2, 3, 1))
    # padded channel pitch (ViT head: 96 channels stored in 128): pad channels untouched (zero here), inverse ignores them
    got16 = emu.pixel_shuffle(emu.to_bf16_bits(x.permute(0, 2, 3, 1)), 2, 3, 4, 8, ld=16)
    torch.testing.assert_close(emu.from_bf16_bits(got16)[..., :8], want)
    assert not got16[..., 8:].any()
    back = emu.pixel_shuffle(got16, 2, 3, 4, 8, inverse=True, ld=16)
    torch.testing.assert_close(emu.from_bf16_bits(back), x.permute(0, 2, 3, 1))


@pytest.mark.parametrize("opt", ["Adam", "AdamW"])
def test_adam_matches_torch(opt):
    gen = torch.Generator().manual_seed(3)
    p0 = torch.randn(1000, generator=gen)
    p = torch.nn.Parameter(p0.clone())
    o = (torch.optim.Adam if opt == "Adam" else torch.optim.AdamW)([p], lr=1e-3)
    pm, m, v = p0.numpy().copy(), np.zeros(1000, np.float32), np.zeros(1000, np.float32)
    for step in range(1, 4):
        g = torch.randn(1000, generator=gen)
        p.grad = g.clone()
        o.step()
        pm, m, v, pb = emu.adam(pm, g.numpy(), m, v, 1e-3, step, wd=0.01 if opt == "AdamW" else 0.0, decoupled=opt == "AdamW")
        np.testing.assert_allclose(pm, p.detach().numpy(), atol=1e-6, rtol=1e-5)
        torch.testing.assert_close(emu.from_bf16_bits(pb), bf(torch.from_numpy(pm)))
    # lr = 0 keeps parameters but still moves the moments (frozen backbone semantics)
    p_before = pm.copy()
    pm2, m2, v2, _ = emu.adam(pm, np.ones(1000, np.float32), m, v, 0.0, 4)
    np.testing.assert_array_equal(pm2, p_before)
    assert np.abs(m2 - m).max() > 0


def test_permute_cba():
    a = (np.arange(4 * 9 * 8).reshape(4, 9, 8) % 65536).astype(np.uint16)
    np.testing.assert_array_equal(emu.permute_cba(a, 4, 9, 8), a.transpose(2, 1, 0))
