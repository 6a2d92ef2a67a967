"""Seeded random ragged shapes through the kernels whose tiling has the most edges (fused attention forward / key-value backward, the
TN GEMM, weight + bias gradient, the fused stem BatchNorm-ReLU-max-pool): partial tiles in every dimension, idle waves, pitches wider
than the data, pad columns that must stay zero.  Each case is checked against plain torch on the same bf16-rounded operands; the
emulator runs them in the CPU suite, the device under -m gpu."""

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.hipemu import emu

pytestmark = pytest.mark.usefixtures("kernel_backend")

bf = lambda t: t.to(torch.bfloat16).float()  # noqa: E731
bits, unbits = emu.to_bf16_bits, emu.from_bf16_bits


def _cases(seed, n, draw):
    rng = np.random.default_rng(seed)
    return [draw(rng) for _ in range(n)]


ATTN = _cases(0, 10, lambda r: (int(r.integers(1, 3)), int(r.integers(1, 4)), int(r.integers(1, 300)), int(r.integers(0, 20)), int(r.integers(0, 3))))


@pytest.mark.parametrize("nb,nh,T,extra_p,extra_ld", ATTN)
def test_attention_forward_and_kv_backward_ragged(nb, nh, T, extra_p, extra_ld):
    d, scale = 64, 0.125
    D = nh * d
    ldp, ld = (T + 7) // 8 * 8 + 8 * extra_p, 3 * D + 8 * extra_ld
    gen = torch.Generator().manual_seed(T * 7 + nh)
    qkv = bf(torch.randn(nb * T, ld, generator=gen) * 1.2)
    heads = lambda t2, off: t2[:, off:off + D].reshape(nb, T, nh, d).permute(0, 2, 1, 3)  # noqa: E731
    q, k, v = heads(qkv, 0), heads(qkv, D), heads(qkv, 2 * D)
    pbits, obits = emu.attn_fwd(bits(qkv).reshape(-1), ld, D, 2 * D, nb, nh, T, scale, ldp, D)
    p = torch.softmax((q @ k.transpose(-1, -2)) * scale, -1)
    got_p = unbits(pbits).reshape(nb, nh, T, ldp)
    assert (got_p[..., T:] == 0).all()
    torch.testing.assert_close(got_p[..., :T], p, atol=4e-3, rtol=1e-2)
    torch.testing.assert_close(unbits(obits).reshape(nb, T, nh, d).permute(0, 2, 1, 3), bf(got_p[..., :T]) @ v, atol=2e-2, rtol=2e-2)
    # backward on the stored probabilities
    ps = bf(p)
    o = bf(ps @ v)
    d_o_rows = bf(torch.randn(nb * T, D, generator=gen))
    d_o = heads(d_o_rows, 0)
    drow = (d_o * o).sum(-1).permute(0, 2, 1).reshape(nb * T, nh).contiguous()
    p_pad = torch.zeros(nb, nh, T, ldp)
    p_pad[..., :T] = ps
    ds_bits, dqkv_bits = emu.attn_bwd_kv(bits(qkv).reshape(-1), ld, 2 * D, bits(d_o_rows).reshape(-1), D, bits(p_pad).reshape(-1), ldp,
                                         drow.numpy(), nb, nh, T, scale, ld, D, 2 * D)
    want_ds = scale * ps * (d_o @ v.transpose(-1, -2) - drow.reshape(nb, T, nh).permute(0, 2, 1).unsqueeze(-1))
    got_ds = unbits(ds_bits).reshape(nb, nh, T, ldp)
    assert (got_ds[..., T:] == 0).all()
    torch.testing.assert_close(got_ds[..., :T], bf(want_ds), atol=2e-3 * want_ds.abs().max().item() + 1e-6, rtol=1e-2)
    dqkv = unbits(dqkv_bits)
    assert not dqkv[:, :D].any() and not dqkv[:, 3 * D:].any()   # dQ columns and the pitch padding are not this kernel's
    want_dv, want_dk = ps.transpose(-1, -2) @ d_o, bf(want_ds).transpose(-1, -2) @ q
    torch.testing.assert_close(heads(dqkv, 2 * D), want_dv, atol=2e-2 * want_dv.abs().max().item(), rtol=2e-2)
    torch.testing.assert_close(heads(dqkv, D), want_dk, atol=2e-2 * max(want_dk.abs().max().item(), 1e-3), rtol=2e-2)


GEMM_TN = _cases(1, 8, lambda r: (int(r.integers(1, 3)), int(r.integers(1, 3)), int(r.integers(1, 260)), int(r.integers(1, 300)),
                                  8 * int(r.integers(1, 24)), int(r.integers(0, 3)), int(r.integers(0, 2))))


@pytest.mark.parametrize("nb,nh,M,J,N,extra_x,extra_o", GEMM_TN)
def test_gemm_tn_ragged(nb, nh, M, J, N, extra_x, extra_o):
    ldx, ldy, ldo = (J + 7) // 8 * 8 + 8 * extra_x, nh * N, N + 8 * extra_o
    gen = torch.Generator().manual_seed(M * 3 + J)
    xb, yb = bits(torch.randn(nb, nh, M, ldx, generator=gen)), bits(torch.randn(nb, M, nh, N, generator=gen))
    out = emu.gemm_tn(xb.reshape(-1), ldx, yb.reshape(-1), ldy, M, J, N, ldo, nb * nh * J * ldo,
                      batch=(nb, nh, nh * M * ldx, M * ldx, M * ldy, N, nh * J * ldo, J * ldo))
    o = unbits(out).reshape(nb, nh, J, ldo)
    ref = torch.einsum("bhmj,bmhn->bhjn", unbits(xb).double()[..., :J], unbits(yb).double()).float()
    torch.testing.assert_close(o[..., :N], ref, rtol=1e-2, atol=1e-2 * max(ref.abs().max().item(), 1e-3))
    assert (o[..., N:] == 0).all()


WGRAD = _cases(2, 6, lambda r: (int(r.integers(1, 3)), int(r.integers(4, 14)), int(r.integers(4, 14)), 64 * int(r.integers(1, 3)),
                                8 * int(r.integers(1, 24)), int(r.choice([1, 3])), int(r.choice([1, 2])), int(r.integers(0, 4))))


@pytest.mark.parametrize("B,H,W,Ci,Co,R,st,split", WGRAD)
def test_weight_and_bias_gradient_ragged(B, H, W, Ci, Co, R, st, split):
    gen = torch.Generator().manual_seed(H * 31 + W)
    x = bf(torch.randn(B, Ci, H, W, generator=gen))
    w = bf(torch.randn(Co, Ci, R, R, generator=gen) / (Ci * R * R) ** 0.5).requires_grad_(True)
    y = F.conv2d(x, w, stride=st, padding=R // 2)
    dy = bf(torch.randn(y.shape, generator=gen))
    y.backward(dy)
    dw, db = emu.conv_wgrad_bias(bits(x.permute(0, 2, 3, 1)), bits(dy.permute(0, 2, 3, 1)), emu.geom(B, H, W, Ci, Co, R, R, st, R // 2), split=split)
    torch.testing.assert_close(torch.from_numpy(dw), w.grad.permute(0, 2, 3, 1).reshape(Co, -1), atol=3e-3, rtol=3e-3)
    torch.testing.assert_close(torch.from_numpy(db), dy.permute(0, 2, 3, 1).reshape(-1, Co).sum(0), atol=2e-3, rtol=1e-4)


STEM = _cases(3, 6, lambda r: (int(r.integers(1, 3)), int(r.integers(3, 15)), int(r.integers(3, 15)), 8 * int(r.choice([1, 2, 4, 8]))))


@pytest.mark.parametrize("B,H,W,C", STEM)
def test_fused_stem_pool_ragged(B, H, W, C):
    M = B * H * W
    gen = torch.Generator().manual_seed(H * 17 + W)
    z = bf(torch.randn(B, C, H, W, generator=gen) * 2)
    gamma, beta = (1 + 0.2 * torch.randn(C, generator=gen)).numpy(), (0.1 * torch.randn(C, generator=gen) - 0.3).numpy()
    zb = bits(z.permute(0, 2, 3, 1)).reshape(M, C)
    a_bits, mean, invstd = emu.bn_forward(zb, M, C, gamma, beta, relu=True)
    y0, arg0 = emu.maxpool(a_bits, B, H, W, C)
    y1, arg1 = emu.bn_relu_maxpool(zb, mean, invstd, gamma, beta, B, H, W, C)
    assert np.array_equal(y0, y1) and np.array_equal(arg0, arg1)
    dyb = bits(bf(torch.randn(B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C, generator=gen)))
    da = emu.maxpool_bwd(arg0, dyb, B, H, W, C).reshape(M, C)
    dz0, _, dg0, db0 = emu.bn_backward(da, a_bits, zb, mean, invstd, gamma, M, C)
    dz1, dg1, db1, _ = emu.bn_pool_backward(arg1, dyb, zb, mean, invstd, gamma, beta, B, H, W, C)
    torch.testing.assert_close(unbits(dz1), unbits(dz0), atol=3e-2, rtol=3e-2)
    np.testing.assert_allclose(dg1, dg0, atol=5e-2, rtol=1e-2)
    np.testing.assert_allclose(db1, db0, atol=5e-2, rtol=1e-2)


CONV = _cases(5, 8, lambda r: (int(r.integers(1, 4)), int(r.integers(3, 15)), int(r.integers(3, 15)), 64 * int(r.integers(1, 4)),
                               64 * int(r.integers(1, 4)), int(r.choice([1, 3])), int(r.choice([1, 2]))))


@pytest.mark.parametrize("B,H,W,Ci,Co,R,st", CONV)
def test_fused_conv_bn_paths_ragged(B, H, W, Ci, Co, R, st):
    """forward with the next BatchNorm's [sum, sum^2] from the store pass, and the data gradient with addend + ReLU gate recomputed from
    z + the BatchNorm-backward reductions, on random geometries (partial row / column tiles, stride 2 parity classes, 1x1 and 3x3)"""
    pad = 0 if (st == 2 and R == 1) else R // 2
    gen = torch.Generator().manual_seed(H * 13 + W * 5 + Ci)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()  # noqa: E731
    x = bf(torch.randn(B, Ci, H, W, generator=gen)).requires_grad_(True)
    w = bf(torch.randn(Co, Ci, R, R, generator=gen) / (Ci * R * R) ** 0.5).requires_grad_(True)
    y = F.conv2d(x, w, stride=st, padding=pad)
    dy = bf(torch.randn(y.shape, generator=gen))
    y.backward(dy)
    g = emu.geom(B, H, W, Ci, Co, R, R, st, pad)
    zb, sums = emu.conv_fwd_bn(bits(nhwc(x.detach())), bits(w.detach().permute(0, 2, 3, 1)), g)
    zf = unbits(zb)
    torch.testing.assert_close(zf, bf(nhwc(y.detach()).reshape(-1, Co)), atol=3e-2, rtol=2e-2)
    np.testing.assert_allclose(sums[0], zf.sum(0).numpy(), rtol=1e-3, atol=1e-2)
    np.testing.assert_allclose(sums[1], (zf * zf).sum(0).numpy(), rtol=1e-3, atol=1e-2)
    Mi = B * H * W
    zin = bf(torch.randn(Mi, Ci, generator=gen))
    mean, invstd = zin.mean(0), 1 / torch.sqrt(zin.var(0, unbiased=False) + 1e-5)
    gamma, beta = 1 + 0.1 * torch.randn(Ci, generator=gen), 0.1 * torch.randn(Ci, generator=gen)
    add = bf(torch.randn(Mi, Ci, generator=gen))
    dxb, s2, _, _ = emu.conv_dgrad_bn(bits(nhwc(dy)), bits(w.detach().permute(1, 2, 3, 0)), g, bits(zin), mean.numpy(), invstd.numpy(),
                                      gamma.numpy(), beta.numpy(), addend_bits=bits(add))
    o = torch.addcmul(beta, zin - mean, invstd * gamma)
    want = (nhwc(x.grad).reshape(-1, Ci) + add) * (bf(torch.relu(o)) > 0)
    got = unbits(dxb)
    sure = o.abs() > 1e-3                                   # (a gate exactly at its threshold may fall either way)
    torch.testing.assert_close(got[sure], bf(want)[sure], atol=4e-2, rtol=3e-2)
    np.testing.assert_allclose(s2[0], got.sum(0).numpy(), rtol=2e-3, atol=2e-2)
    np.testing.assert_allclose(s2[1], (got * (zin - mean) * invstd).sum(0).numpy(), rtol=2e-3, atol=3e-2)


DECODE = _cases(9, 6, lambda r: (int(r.choice([1, 2, 3])), int(r.integers(11, 40)), int(r.integers(11, 70)), int(r.integers(1, 3)),
                                 int(r.integers(1, 4)), float(r.uniform(1, 6))))


@pytest.mark.parametrize("ds,h,w,b,k,sharp", DECODE)
def test_decode_forward_ragged(ds, h, w, b, k, sharp):
    """fused decode at odd map sizes (partial row groups / column strips) and every downsample factor against the oracle's
    upsample -> softmax(T=1000) -> expectation -> 5x5 confidence"""
    from oracle import restated as O

    g = torch.Generator().manual_seed(h * 100 + w)
    heat = torch.softmax(sharp * torch.randn(b, k, h * w, generator=g), -1).reshape(b, k, h, w)
    kp, _, conf, _ = emu.decode_fwd(heat.numpy(), ds)
    want_kp, want_conf = O.soft_argmax(heat, ds, 1000.0)
    np.testing.assert_allclose(kp.reshape(b, -1), want_kp.numpy(), atol=2e-3)
    np.testing.assert_allclose(conf, want_conf.numpy(), atol=5e-5)
