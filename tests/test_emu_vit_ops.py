"""Kernel-logic tests of csrc/vit.hip (the ViT-S/16 glue between the GEMMs) against plain torch fp32, on the CPU emulator and,
under -m gpu, on the device."""

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.hipemu import emu

pytestmark = pytest.mark.usefixtures("kernel_backend")

bf = lambda t: t.to(torch.bfloat16).float()  # noqa: E731
bits, unbits = emu.to_bf16_bits, emu.from_bf16_bits


def test_patchify_and_tokens():
    gen = torch.Generator().manual_seed(0)
    B, H, W, P, D = 2, 32, 48, 16, 64
    img = torch.randn(B, 3, H, W, generator=gen)
    got = unbits(emu.vit_patchify(img.numpy(), P))
    want = F.unfold(img, kernel_size=P, stride=P).transpose(1, 2).reshape(-1, 3 * P * P)  # (c, ky, kx) order, row-major patches
    torch.testing.assert_close(got, bf(want), atol=0, rtol=0)
    Np = (H // P) * (W // P)
    patch = bf(torch.randn(B * Np, D, generator=gen))
    cls, pos = torch.randn(D, generator=gen), torch.randn(Np + 1, D, generator=gen)
    x = torch.from_numpy(emu.vit_tokens_fwd(bits(patch), cls.numpy(), pos.numpy(), B, Np, D))
    want = torch.cat([cls.expand(B, 1, D), patch.reshape(B, Np, D)], 1) + pos
    torch.testing.assert_close(x, want, atol=1e-6, rtol=0)
    dx = torch.randn(B, Np + 1, D, generator=gen)
    dpatch, dpos = emu.vit_tokens_bwd(dx.numpy(), B, Np, D)
    torch.testing.assert_close(unbits(dpatch), bf(dx[:, 1:].reshape(-1, D)), atol=0, rtol=0)
    torch.testing.assert_close(torch.from_numpy(dpos), dx.sum(0), atol=1e-5, rtol=1e-5)


def test_small_matmul_is_interpolation_and_adjoint():
    gen = torch.Generator().manual_seed(1)
    w, x = torch.randn(24, 9, generator=gen), torch.randn(9, 40, generator=gen)
    torch.testing.assert_close(torch.from_numpy(emu.small_matmul(w.numpy(), x.numpy())), w @ x, atol=1e-5, rtol=1e-5)
    g = torch.randn(24, 40, generator=gen)
    y0 = torch.randn(9, 40, generator=gen)
    got = emu.small_matmul(w.numpy(), g.numpy(), transpose_w=True, y0=y0.numpy())
    torch.testing.assert_close(torch.from_numpy(got), y0 + w.T @ g, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("M,D,drop_T", [(37, 384, 0), (3 * 5, 96, 5), (9, 1024, 0), (10, 768, 0), (7, 640, 0), (1, 128, 0), (4 * 7, 256, 7)])
def test_layernorm_fwd_bwd(M, D, drop_T):
    gen = torch.Generator().manual_seed(M + D)
    x = (torch.randn(M, D, generator=gen) * 2 + 0.3)
    delta = bf(torch.randn(M, D, generator=gen))
    gamma, beta = torch.rand(D, generator=gen) + 0.5, torch.randn(D, generator=gen)
    y, mean, rstd, xo = emu.layernorm_fwd(x.numpy(), gamma.numpy(), beta.numpy(), 1e-12, delta_bits=bits(delta), drop_T=drop_T)
    xs = (x + delta).requires_grad_(True)
    ref = F.layer_norm(xs, (D,), gamma, beta, 1e-12)
    keep = torch.ones(M, dtype=torch.bool)
    if drop_T:
        keep[::drop_T] = False
    torch.testing.assert_close(torch.from_numpy(xo), xs.detach(), atol=1e-6, rtol=0)
    torch.testing.assert_close(unbits(y), bf(ref.detach()[keep]), atol=2e-2, rtol=1e-2)
    torch.testing.assert_close(torch.from_numpy(mean), xs.detach().mean(1), atol=1e-5, rtol=1e-5)
    dy = bf(torch.randn(int(keep.sum()), D, generator=gen))
    dy_full = torch.zeros(M, D)
    dy_full[keep] = dy
    gamma_r, beta_r = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    F.layer_norm(xs, (D,), gamma_r, beta_r, 1e-12).backward(dy_full)
    dx0 = torch.randn(M, D, generator=gen)
    dx, dg, db = emu.layernorm_bwd(bits(dy), xs.detach().numpy(), mean, rstd, gamma.numpy(), dx0.numpy(), drop_T=drop_T)
    torch.testing.assert_close(torch.from_numpy(dx), dx0 + xs.grad, atol=2e-4, rtol=2e-4)
    torch.testing.assert_close(torch.from_numpy(dg), gamma_r.grad, atol=2e-3, rtol=2e-4)
    torch.testing.assert_close(torch.from_numpy(db), beta_r.grad, atol=2e-3, rtol=2e-4)
    # the variant that also leaves the updated stream gradient in bf16 (the next Linear backward's operand)
    dx2, dg2, db2, d16 = emu.layernorm_bwd(bits(dy), xs.detach().numpy(), mean, rstd, gamma.numpy(), dx0.numpy(), drop_T=drop_T, want_bf16=True)
    assert np.array_equal(dx2, dx)
    torch.testing.assert_close(unbits(d16), bf(torch.from_numpy(dx)), atol=0, rtol=0)


def test_gelu_fwd_bwd():
    gen = torch.Generator().manual_seed(2)
    x = bf(torch.randn(40, 64, generator=gen) * 2).requires_grad_(True)
    y = F.gelu(x)
    torch.testing.assert_close(unbits(emu.gelu(bits(x.detach()))), bf(y.detach()), atol=1e-2, rtol=1e-2)
    dy = bf(torch.randn(40, 64, generator=gen))
    y.backward(dy)
    torch.testing.assert_close(unbits(emu.gelu(bits(x.detach()), bits(dy))), bf(x.grad), atol=2e-2, rtol=2e-2)


def test_softmax_rows_fwd_bwd():
    gen = torch.Generator().manual_seed(3)
    rows, n, ld, scale = 21, 77, 128, 0.125
    s = bf(torch.randn(rows, ld, generator=gen) * 4)
    p = emu.softmax_rows(bits(s), n, scale)
    pf = unbits(p)
    want = torch.softmax(s[:, :n] * scale, -1)
    torch.testing.assert_close(pf[:, :n], bf(want), atol=4e-3, rtol=1e-2)
    assert not pf[:, n:].any()
    dp = bf(torch.randn(rows, ld, generator=gen))
    ds = unbits(emu.softmax_rows(bits(dp), n, scale, p_bits=p))
    pp, dd = pf[:, :n], dp[:, :n]
    want_ds = scale * pp * (dd - (dd * pp).sum(-1, keepdim=True))
    torch.testing.assert_close(ds[:, :n], bf(want_ds), atol=2e-3, rtol=2e-2)
    assert not ds[:, n:].any()


def test_transpose_batched_with_head_strides_and_zero_pad():
    gen = torch.Generator().manual_seed(4)
    nb, nh, T, d, ld = 2, 3, 77, 64, 3 * 3 * 64
    qkv = bf(torch.randn(nb * T, ld, generator=gen))
    ldo = 128  # V^T[z][d][t], t padded to 128 with zeros
    v_off = 2 * nh * d
    out = emu.transpose_batched(bits(qkv).reshape(-1)[v_off:], T, d, ld, T * ld, d, ldo, nh * d * ldo, d * ldo, nb, nh, nb * nh * d * ldo)
    got = unbits(out.reshape(nb, nh, d, ldo))
    v = qkv.reshape(nb, T, 3, nh, d)[:, :, 2].permute(0, 2, 3, 1)  # (nb, nh, d, T)
    torch.testing.assert_close(got[..., :T], v, atol=0, rtol=0)
    assert not got[..., T:].any()


def test_transpose_batched_scalar_path():
    gen = torch.Generator().manual_seed(6)
    R, Cc, ldi, ldo = 19, 13, 15, 21  # odd pitches: the 2-byte path
    x = bf(torch.randn(3, R, ldi, generator=gen))
    out = emu.transpose_batched(bits(x).reshape(-1), R, Cc, ldi, R * ldi, 0, ldo, Cc * ldo, 0, 3, 1, 3 * Cc * ldo)
    got = unbits(out.reshape(3, Cc, ldo))
    torch.testing.assert_close(got[..., :R], x[..., :Cc].transpose(1, 2), atol=0, rtol=0)
    assert not got[..., R:].any()


@pytest.mark.parametrize("nh", [1, 6, 12])
def test_attention_rowdot_head_counts(nh):
    """lp_attn_rowdot for ViT-S (6 heads: one pass of the wave) and ViT-B (12 heads: two passes)"""
    gen = torch.Generator().manual_seed(nh)
    rows, D = 37, nh * 64
    a, b = bf(torch.randn(rows, D + 8, generator=gen)), bf(torch.randn(rows, D + 8, generator=gen))
    got = emu.attn_rowdot(bits(a).reshape(-1), bits(b).reshape(-1), rows, nh, D + 8)
    want = (a[:, :D] * b[:, :D]).reshape(rows, nh, 64).sum(-1)
    torch.testing.assert_close(torch.from_numpy(got), want, atol=1e-4, rtol=1e-5)


def test_attention_score_gradient_fused():
    """lp_attn_rowdot + lp_attn_dscores == autograd of softmax(Q K^T scale) V with respect to the scaled scores' pre-scale input:
    dS = scale * P o (dO V^T - rowsum(dO o O)), heads interleaved in the token rows (QKV layout), ragged T, padded score pitch."""
    gen = torch.Generator().manual_seed(11)
    nb, nh, T, d = 2, 3, 77, 64
    D, Tp, scale = nh * d, 128 + 8, 0.125
    q, k, v = (bf(torch.randn(nb, nh, T, d, generator=gen)) for _ in range(3))
    s = (q @ k.transpose(-1, -2)).requires_grad_(True)
    p = torch.softmax(s * scale, -1)
    pb = bf(p.detach())                                         # the probabilities as the forward pass stored them
    o = bf(pb @ v)                                              # attention output (b, h, T, d)
    d_o = bf(torch.randn(nb, nh, T, d, generator=gen))
    # reference: the soft-max backward on the stored probabilities
    dp = d_o @ v.transpose(-1, -2)
    want = scale * pb * (dp - (dp * pb).sum(-1, keepdim=True))
    # device layout: token rows [nb*T][nh*d]
    rows = lambda t: t.permute(0, 2, 1, 3).reshape(nb * T, D).contiguous()  # noqa: E731
    do_bits, o_bits, v_bits = bits(rows(d_o)).reshape(-1), bits(rows(o)).reshape(-1), bits(rows(v)).reshape(-1)
    drow = emu.attn_rowdot(do_bits, o_bits, nb * T, nh, D)
    want_d = (d_o * o).sum(-1).permute(0, 2, 1).reshape(nb * T, nh)
    torch.testing.assert_close(torch.from_numpy(drow), want_d, atol=1e-4, rtol=1e-5)
    p_pad = torch.zeros(nb, nh, T, Tp)
    p_pad[..., :T] = pb
    out = emu.attn_dscores(do_bits, D, v_bits, D, bits(p_pad).reshape(-1), drow.reshape(-1), nh, T * nh, 1, scale, Tp, nb * nh * T * Tp,
                           T, T, d, batch=(nb, nh, T * D, d, T * D, d, nh * T * Tp, T * Tp))
    got = unbits(out).reshape(nb, nh, T, Tp)
    assert (got[..., T:] == 0).all()
    # rowsum(dP o P) and rowsum(dO o O) differ only by the bf16 rounding of O: compare at that level
    assert torch.allclose(got[..., :T], want, atol=2e-2 * want.abs().max().item(), rtol=2e-2)
    cos = F.cosine_similarity(got[..., :T].flatten(), want.flatten(), dim=0)
    assert cos > 0.9995, cos


@pytest.mark.parametrize("nb,nh,T,ldp", [(2, 2, 77, 80), (1, 3, 130, 192), (1, 1, 64, 64), (1, 2, 5, 8)])
def test_attention_forward_fused(nb, nh, T, ldp):
    """lp_attn_fwd == softmax(scale Q K^T) and P V per (image, head) on bf16-rounded Q / K / V out of interleaved QKV token rows:
    ragged T (partial key tile, partial query tile, idle waves), probability pitch wider than T (pad columns zero, incl. a whole
    pad-only key tile), several heads."""
    gen = torch.Generator().manual_seed(nb * 100 + T)
    d, scale = 64, 0.125
    D = nh * d
    ld = 3 * D + 8
    qkv = bf(torch.randn(nb * T, ld, generator=gen) * 1.5)
    pbits, obits = emu.attn_fwd(bits(qkv).reshape(-1), ld, D, 2 * D, nb, nh, T, scale, ldp, D)
    heads = lambda off: qkv[:, off:off + D].reshape(nb, T, nh, d).permute(0, 2, 1, 3)  # noqa: E731
    q, k, v = heads(0), heads(D), heads(2 * D)
    p = torch.softmax((q @ k.transpose(-1, -2)) * scale, -1)
    got_p = unbits(pbits).reshape(nb, nh, T, ldp)
    assert (got_p[..., T:] == 0).all()
    torch.testing.assert_close(got_p[..., :T], p, atol=4e-3, rtol=1e-2)          # bf16 storage of values <= 1
    torch.testing.assert_close(got_p[..., :T].sum(-1), torch.ones(nb, nh, T), atol=1e-2, rtol=0)
    got_o = unbits(obits).reshape(nb, T, nh, d).permute(0, 2, 1, 3)
    want_o = bf(got_p[..., :T]) @ v                                               # O is accumulated from the STORED probabilities
    torch.testing.assert_close(got_o, want_o, atol=2e-2, rtol=2e-2)
    torch.testing.assert_close(got_o, p @ v, atol=3e-2, rtol=3e-2)


@pytest.mark.parametrize("nb,nh,T,ldp", [(2, 2, 77, 80), (1, 3, 130, 192), (1, 1, 64, 64), (1, 2, 5, 8), (1, 1, 200, 256)])
def test_attention_backward_kv_fused(nb, nh, T, ldp):
    """lp_attn_bwd_kv: dS, dV = P^T dO and dK = dS^T Q in one pass over the stored probabilities == the soft-max backward and the two
    products spelled in torch on the same bf16 operands (ragged T: partial key workgroups, partial query tiles, idle waves, pad-only
    key tiles)."""
    gen = torch.Generator().manual_seed(nb * 1000 + T)
    d, scale = 64, 0.125
    D = nh * d
    ld = 3 * D
    qkv = bf(torch.randn(nb * T, ld, generator=gen))
    heads = lambda t2, off: t2[:, off:off + D].reshape(nb, T, nh, d).permute(0, 2, 1, 3)  # noqa: E731
    q, k, v = heads(qkv, 0), heads(qkv, D), heads(qkv, 2 * D)
    p = bf(torch.softmax((q @ k.transpose(-1, -2)) * scale, -1))          # the probabilities as stored by the forward pass
    o = bf(p @ v)
    d_o_rows = bf(torch.randn(nb * T, D, generator=gen))
    d_o = heads(d_o_rows, 0)
    drow = (d_o * o).sum(-1).permute(0, 2, 1).reshape(nb * T, nh).contiguous()
    p_pad = torch.zeros(nb, nh, T, ldp)
    p_pad[..., :T] = p
    ds_bits, dqkv_bits = emu.attn_bwd_kv(bits(qkv).reshape(-1), ld, 2 * D, bits(d_o_rows).reshape(-1), D, bits(p_pad).reshape(-1), ldp,
                                         drow.numpy(), nb, nh, T, scale, ld, D, 2 * D)
    dp = d_o @ v.transpose(-1, -2)
    want_ds = scale * p * (dp - drow.reshape(nb, T, nh).permute(0, 2, 1).unsqueeze(-1))
    got_ds = unbits(ds_bits).reshape(nb, nh, T, ldp)
    assert (got_ds[..., T:] == 0).all()
    torch.testing.assert_close(got_ds[..., :T], bf(want_ds), atol=2e-3 * want_ds.abs().max().item() + 1e-6, rtol=1e-2)
    dqkv = unbits(dqkv_bits)
    assert not dqkv[:, :D].any()                                           # the dQ columns are not this kernel's
    want_dv = p.transpose(-1, -2) @ d_o
    want_dk = bf(want_ds).transpose(-1, -2) @ q
    for name, off, want in (("dK", D, want_dk), ("dV", 2 * D, want_dv)):
        got = heads(dqkv, off)
        torch.testing.assert_close(got, want, atol=2e-2 * want.abs().max().item(), rtol=2e-2, msg=lambda m, n=name: f"{n}: {m}")




@pytest.mark.parametrize("M,N,K", [(300, 128, 64), (577, 256, 192), (256, 384, 128)])
def test_gemm_with_gelu_backward_in_the_store_pass(kernel_backend, M, N, K, monkeypatch):
    """lp_gemm_nt_gelu_bwd = lp_gemm_nt followed by lp_gelu_bwd, bit for bit (the inner rounding is kept), and its column sums are the
    bias gradient lp_gelu_bwd_colsum accumulates; several tiles per workgroup, a ragged last row tile"""
    monkeypatch.setenv("LP_CONV_MAX_WGS", "2")
    gen = torch.Generator().manual_seed(M + N + K)
    a = bf(torch.randn(M, K, generator=gen))
    b = bf(torch.randn(N, K, generator=gen) * 0.2)
    u = bf(torch.randn(M, N, generator=gen) * 1.5)
    two_pass = emu.gelu(bits(u), emu.gemm_nt(bits(a).ravel(), K, bits(b).ravel(), K, M, N, K, N, M).reshape(M, N))
    fused, colsum = emu.gemm_nt_gelu_bwd(bits(a), bits(b), bits(u), M, N, K)
    assert np.array_equal(fused, two_pass)
    want = unbits(fused).double().sum(0)
    torch.testing.assert_close(torch.from_numpy(colsum).double(), want, atol=1e-4 * float(want.abs().max()) + 1e-5, rtol=1e-5)
    with pytest.raises(Exception):   # a column count the pipelined kernel does not tile: reported, the caller keeps the two-pass form
        emu.gemm_nt_gelu_bwd(bits(a), bits(b)[:64], bits(u)[:, :64].copy(), M, 64, K)


@pytest.mark.parametrize("M,N,K", [(300, 128, 64), (577, 256, 192)])
def test_gemm_that_also_writes_its_gelu(kernel_backend, M, N, K, monkeypatch):
    """lp_gemm_nt_gelu_fwd = lp_gemm_nt (+ bias) and lp_gelu_fwd of its output, both bit for bit"""
    monkeypatch.setenv("LP_CONV_MAX_WGS", "2")
    gen = torch.Generator().manual_seed(M + N + K + 1)
    a = bf(torch.randn(M, K, generator=gen))
    b = bf(torch.randn(N, K, generator=gen) * 0.2)
    bias = torch.randn(N, generator=gen) * 0.3
    c0 = emu.gemm_nt(bits(a).ravel(), K, bits(b).ravel(), K, M, N, K, N, M, bias=bias.numpy()).reshape(M, N)
    c, act = emu.gemm_nt_gelu_fwd(bits(a), bits(b), bias.numpy(), M, N, K)
    assert np.array_equal(c, c0)
    assert np.array_equal(act, emu.gelu(c0))
    torch.testing.assert_close(unbits(act), bf(F.gelu(unbits(c0))), atol=1e-2, rtol=1e-2)


def test_gelu_keeps_its_relative_accuracy_in_the_tails():
    """gelu_phi (lp_common.h): x Phi(x) through an erfc fit with 1.2e-7 FRACTIONAL error - against fp64 the bf16 results agree everywhere,
    also where 0.5 x (1 + erf) cancels (torch's own fp32 F.gelu differs from fp64 in ~0.4 % of bf16 results there)"""
    gen = torch.Generator().manual_seed(9)
    x = bf(torch.cat([torch.randn(200000, generator=gen) * 1.5, torch.linspace(-9, 9, 20000)]))
    x = x[: x.numel() // 8 * 8]
    exact = (x.double() * 0.5 * torch.erfc(-x.double() / 2 ** 0.5)).float()
    got = unbits(emu.gelu(bits(x)))
    assert float((got != bf(exact)).float().mean()) < 2e-4
    dexact = (0.5 * torch.erfc(-x.double() / 2 ** 0.5) + x.double() * torch.exp(-0.5 * x.double() ** 2) / (2 * np.pi) ** 0.5).float()
    one = bf(torch.ones_like(x))
    dgot, dwant = unbits(emu.gelu(bits(x), bits(one))), bf(dexact)
    body = x.abs() < 3   # (the A&S form: 1.5e-7 ABSOLUTE - one bf16 place of the 1e-2-sized values far out in the negative tail, nothing where GELU' is O(1))
    assert float((dgot != dwant)[body].float().mean()) < 5e-4
    assert float((dgot - dwant).abs().max()) < 1e-6 + 2 ** -8 * float(dwant.abs().max()) and bool(((dgot - dwant).abs() <= 2 ** -7 * dwant.abs() + 1e-6).all())
