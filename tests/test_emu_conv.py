"""Kernel-logic tests of csrc/conv.hip (MFMA implicit GEMM) on the CPU emulator vs torch fp32 convolutions of the
SAME bf16-rounded operands (so the only difference is fp32 summation order)."""

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.hipemu import emu

pytestmark = pytest.mark.usefixtures("kernel_backend")

bf = lambda t: t.to(torch.bfloat16).float()  # noqa: E731


def nhwc(t):  # (B,C,H,W) -> (B,H,W,C)
    return t.permute(0, 2, 3, 1).contiguous()


CASES = [
    # B, Hi, Wi, Ci, Co, R, stride, pad
    (2, 8, 8, 64, 64, 1, 1, 0),      # 1x1, N = 64 tile
    (1, 9, 7, 64, 128, 3, 1, 1),     # 3x3, M = 63 (partial tile), N = 128
    (2, 10, 10, 128, 192, 3, 2, 1),  # 3x3 stride 2, N = 192 (partial second N tile), two K steps per tap
    (3, 8, 8, 128, 64, 1, 2, 0),     # 1x1 stride 2 (downsample)
    (2, 12, 9, 64, 64, 3, 1, 1),     # 3x3 "same": the weight-gradient's fixed-offset addressing incl. image wrap, ragged last K step
    (3, 16, 16, 64, 128, 3, 1, 1),   # same, M a multiple of 64, several images per pixel slice
]


@pytest.mark.parametrize("case", CASES)
def test_conv_fwd_dgrad_wgrad(case):
    B, Hi, Wi, Ci, Co, R, st, pad = case
    gen = torch.Generator().manual_seed(sum(case))
    x = bf(torch.randn(B, Ci, Hi, Wi, generator=gen)).requires_grad_(True)
    w = bf(torch.randn(Co, Ci, R, R, generator=gen) / (Ci * R * R) ** 0.5).requires_grad_(True)
    y = F.conv2d(x, w, stride=st, padding=pad)
    dy = bf(torch.randn(y.shape, generator=gen))
    y.backward(dy)
    g = emu.geom(B, Hi, Wi, Ci, Co, R, R, st, pad)
    xb = emu.to_bf16_bits(nhwc(x.detach()))
    wg = emu.to_bf16_bits(w.detach().permute(0, 2, 3, 1))  # [Co][R][S][Ci]
    wd = emu.to_bf16_bits(w.detach().permute(1, 2, 3, 0))  # [Ci][R][S][Co]
    dyb = emu.to_bf16_bits(nhwc(dy))
    # forward (bf16 + fp32 outputs)
    ob, of = emu.conv_fwd(xb, wg, g, f32_out=True)
    want = nhwc(y.detach()).reshape(-1, Co)
    torch.testing.assert_close(torch.from_numpy(of), want, atol=2e-4, rtol=2e-4)
    torch.testing.assert_close(emu.from_bf16_bits(ob), bf(want), atol=2e-2, rtol=1e-2)
    # data gradient (+ addend)
    add = bf(torch.randn(B * Hi * Wi, Ci, generator=gen))
    _, df = emu.conv_dgrad(dyb, wd, g, addend_bits=emu.to_bf16_bits(add), f32_out=True)
    want = nhwc(x.grad).reshape(-1, Ci) + add
    torch.testing.assert_close(torch.from_numpy(df), want, atol=5e-4, rtol=5e-4)
    # fused ReLU backward: bf16 output zeroed where the activation is <= 0, single rounding after the addend
    act = bf(torch.relu(torch.randn(B * Hi * Wi, Ci, generator=gen)))
    dbits, _ = emu.conv_dgrad(dyb, wd, g, addend_bits=emu.to_bf16_bits(add), mask_bits=emu.to_bf16_bits(act))
    torch.testing.assert_close(emu.from_bf16_bits(dbits), bf(want * (act > 0)), atol=3e-2, rtol=2e-2)
    # weight gradient (pixel slices reduced through the workspace)
    for split in (0, 3):
        dw = emu.conv_wgrad(xb, dyb, g, split=split)
        want = w.grad.permute(0, 2, 3, 1).reshape(Co, -1)
        torch.testing.assert_close(torch.from_numpy(dw), want, atol=2e-3, rtol=2e-3)
    # ... and with the bias gradient (column sums of dy) taken in the same pass
    dw, db = emu.conv_wgrad_bias(xb, dyb, g, split=3)
    torch.testing.assert_close(torch.from_numpy(dw), want, atol=2e-3, rtol=2e-3)
    torch.testing.assert_close(torch.from_numpy(db), nhwc(dy).reshape(-1, Co).sum(0), atol=1e-3, rtol=1e-4)


@pytest.mark.parametrize("R", [1, 3])
@pytest.mark.parametrize("B", [2, 3, 4, 5, 6])
def test_weight_gradient_two_steps_in_flight(B, R, monkeypatch):
    """conv_wgrad_kernel keeps two K steps in flight where every step takes the fixed-offset loads (round 3): pixel slices of 2 .. 6 K steps
    (even and odd tails of the loop unrolled by two), with and without the padding walk, one slice and several (a one-step slice takes the
    old loop), against autograd on the same bf16 operands"""
    monkeypatch.setenv("LP_WGRAD_PIPE", "0")
    Hi = Wi = 8
    Ci, Co, pad = 64, 128, R // 2
    gen = torch.Generator().manual_seed(100 * B + R)
    x = bf(torch.randn(B, Ci, Hi, Wi, generator=gen))
    w = bf(torch.randn(Co, Ci, R, R, generator=gen) / (Ci * R * R) ** 0.5).requires_grad_(True)
    y = F.conv2d(x, w, padding=pad)
    dy = bf(torch.randn(y.shape, generator=gen))
    y.backward(dy)
    g = emu.geom(B, Hi, Wi, Ci, Co, R, R, 1, pad)
    want = w.grad.permute(0, 2, 3, 1).reshape(Co, -1)
    for split in (1, 2, B):
        dw = emu.conv_wgrad(emu.to_bf16_bits(nhwc(x)), emu.to_bf16_bits(nhwc(dy)), g, split=split)
        assert emu.lib().lp_conv_last_kernel() == 2   # LP_CONV_KERNEL_WGRAD
        torch.testing.assert_close(torch.from_numpy(dw), want, atol=2e-3, rtol=2e-3)


def test_conv_bias_and_column_mask():
    gen = torch.Generator().manual_seed(3)
    x = bf(torch.randn(1, 64, 6, 6, generator=gen))
    w = bf(torch.randn(64, 64, 1, 1, generator=gen) / 8)
    bias = torch.randn(64, generator=gen)
    g = emu.geom(1, 6, 6, 64, 64, 1, 1, 1, 0)
    _, of = emu.conv_fwd(emu.to_bf16_bits(nhwc(x)), emu.to_bf16_bits(w.permute(0, 2, 3, 1)), g, bias=bias.numpy(), f32_out=True,
                         ldo=32, n_store=17)
    want = nhwc(F.conv2d(x, w, bias)).reshape(-1, 64)
    torch.testing.assert_close(torch.from_numpy(of[:, :17]), want[:, :17], atol=2e-4, rtol=2e-4)
    assert not of[:, 17:].any()


def test_conv_transpose_via_dgrad():
    """ConvTranspose2d(k3,s2,p1,op1) forward == lp_conv_dgrad of the mirrored conv; backward-data == lp_conv_fwd."""
    gen = torch.Generator().manual_seed(4)
    B, cin, cout, h = 2, 64, 17, 6
    x = bf(torch.randn(B, cin, h, h, generator=gen)).requires_grad_(True)
    wt = bf(torch.randn(cin, cout, 3, 3, generator=gen) / 24).requires_grad_(True)
    bias = torch.randn(cout, generator=gen)
    y = F.conv_transpose2d(x, wt, bias, stride=2, padding=1, output_padding=1)
    dy = bf(torch.randn(y.shape, generator=gen))
    y.backward(dy)
    cpad = 64
    # mirrored conv: input = big tensor (2h, cpad channels), output = small tensor (h, cin)
    g = emu.geom(B, 2 * h, 2 * h, cpad, cin, 3, 3, 2, 1, Ho=h, Wo=h)
    wg = torch.zeros(cin, 3, 3, cpad)
    wg[..., :cout] = wt.detach().permute(0, 2, 3, 1)           # [Co=cin][R][S][Ci=cout_pad]
    wd = wg.permute(3, 1, 2, 0).contiguous()                    # [Ci][R][S][Co]
    bpad = torch.zeros(cpad)
    bpad[:cout] = bias
    _, yf = emu.conv_dgrad(emu.to_bf16_bits(nhwc(x.detach())), emu.to_bf16_bits(wd), g, bias=bpad.numpy(), f32_out=True)
    want = nhwc(y.detach()).reshape(-1, cout)
    torch.testing.assert_close(torch.from_numpy(yf[:, :cout]), want, atol=3e-4, rtol=3e-4)
    assert not yf[:, cout:].any()
    # backward-data through lp_conv_fwd on the padded gradient
    dyp = torch.zeros(B, 2 * h, 2 * h, cpad)
    dyp[..., :cout] = nhwc(dy)
    _, dxf = emu.conv_fwd(emu.to_bf16_bits(dyp), emu.to_bf16_bits(wg), g, f32_out=True)
    torch.testing.assert_close(torch.from_numpy(dxf), nhwc(x.grad).reshape(-1, cin), atol=5e-4, rtol=5e-4)
    # weight gradient
    dw = emu.conv_wgrad(emu.to_bf16_bits(dyp), emu.to_bf16_bits(nhwc(x.detach())), g)
    want = wt.grad.permute(0, 2, 3, 1)  # [cin][3][3][cout]
    torch.testing.assert_close(torch.from_numpy(dw).reshape(cin, 3, 3, cpad)[..., :cout], want, atol=2e-3, rtol=2e-3)


def test_stem_fwd_wgrad():
    gen = torch.Generator().manual_seed(5)
    B, H = 2, 20
    x = bf(torch.randn(B, 3, H, H, generator=gen))
    w = bf(torch.randn(64, 3, 7, 7, generator=gen) / 12).requires_grad_(True)
    y = F.conv2d(x, w, stride=2, padding=3)
    dy = bf(torch.randn(y.shape, generator=gen))
    y.backward(dy)
    x4 = torch.zeros(B, H, H, 4)
    x4[..., :3] = nhwc(x)
    wp = torch.zeros(64, 8, 8, 4)
    wp[:, :7, :7, :3] = w.detach().permute(0, 2, 3, 1)
    g = emu.geom(B, H, H, 4, 64, 7, 7, 2, 3)
    ob = emu.stem_fwd(emu.to_bf16_bits(x4), emu.to_bf16_bits(wp), g)
    want = nhwc(y.detach()).reshape(-1, 64)
    torch.testing.assert_close(emu.from_bf16_bits(ob), bf(want), atol=3e-2, rtol=2e-2)
    dw = torch.from_numpy(emu.stem_wgrad(emu.to_bf16_bits(x4), emu.to_bf16_bits(nhwc(dy)), g)).reshape(64, 8, 8, 4)
    torch.testing.assert_close(dw[:, :7, :7, :3], w.grad.permute(0, 2, 3, 1), atol=3e-3, rtol=3e-3)
    assert not dw[:, 7].any() and not dw[:, :, 7].any() and not dw[..., 3].any()


@pytest.mark.parametrize("B,H,W,split", [(1, 128, 128, 0), (2, 24, 256, 0), (1, 12, 128, 1), (3, 10, 128, 7)])
def test_stem_wgrad_from_the_staged_neighbourhood(B, H, W, split, monkeypatch):
    """stem_wgrad_nb_kernel (csrc/conv_stem_wgrad.h; output rows of 64 x n pixels) against autograd and against conv_wgrad_kernel<64, stem>
    on the same operands: image borders on all four sides, several 64-pixel strips per output row, images / rows changing inside a slice,
    one slice and ragged slice counts; the padding entries of the [64][8][8][4] layout stay zero."""
    gen = torch.Generator().manual_seed(11 + B + H + W + split)
    x = bf(torch.randn(B, 3, H, W, generator=gen))
    w = bf(torch.randn(64, 3, 7, 7, generator=gen) / 12).requires_grad_(True)
    y = F.conv2d(x, w, stride=2, padding=3)
    dy = bf(torch.randn(y.shape, generator=gen))
    y.backward(dy)
    x4 = torch.zeros(B, H, W, 4)
    x4[..., :3] = nhwc(x)
    g = emu.geom(B, H, W, 4, 64, 7, 7, 2, 3)
    xb, db = emu.to_bf16_bits(x4), emu.to_bf16_bits(nhwc(dy))
    dw = torch.from_numpy(emu.stem_wgrad(xb, db, g, split)).reshape(64, 8, 8, 4)
    assert emu.lib().lp_conv_last_kernel() == 8   # LP_CONV_KERNEL_STEM_WGRAD_NB
    scale = float(w.grad.abs().max())
    torch.testing.assert_close(dw[:, :7, :7, :3], w.grad.permute(0, 2, 3, 1), atol=2e-5 * scale * (B * H * W) ** 0.5, rtol=1e-4)
    assert not dw[:, 7].any() and not dw[:, :, 7].any() and not dw[..., 3].any()
    monkeypatch.setenv("LP_STEM_WGRAD_NB", "0")
    old = torch.from_numpy(emu.stem_wgrad(xb, db, g, 0)).reshape(64, 8, 8, 4)
    assert emu.lib().lp_conv_last_kernel() == 2   # LP_CONV_KERNEL_WGRAD
    torch.testing.assert_close(dw, old, atol=2e-5 * scale * (B * H * W) ** 0.5, rtol=1e-4)
    if split == 1:   # one slice each: the same pixels in the same order through the same MFMA - bit for bit
        one = torch.from_numpy(emu.stem_wgrad(xb, db, g, 1)).reshape(64, 8, 8, 4)
        assert torch.equal(dw, one)


@pytest.mark.parametrize("case", CASES)
def test_conv_fused_batchnorm_reductions(case):
    """lp_conv_fwd_bn == lp_conv_fwd + lp_bn_stats;  lp_conv_dgrad_bn == lp_conv_dgrad(+mask) + lp_bn_bwd_reduce, for both mask
    sources (the activation tensor, its 1-bit form written by lp_bn_apply, or recomputed from the pre-normalisation tensor)."""
    B, Hi, Wi, Ci, Co, R, st, pad = case
    gen = torch.Generator().manual_seed(7 + sum(case))
    g = emu.geom(B, Hi, Wi, Ci, Co, R, R, st, pad)
    x = emu.to_bf16_bits(torch.randn(B, Hi, Wi, Ci, generator=gen))
    w = torch.randn(Co, R, R, Ci, generator=gen) / (Ci * R * R) ** 0.5
    wg, wd = emu.to_bf16_bits(w), emu.to_bf16_bits(w.permute(3, 1, 2, 0))
    # ---- forward statistics
    z_plain, _ = emu.conv_fwd(x, wg, g)
    z, sums = emu.conv_fwd_bn(x, wg, g)
    assert np.array_equal(z, z_plain)
    zf = emu.from_bf16_bits(z).double()
    np.testing.assert_allclose(sums[0], zf.sum(0).numpy(), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(sums[1], (zf * zf).sum(0).numpy(), rtol=1e-5, atol=1e-4)
    # ---- backward reductions: dx is the gradient of a = relu(BN(zin) [+ residual]) where zin has the conv INPUT's shape
    Mi = B * Hi * Wi
    zin = torch.randn(Mi, Ci, generator=gen)
    zin_bits = emu.to_bf16_bits(zin)
    gamma, beta = torch.rand(Ci, generator=gen) + 0.5, torch.randn(Ci, generator=gen) * 0.3
    a_bits, mean, invstd, relu_bits = emu.bn_forward(zin_bits, Mi, Ci, gamma.numpy(), beta.numpy(), relu=True, want_bits=True)
    want_bits = np.packbits((emu.from_bf16_bits(a_bits).numpy() > 0).reshape(-1, 8), axis=1, bitorder="little").reshape(-1)
    assert np.array_equal(relu_bits, want_bits)
    dy = emu.to_bf16_bits(torch.randn(B * g.Ho * g.Wo, Co, generator=gen))
    add = emu.to_bf16_bits(torch.randn(Mi, Ci, generator=gen))
    want_dx, _ = emu.conv_dgrad(dy, wd, g, addend_bits=add, mask_bits=a_bits)
    _, _, want_dgamma, want_dbeta = emu.bn_backward(want_dx, None, zin_bits, mean, invstd, gamma.numpy(), Mi, Ci)
    for mask, bits in ((a_bits, None), (None, None), (None, relu_bits)):
        dx, sums, dbeta, dgamma = emu.conv_dgrad_bn(dy, wd, g, zin_bits, mean, invstd, gamma.numpy(), beta.numpy(), addend_bits=add,
                                                    mask_bits=mask, relu_bits=bits)
        assert np.array_equal(dx, want_dx)
        np.testing.assert_allclose(sums[0], want_dbeta, rtol=1e-4, atol=1e-3)
        np.testing.assert_allclose(sums[1], want_dgamma, rtol=1e-4, atol=1e-3)
        np.testing.assert_allclose(dbeta, sums[0], rtol=1e-5, atol=1e-5)   # same addends (atomic order may differ on the device)
        np.testing.assert_allclose(dgamma, sums[1], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("cap", ["1", "3"])
def test_conv_persistent_tile_walk(cap, kernel_backend):
    """The MFMA convolution kernel is persistent (one workgroup walks several tiles, prefetching the next tile's operands under
    the current tile's store pass).  The workgroup cap is read once per process, so the multi-tile walk is exercised by
    re-running this file's kernel tests in a child process with a tiny cap."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, LP_CONV_MAX_WGS=cap)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_emu_conv.py", "-q", "-x", "-m", "gpu" if kernel_backend == "gpu" else "not gpu", "-k",
                        "fwd_dgrad_wgrad or fused_batchnorm or transpose", "-p", "no:cacheprovider"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_gemm_nt_batched_strided():
    """lp_gemm_nt: per-(image, head) products out of an interleaved QKV-like tensor (row pitch > K, head offset inside the row),
    ragged M and N (not multiples of the tile), zero-padded output pitch, bias, and the plain un-batched Linear case."""
    gen = torch.Generator().manual_seed(5)
    nb, nh, T, d, ld = 2, 3, 77, 64, 3 * 3 * 64          # qkv rows: [q heads | k heads | v heads]
    qkv = bf(torch.randn(nb * T, ld, generator=gen))
    bits = emu.to_bf16_bits(qkv).reshape(-1)
    ldc = 128                                               # scores padded to 128 columns per row
    # S[z] = Q_z K_z^T : A = q slice (offset h*64), B = k slice (offset 192 + h*64); both pitch ld
    k_off = nh * d
    b_view = bits[k_off:]                                   # B base pointer = first K column
    out = emu.gemm_nt(bits, ld, b_view, ld, T, T, d, ldc, nb * nh * T, n_store=ldc,
                      batch=(nb, nh, T * ld, d, T * ld, d, nh * T * ldc, T * ldc))
    got = emu.from_bf16_bits(out.reshape(nb, nh, T, ldc))[..., :T]
    q = qkv.reshape(nb, T, 3, nh, d)[:, :, 0].permute(0, 2, 1, 3)
    k = qkv.reshape(nb, T, 3, nh, d)[:, :, 1].permute(0, 2, 1, 3)
    want = q @ k.transpose(-1, -2)
    torch.testing.assert_close(got, bf(want), atol=3e-2, rtol=2e-2)
    # plain Linear with bias, fp32 output, N not a multiple of 8
    M, K, N = 150, 128, 52
    x = bf(torch.randn(M, K, generator=gen))
    w = bf(torch.randn(N, K, generator=gen) / K ** 0.5)
    bias = torch.randn(N, generator=gen)
    of = emu.gemm_nt(emu.to_bf16_bits(x).reshape(-1), K, emu.to_bf16_bits(w).reshape(-1), K, M, N, K, N, M, bias=bias.numpy(), f32_out=True)
    torch.testing.assert_close(torch.from_numpy(of.reshape(M, N)), x @ w.T + bias, atol=3e-4, rtol=3e-4)


@pytest.mark.parametrize("nh,M,J,N,ldx", [(2, 77, 77, 64, 80), (1, 130, 150, 136, 152), (2, 64, 8, 8, 8)])
def test_gemm_tn_batched_strided(nh, M, J, N, ldx):
    """lp_gemm_tn (attention's dV = P^T dO, dK = dS^T Q): both operands contracted over their row index; ragged M and J, x pitch wider
    than J, y rows interleaving the heads (head h at column offset h * N, like the QKV rows), output pitch wider than N."""
    torch.manual_seed(5)
    nb = 2
    ldy, ldo = nh * N, N + 8
    xb = emu.to_bf16_bits(torch.randn(nb, nh, M, ldx))
    yb = emu.to_bf16_bits(torch.randn(nb, M, nh, N))
    xf, yf = emu.from_bf16_bits(xb).double(), emu.from_bf16_bits(yb).double()
    out = emu.gemm_tn(xb.reshape(-1), ldx, yb.reshape(-1), ldy, M, J, N, ldo, nb * nh * J * ldo,
                      batch=(nb, nh, nh * M * ldx, M * ldx, M * ldy, N, nh * J * ldo, J * ldo))
    o = emu.from_bf16_bits(out).reshape(nb, nh, J, ldo)
    ref = torch.einsum("bhmj,bmhn->bhjn", xf[..., :J], yf).float()
    assert torch.allclose(o[..., :N], ref, rtol=1e-2, atol=1e-2 * ref.abs().max().item())
    assert (o[..., N:] == 0).all()  # the pitch padding of the output stays untouched
