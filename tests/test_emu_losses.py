"""Kernel-logic tests of csrc/heatmap.hip and csrc/kploss.hip on the CPU emulator vs the reference golden vectors
and the oracle (values and gradients)."""

import numpy as np
import pytest
import torch

from oracle import restated as O
from tests.hipemu import emu

pytestmark = pytest.mark.usefixtures("kernel_backend")


def test_heatmap_gen_golden(golden):
    g = golden("heatmaps")
    np.testing.assert_allclose(emu.heatmap_gen(g["kp"], None, 128, 128, 32, 32), g["hm_novis"], atol=2e-7)
    np.testing.assert_allclose(emu.heatmap_gen(g["kp"], g["vis"], 128, 128, 32, 32), g["hm_vis"], atol=2e-7)
    np.testing.assert_allclose(emu.heatmap_gen(g["kp"], None, 128, 160, 32, 40, sigma=2.0), g["hm_rect"], atol=2e-7)


@pytest.mark.parametrize("h,w", [(24, 30), (7, 5), (96, 96), (520, 12), (3, 516)])
def test_heatmap_gen_paths_vs_oracle(h, w):
    """every store path of heatmap_gen_kernel against the oracle: separable profiles with 16-byte stores (w % 4 == 0), with scalar stores
    (ragged widths), and the per-pixel form for an axis longer than the profile buffers (> 512); out-of-range, NaN and edge keypoints"""
    gen = torch.Generator().manual_seed(h * 31 + w)
    img_h, img_w = 4 * h, 4 * w
    kp = torch.rand(3, 5, 2, generator=gen) * torch.tensor([img_w, img_h], dtype=torch.float32)
    kp[0, 0] = torch.tensor([float("nan"), 3.0])
    kp[0, 1] = torch.tensor([-30.0, 5.0])              # more than one heat-map pixel outside: zeros
    kp[1, 0] = torch.tensor([0.0, 0.0])
    kp[1, 1] = torch.tensor([img_w - 1e-3, img_h - 1e-3])
    kp[2, 0] = torch.tensor([-2.0, img_h + 2.0])       # within one heat-map pixel of the border: clamped Gaussian
    vis = torch.tensor([[2, 2, 1, 0, 2], [2, 2, 2, 1, 0], [2, 0, 1, 2, 2]], dtype=torch.int32)
    for v in (None, vis):
        want = O.generate_heatmaps(kp, img_h, img_w, (h, w), sigma=1.25, visibility=v).numpy()
        got = emu.heatmap_gen(kp.numpy(), None if v is None else v.numpy(), img_h, img_w, h, w)
        np.testing.assert_allclose(got, want, atol=2e-7)
        np.testing.assert_allclose(got.sum((2, 3)), want.sum((2, 3)), atol=2e-6)


def test_heatmap_mse_golden_and_grad(golden):
    g = golden("losses")
    loss, grad = emu.heatmap_mse(g["hm_targ"], g["hm_pred"], gout=0.7)
    assert loss == pytest.approx(float(g["heatmap_mse"]), rel=1e-5)
    p = g.t("hm_pred").clone().requires_grad_(True)
    (0.7 * O.heatmap_mse_loss(g.t("hm_targ"), p)).backward()
    np.testing.assert_allclose(grad, p.grad.numpy(), atol=1e-7, rtol=1e-5)


@pytest.mark.parametrize("name", ["heatmap_kl", "heatmap_js"])
def test_heatmap_divergences_golden_and_grad(golden, name):
    """HeatmapKLLoss / HeatmapJSLoss: value vs the golden of the verbatim reference classes, gradient vs autograd through the
    restated kornia divergences"""
    from lightning_pose_amd import _lib
    g = golden("losses")
    kind = _lib.HM_KL if name == "heatmap_kl" else _lib.HM_JS
    loss, grad = emu.heatmap_div(kind, g["hm_targ"], g["hm_pred"], gout=0.7)
    assert loss == pytest.approx(float(g[name]), rel=2e-5, abs=1e-7)
    p = g.t("hm_pred").clone().requires_grad_(True)
    fn = O.heatmap_kl_loss if name == "heatmap_kl" else O.heatmap_js_loss
    (0.7 * fn(g.t("hm_targ"), p)).backward()
    np.testing.assert_allclose(grad, p.grad.numpy(), rtol=2e-4, atol=1e-6)


def test_heatmap_mse_all_invalid_is_nan():
    t = np.zeros((2, 2, 8, 8), np.float32)
    loss, _ = emu.heatmap_mse(t, np.full_like(t, 0.1))
    assert np.isnan(loss)


def test_unimodal_mse_matches_oracle():
    gen = torch.Generator().manual_seed(11)
    s, k, h, w = 5, 4, 16, 16
    pred = torch.softmax(3 * torch.randn(s, k, h * w, generator=gen), -1).reshape(s, k, h, w).requires_grad_(True)
    kp = torch.rand(s, 2 * k, generator=gen) * 64
    kp[0, 0:2] = float("nan")
    kp[1, 2] = 500.0
    conf = torch.rand(s, k, generator=gen)
    want = O.unimodal_mse_loss(kp, pred, conf, 64, 64, prob_threshold=0.4)
    (1.3 * want).backward()
    loss, grad = emu.unimodal_mse(kp.reshape(s, k, 2).numpy(), pred.detach().numpy(), conf.numpy(), 64, 64, 0.4, gout=1.3)
    assert loss == pytest.approx(float(want), rel=1e-5)
    np.testing.assert_allclose(grad, pred.grad.numpy(), atol=1e-7, rtol=1e-4)
    loss0, grad0 = emu.unimodal_mse(kp.reshape(s, k, 2).numpy(), pred.detach().numpy(), conf.numpy(), 64, 64, 2.0)
    assert loss0 == 0.0 and not grad0.any()


@pytest.mark.parametrize("b,k,n,c", [(2, 5, 48, 8), (2, 17, 2300, 64), (1, 5, 70, 6), (2, 17, 4101, 64), (1, 3, 9216, 8)])
def test_softmax2d_fwd_bwd(b, k, n, c):
    """forward + backward; c % 8 == 0 takes the pixel-major backward (whole channel rows, pad channels zeroed by the kernel: the output
    buffer is pre-filled with garbage to prove it), c = 6 the per-map fallback (pad channels untouched: pre-filled with zeros)"""
    gen = torch.Generator().manual_seed(12)
    logits = torch.randn(b, n, c, generator=gen)
    x = logits[:, :, :k].permute(0, 2, 1).clone().requires_grad_(True)  # (b,k,n)
    p = torch.softmax(x, -1)
    gp = torch.randn(b, k, n, generator=gen)
    (p * gp).sum().backward()
    got = emu.softmax2d(logits.numpy(), k)
    np.testing.assert_allclose(got, p.detach().numpy(), atol=1e-7, rtol=1e-5)
    gin = emu.softmax2d_bwd(got, gp.numpy(), c)
    np.testing.assert_allclose(gin[:, :, :k], x.grad.permute(0, 2, 1).numpy(), atol=1e-4, rtol=1e-2)  # bf16 output
    assert not gin[:, :, k:].any()


@pytest.mark.parametrize("b,k,n,c", [(3, 17, 2500, 64), (2, 17, 300, 17), (1, 32, 1100, 32), (2, 32, 4099, 32)])
def test_softmax2d_pixel_major_and_fallback(b, k, n, c):
    """the head's layout (K logits of a pixel contiguous, padded to c): one 1024-lane workgroup per frame; c = 17 (rows not
    16-B readable) takes the per-map kernel"""
    gen = torch.Generator().manual_seed(b * 100 + k)
    logits = torch.randn(b, n, c, generator=gen) * 3
    want = torch.softmax(logits[:, :, :k].permute(0, 2, 1), -1)
    got = emu.softmax2d(logits.numpy(), k)
    np.testing.assert_allclose(got, want.numpy(), atol=1e-7, rtol=2e-5)


def test_temporal_golden_and_grad(golden):
    g = golden("losses")
    kp, conf = g["t_kp"], g["t_conf"]
    s = kp.shape[0]
    kp3 = kp.reshape(s, -1, 2)
    assert emu.temporal(kp3, None, 0.0, 0.0)[0] == pytest.approx(float(g["temporal_plain"]), rel=1e-5)
    assert emu.temporal(kp3, None, 5.0, 0.0)[0] == pytest.approx(float(g["temporal_eps"]), rel=1e-5)
    assert emu.temporal(kp3, conf, 3.0, 0.3)[0] == pytest.approx(float(g["temporal_conf"]), rel=1e-5)
    loss, grad = emu.temporal(kp3, conf, g["t_eps_list"], 0.3)
    assert loss == pytest.approx(float(g["temporal_epslist"]), rel=1e-5)
    x = g.t("t_kp").clone().requires_grad_(True)
    O.temporal_loss(x, g.t("t_conf"), g.t("t_eps_list"), 0.3).backward()
    np.testing.assert_allclose(grad.reshape(s, -1), x.grad.numpy(), atol=1e-6, rtol=1e-4)


def test_temporal_kat():
    """reference tests/losses/test_losses.py:314-408."""
    kp = np.array([[[0.0, 0.0]], [[2 ** 0.5, 2 ** 0.5]]], np.float32)
    assert emu.temporal(kp, None, 0.0, 0.0)[0] == pytest.approx(2.0, rel=1e-6)


@pytest.mark.parametrize("tag", ["sv99", "sv3"])
def test_pca_singleview_golden_and_grad(golden, tag):
    g = golden("losses")
    cols = g["pca_cols"].astype(np.int32).reshape(1, -1)
    test = g[f"pca_{tag}_test"]
    s = test.shape[0]
    loss, grad = emu.pca(test.reshape(s, -1, 2), cols, g[f"pca_{tag}_mean"], g[f"pca_{tag}_kept"], float(g[f"pca_{tag}_eps"]))
    assert loss == pytest.approx(float(g[f"pca_{tag}_loss"]), rel=1e-4, abs=1e-6)
    x = g.t(f"pca_{tag}_test").clone().requires_grad_(True)
    O.pca_loss(O.pca_format_singleview(x, [int(c) for c in cols[0]]), g.t(f"pca_{tag}_mean"), g.t(f"pca_{tag}_kept"),
               float(g[f"pca_{tag}_eps"])).backward()
    np.testing.assert_allclose(grad.reshape(s, -1), x.grad.numpy(), atol=1e-6, rtol=1e-3)


def test_pca_multiview_golden_and_grad(golden):
    g = golden("losses")
    mcm = g["pca_mv_mcm"].astype(np.int32)  # (views, J)
    index = np.ascontiguousarray(mcm.T)     # (J rows, V points)
    test = g["pca_mv_test"]
    s = test.shape[0]
    loss, grad = emu.pca(test.reshape(s, -1, 2), index, g["pca_mv_mean"], g["pca_mv_kept"], float(g["pca_mv_eps"]))
    assert loss == pytest.approx(float(g["pca_mv_loss"]), rel=1e-4, abs=1e-6)
    x = g.t("pca_mv_test").clone().requires_grad_(True)
    O.pca_loss(O.pca_format_multiview(x, [[int(c) for c in r] for r in mcm]), g.t("pca_mv_mean"), g.t("pca_mv_kept"),
               float(g["pca_mv_eps"])).backward()
    np.testing.assert_allclose(grad.reshape(s, -1), x.grad.numpy(), atol=1e-6, rtol=1e-3)


def test_rmse_golden(golden):
    g = golden("losses")
    assert emu.rmse(g["r_targ"], g["r_pred"]) == pytest.approx(float(g["rmse"]), rel=1e-5)


@pytest.mark.parametrize("n", [1, 3, 8, 11, 19])
def test_loss_combine_forward_backward_vs_autograd(stack_backend, n):
    """ops.loss_combine (lp_loss_combine / lp_loss_combine_bwd, LossFactory's weighted sum: reference losses/factory.py:229-285) against
    torch autograd of w * x and sum(a * w * x): a gradient flowing through weighted[i] as well as through the total, inputs that do and do
    not require gradients mixed, and more than 8 losses (the reference sums any number: groups of 8)"""
    from lightning_pose_amd import ops

    dev = stack_backend
    gen = torch.Generator().manual_seed(n)
    vals = torch.randn(n, generator=gen) * 3
    w = (torch.rand(n, generator=gen) + 0.1).tolist()
    a = [1.0 if i % 3 == 0 else 0.4 for i in range(n)]
    needs = [i % 4 != 1 for i in range(n)]
    xs = [vals[i].clone().to(dev).requires_grad_(needs[i]) for i in range(n)]
    weighted, total = ops.loss_combine(xs, w, a)
    ref = [vals[i].clone().requires_grad_(needs[i]) for i in range(n)]
    rw = torch.stack([w[i] * ref[i] for i in range(n)])
    rt = sum(a[i] * rw[i] for i in range(n))
    torch.testing.assert_close(weighted.detach().cpu(), rw.detach(), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(total.detach().cpu(), rt.detach(), rtol=1e-5, atol=1e-5)
    gsel = torch.randn(n, generator=gen)
    if any(needs):
        (2.5 * total + (weighted * gsel.to(dev)).sum()).backward()
        (2.5 * rt + (rw * gsel).sum()).backward()
    for i in range(n):
        if needs[i]:
            torch.testing.assert_close(xs[i].grad.cpu(), ref[i].grad, rtol=1e-5, atol=1e-6)
        else:
            assert xs[i].grad is None
