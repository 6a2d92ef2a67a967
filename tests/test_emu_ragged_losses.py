"""Loss and decode kernels on randomly drawn, deliberately awkward shapes (odd heat-map sizes, one-pixel tails behind the 16-byte paths,
a single keypoint, the shortest legal window, masked rows, every map invalid but one) against the oracle's torch restatement with autograd
gradients - the golden vectors pin these kernels at the reference's own shapes, this file pins their index arithmetic everywhere else.
Runs on the CPU emulator build of the kernel sources and, under -m gpu, on the device."""

import numpy as np
import pytest
import torch

from lightning_pose_amd import _lib
from oracle import restated as O
from tests.hipemu import emu

pytestmark = pytest.mark.usefixtures("kernel_backend")


def _cases(seed, n, draw):
    r = np.random.default_rng(seed)
    return [draw(r) for _ in range(n)]


HM = _cases(11, 8, lambda r: (int(r.integers(1, 5)), int(r.integers(1, 6)), int(r.integers(3, 23)), int(r.integers(3, 23))))


@pytest.mark.parametrize("b,k,h,w", HM)
def test_heatmap_losses_ragged(b, k, h, w):
    gen = torch.Generator().manual_seed(b * 1000 + k * 100 + h * 10 + w)
    targ = torch.softmax(torch.randn(b, k, h * w, generator=gen) * 3, -1).reshape(b, k, h, w)
    drop = torch.rand(b, k, generator=gen) < 0.35          # unlabeled keypoints: all-zero target maps are left out of the mean
    if bool(drop.all()):
        drop[0, 0] = False
    targ[drop] = 0.0
    pred0 = torch.softmax(torch.randn(b, k, h * w, generator=gen), -1).reshape(b, k, h, w)
    for kind, fn in ((_lib.HM_MSE, O.heatmap_mse_loss), (_lib.HM_KL, O.heatmap_kl_loss), (_lib.HM_JS, O.heatmap_js_loss)):
        pred = pred0.clone().requires_grad_(True)
        want = fn(targ, pred)
        want.backward()
        loss, grad = emu.heatmap_div(kind, targ.numpy(), pred0.numpy())
        assert loss == pytest.approx(float(want.detach()), rel=2e-5, abs=1e-9), kind
        np.testing.assert_allclose(grad, pred.grad.numpy(), atol=2e-6 * float(pred.grad.abs().max()) + 1e-12, rtol=2e-4)


@pytest.mark.parametrize("s,k,h,w", _cases(12, 6, lambda r: (int(r.integers(1, 5)), int(r.integers(1, 5)), int(r.integers(4, 21)), int(r.integers(4, 21)))))
def test_unimodal_mse_ragged(s, k, h, w):
    gen = torch.Generator().manual_seed(s * 1000 + k * 100 + h * 10 + w)
    img_h, img_w = 4 * h, 4 * w
    pred0 = torch.softmax(torch.randn(s, k, h * w, generator=gen) * 2, -1).reshape(s, k, h, w)
    kp = torch.rand(s, k, 2, generator=gen) * torch.tensor([img_w, img_h], dtype=torch.float32)
    kp[0, 0, 0] = float("nan") if s * k > 1 else kp[0, 0, 0]      # an undefined prediction drops its map
    conf = torch.rand(s, k, generator=gen)
    thr = 0.4
    if not bool(((conf >= thr) & ~torch.isnan(kp[..., 0])).any()):
        conf[-1, -1] = 0.9
    pred = pred0.clone().requires_grad_(True)
    want = O.unimodal_mse_loss(kp.reshape(s, 2 * k), pred, conf, img_h, img_w, prob_threshold=thr)
    want.backward()
    loss, grad = emu.unimodal_mse(kp.numpy(), pred0.numpy(), conf.numpy(), img_h, img_w, thr)
    assert loss == pytest.approx(float(want.detach()), rel=2e-5, abs=1e-9)
    np.testing.assert_allclose(grad, pred.grad.numpy(), atol=2e-6 * float(pred.grad.abs().max()) + 1e-12, rtol=2e-4)


@pytest.mark.parametrize("s,k", [(2, 1), (2, 7), (3, 17), (9, 2), (33, 5), (5, 64)])
def test_temporal_ragged(s, k):
    gen = torch.Generator().manual_seed(s * 100 + k)
    kp0 = torch.rand(s, 2 * k, generator=gen) * 50
    conf = torch.rand(s, k, generator=gen)
    eps = torch.rand(k, generator=gen) * 8          # per-keypoint epsilon (reference: rectify_epsilon on a (S - 1, K) tensor)
    for c, thr in ((None, 0.0), (conf, 0.3)):
        kp = kp0.clone().requires_grad_(True)
        want = O.temporal_loss(kp, c, eps, thr)
        want.backward()
        loss, grad = emu.temporal(kp0.reshape(s, k, 2).numpy(), None if c is None else c.numpy(), eps.numpy(), thr)
        assert loss == pytest.approx(float(want.detach()), rel=1e-5, abs=1e-7)
        np.testing.assert_allclose(grad.reshape(s, 2 * k), kp.grad.numpy(), atol=1e-6, rtol=1e-4)


@pytest.mark.parametrize("s,k,pts,ncomp", [(1, 3, 3, 1), (4, 9, 5, 4), (7, 17, 17, 10), (3, 64, 64, 100), (2, 4, 2, 4)])
def test_pca_singleview_ragged(s, k, pts, ncomp):
    gen = torch.Generator().manual_seed(s * 100 + k)
    kp0 = torch.rand(s, 2 * k, generator=gen) * 40
    cols = sorted(torch.randperm(k, generator=gen)[:pts].tolist())
    mean = torch.rand(2 * pts, generator=gen) * 40
    kept = torch.linalg.qr(torch.randn(2 * pts, 2 * pts, generator=gen))[0][:ncomp].contiguous()   # orthonormal rows, as a fitted PCA has
    eps = 2.0
    kp = kp0.clone().requires_grad_(True)
    want = O.pca_loss(O.pca_format_singleview(kp, cols), mean, kept, eps)
    want.backward()
    loss, grad = emu.pca(kp0.reshape(s, k, 2).numpy(), np.asarray([cols], np.int32), mean.numpy(), kept.numpy(), eps)
    assert loss == pytest.approx(float(want.detach()), rel=2e-5, abs=1e-6)
    np.testing.assert_allclose(grad.reshape(s, 2 * k), kp.grad.numpy(), atol=2e-6, rtol=2e-4)


@pytest.mark.parametrize("s,views,kv,matched", [(1, 2, 3, 2), (5, 3, 4, 4), (3, 4, 17, 9)])
def test_pca_multiview_ragged(s, views, kv, matched):
    gen = torch.Generator().manual_seed(s * 100 + views * 10 + kv)
    k = views * kv
    kp0 = torch.rand(s, 2 * k, generator=gen) * 40
    pick = sorted(torch.randperm(kv, generator=gen)[:matched].tolist())
    mcm = [[v * kv + j for j in pick] for v in range(views)]           # keypoint j of every view is the same body part
    mean = torch.rand(2 * views, generator=gen) * 40
    kept = torch.linalg.qr(torch.randn(2 * views, 2 * views, generator=gen))[0][:3].contiguous()
    kp = kp0.clone().requires_grad_(True)
    want = O.pca_loss(O.pca_format_multiview(kp, mcm), mean, kept, 0.5)
    want.backward()
    index = np.asarray(mcm, np.int32).T.copy()                         # (matched rows, views): one PCA sample per body part
    loss, grad = emu.pca(kp0.reshape(s, k, 2).numpy(), index, mean.numpy(), kept.numpy(), 0.5)
    assert loss == pytest.approx(float(want.detach()), rel=2e-5, abs=1e-6)
    np.testing.assert_allclose(grad.reshape(s, 2 * k), kp.grad.numpy(), atol=2e-6, rtol=2e-4)


@pytest.mark.parametrize("n,k", [(1, 1), (3, 17), (40, 2)])
def test_rmse_ragged(n, k):
    gen = torch.Generator().manual_seed(n * 10 + k)
    t = torch.rand(n, 2 * k, generator=gen) * 100
    p = t + torch.randn(n, 2 * k, generator=gen)
    miss = torch.rand(n, k, generator=gen) < 0.3
    if bool(miss.all()):
        miss[0, 0] = False
    t.reshape(n, k, 2)[miss] = float("nan")                             # unlabeled keypoints: NaN pairs
    assert emu.rmse(t.numpy(), p.numpy()) == pytest.approx(float(O.rmse_loss(t, p)), rel=1e-5)


DECODE_BWD = _cases(13, 6, lambda r: (int(r.choice([1, 2, 3])), int(r.integers(11, 24)), int(r.integers(11, 34)), int(r.integers(1, 3)), int(r.integers(1, 4))))   # (axes >= 11 px: _tables.decode_window)


@pytest.mark.parametrize("ds,h,w,b,k", DECODE_BWD)
def test_decode_backward_ragged(ds, h, w, b, k):
    gen = torch.Generator().manual_seed(ds * 1000 + h * 30 + w)
    heat = torch.softmax(3 * torch.randn(b, k, h * w, generator=gen), -1).reshape(b, k, h, w).requires_grad_(True)
    kp, _ = O.soft_argmax(heat, ds, 1000.0)
    gk = torch.randn(kp.shape, generator=gen)
    (kp * gk).sum().backward()
    _, _, _, stats = emu.decode_fwd(heat.detach().numpy(), ds)
    g_heat = emu.decode_bwd(heat.detach().numpy(), ds, stats, g_aug=gk.reshape(b, k, 2).numpy())
    ref = heat.grad.numpy()
    np.testing.assert_allclose(g_heat, ref, atol=2e-3 * np.abs(ref).max(), rtol=0)
