"""The reference's OWN hot-path test cases, restated against ``lightning_pose_amd`` (SURVEY.md section 8c lists them).

Each test names the reference test it mirrors (file::class::test under /root/reference/tests).  They call the product's public API -
the same class / function names and keyword arguments a user of ``lightning_pose`` would use - so they run on the emulated kernels in
the CPU suite and on the MI355X through ``liblp_hip.so`` under ``-m gpu``.  Cases that need the reference's toy dataset fixtures
(``heatmap_data_module`` ...) are rebuilt from literal tensors with the same structure.
"""

import math

import numpy as np
import pytest
import torch

STAGE = "train"


@pytest.fixture
def dev(stack_backend):
    return stack_backend


# ------------------------------------------------------------------------------------------------------------ losses/test_losses.py
def test_loss_base_class(dev):
    """TestLoss: weight = 1 / (2 e^log_weight), rectify_epsilon, reduce_loss, and the abstract methods raise"""
    from lightning_pose_amd.losses.losses import Loss

    assert torch.isclose(Loss(log_weight=0.0).weight, torch.tensor(0.5))
    assert torch.isclose(Loss(log_weight=1.0).weight, torch.tensor(1.0 / (2.0 * math.exp(1.0)), dtype=torch.float))
    r = Loss(epsilon=0.5).rectify_epsilon(torch.tensor([0.1, 0.5, 1.0]))
    assert r[0] == 0.0 and r[1] == 0.0 and torch.isclose(r[2], torch.tensor(0.5))
    vals = torch.tensor([1.0, 2.0, 3.0])
    assert Loss().reduce_loss(vals, method="mean") == 2.0 and Loss().reduce_loss(vals, method="sum") == 6.0
    for call in (Loss().remove_nans, Loss().compute_loss, Loss()):
        with pytest.raises(NotImplementedError):
            call()


def _check_logs(logs, loss, name, weight):
    assert logs[0]["name"] == f"{STAGE}_{name}_loss" and logs[0]["value"] == loss
    assert logs[1]["name"] == f"{name}_weight" and logs[1]["value"] == weight


def test_heatmap_mse_loss(dev):
    """TestHeatmapMSELoss: uniform maps equal -> exactly 0 (+ log entries); perturbed predictions -> positive"""
    from lightning_pose_amd.losses.losses import HeatmapMSELoss

    fn = HeatmapMSELoss()
    targets = (torch.ones((3, 7, 48, 48)) / (48 * 48)).to(dev)
    loss, logs = fn(heatmaps_targ=targets, heatmaps_pred=targets.clone(), stage=STAGE)
    assert loss.shape == torch.Size([]) and loss == 0.0
    _check_logs(logs, loss, "heatmap_mse", fn.weight)
    noisy = targets + 0.01 * torch.randn(targets.shape, generator=torch.Generator().manual_seed(0)).to(dev)
    assert fn(heatmaps_targ=targets, heatmaps_pred=noisy, stage=STAGE)[0] > 0.0


@pytest.mark.parametrize("name", ["heatmap_kl", "heatmap_js"])
def test_heatmap_divergence_losses(dev, name):
    """TestHeatmapKLLoss / TestHeatmapJSLoss: ~0 on equal maps made by generate_heatmaps, larger once predictions are rolled along the
    batch axis"""
    from lightning_pose_amd.data.heatmaps import generate_heatmaps
    from lightning_pose_amd.losses.losses import HeatmapJSLoss, HeatmapKLLoss

    fn = {"heatmap_kl": HeatmapKLLoss, "heatmap_js": HeatmapJSLoss}[name]()
    kp = 100 * torch.rand((3, 7, 2), generator=torch.Generator().manual_seed(1))
    maps = generate_heatmaps(kp.to(dev), height=100, width=100, output_shape=(32, 32))
    loss, logs = fn(heatmaps_targ=maps, heatmaps_pred=maps.clone(), stage=STAGE)
    assert loss.shape == torch.Size([]) and abs(float(loss)) < 1e-5
    _check_logs(logs, loss, name, fn.weight)
    loss2, _ = fn(heatmaps_targ=maps, heatmaps_pred=torch.roll(maps, shifts=1, dims=0), stage=STAGE)
    assert loss2 > loss


def test_temporal_loss_cases(dev):
    """TestTemporalLoss: constant -> 0; random -> > 0; analytic norms; epsilon rectification; confidence masking"""
    from lightning_pose_amd.losses.losses import TemporalLoss

    fn = TemporalLoss(epsilon=0.0)
    const = torch.ones(12, 32)
    const[:, 1], const[:, 2], const[:, 3] = 2, 4, 8
    loss, logs = fn(const.to(dev), stage=STAGE)
    assert loss.shape == torch.Size([]) and loss == 0.0
    _check_logs(logs, loss, "temporal", fn.weight)
    assert fn(torch.rand(12, 32, generator=torch.Generator().manual_seed(2)).to(dev), stage=STAGE)[0] > 0.0
    s2, s3 = math.sqrt(2.0), math.sqrt(3.0)
    two = torch.tensor([[0.0, 0.0], [s2, s2]])
    assert float(fn(two.to(dev), stage=STAGE)[0]) == pytest.approx(2.0, abs=1e-6)
    three = torch.tensor([[0.0, 0.0], [s2, s2], [s3 + s2, s3 + s2]])  # step norms 2 and sqrt(6); the loss is their mean
    assert float(TemporalLoss(epsilon=0.0)(three.to(dev), stage=STAGE)[0]) == pytest.approx((2 + math.sqrt(6)) / 2, abs=1e-6)
    assert float(TemporalLoss(epsilon=2.1)(three.to(dev), stage=STAGE)[0]) == pytest.approx((math.sqrt(6) - 2.1) / 2, abs=1e-6)
    # per-keypoint epsilon (rectify_epsilon on a (S-1, K) tensor)
    rep = torch.tensor([0.0, 1.0, 0.4]).unsqueeze(0).repeat(5, 1)
    r = TemporalLoss(epsilon=[0.1, 0.0, 0.5]).rectify_epsilon(rep)
    assert r.shape == (5, 3) and torch.all(r[:, 0] == 0) and torch.all(r[:, 1] == 1.0) and torch.all(r[:, 2] == 0)
    r = TemporalLoss(epsilon=[0.1, 0.0, 0.3]).rectify_epsilon(rep)
    assert torch.allclose(r[:, 2], torch.tensor([0.1]))
    fancy = torch.tensor([[1.0, 2.0, 1.5], [0.05, 0.12, 0.2]])
    r = TemporalLoss(epsilon=[0.1, 0.15, 0.3]).rectify_epsilon(fancy)
    assert torch.allclose(r[0], torch.tensor([0.9, 1.85, 1.2])) and torch.allclose(r[1], torch.zeros(3))
    # ... and the same per-keypoint epsilons through the fused kernel: 3 keypoints moving 1.0 / 2.0 / 1.5 px per frame
    kp = torch.zeros(3, 6)
    kp[:, 0::2] = torch.arange(3.0).unsqueeze(1) * torch.tensor([1.0, 2.0, 1.5])
    got = TemporalLoss(epsilon=[0.1, 0.15, 0.3])(kp.to(dev), stage=STAGE)[0]
    assert float(got) == pytest.approx((0.9 + 1.85 + 1.2) / 3, abs=1e-6)
    # confidences below the threshold switch a keypoint's differences off
    masked_fn = TemporalLoss(epsilon=0.0, prob_threshold=0.5)
    kp = torch.zeros(4, 4)
    kp[1::2] = 1.0
    conf = torch.zeros(4, 2)
    conf[:, 1] = 1.0
    with_conf, _ = masked_fn(kp.to(dev), confidences=conf.to(dev), stage=STAGE)
    without, _ = masked_fn(kp.to(dev), stage=STAGE)
    assert with_conf.shape == torch.Size([]) and 0.0 <= float(with_conf) <= float(without) + 1e-6
    assert float(with_conf) == pytest.approx(float(without) / 2, abs=1e-6)  # exactly one of the two keypoints is left


def test_rmse_loss(dev):
    """TestRegressionRMSELoss: equal -> 0; targets 2 vs predictions 0 -> exactly 2; NaN targets are skipped"""
    from lightning_pose_amd.losses.losses import RegressionRMSELoss

    fn = RegressionRMSELoss()
    same = torch.ones(4, 10).to(dev)
    loss, logs = fn(same, same.clone(), stage=STAGE)
    assert loss.shape == torch.Size([]) and loss == 0.0
    _check_logs(logs, loss, "rmse", fn.weight)
    assert fn((2 * torch.ones(5, 4)).to(dev), torch.zeros(5, 4).to(dev), stage=STAGE)[0] == 2.0
    targ = 2 * torch.ones(5, 4)
    targ[0, :2] = float("nan")  # one unlabeled keypoint: the mean runs over the remaining nine
    assert fn(targ.to(dev), torch.zeros(5, 4).to(dev), stage=STAGE)[0] == 2.0


def test_pca_loss_constructor_errors_and_subspace(dev):
    """TestPCALoss: missing mirrored_column_matches / unknown loss name -> ValueError; data inside the kept subspace -> exactly 0"""
    from lightning_pose_amd.losses.losses import PCALoss

    data = np.random.default_rng(0).normal(size=(40, 8)).astype(np.float32)
    with pytest.raises(ValueError):
        PCALoss(loss_name="pca_multiview", data_arr=data, device=dev)
    with pytest.raises(ValueError):
        PCALoss(loss_name="pca_everything", data_arr=data, device=dev)
    fn = PCALoss(loss_name="pca_multiview", components_to_keep=3, mirrored_column_matches=[[0, 1], [2, 3]], data_arr=data, device=dev)
    kept = torch.eye(4)[:, :3].T  # (a transposed VIEW, as in the reference's test: the op must not assume dense operands)
    obs = torch.randn(10, 3, generator=torch.Generator().manual_seed(3)) @ kept  # (10, 4) = one 2-view keypoint, inside span(kept)
    fn.pca.parameters["kept_eigenvectors"] = kept.to(dev)
    fn.pca.parameters["mean"] = obs.mean(0).to(dev)
    fn.pca.mirrored_column_matches = [[0], [1]]
    fn._index = None
    fn.epsilon = torch.tensor(0.0)
    loss, logs = fn(obs.to(dev), stage=STAGE)
    # (exactly 0 in the reference's CPU test; the device contracts the two skinny products with fused multiply-adds)
    assert float(loss) == pytest.approx(0.0, abs=2e-5), float(loss)
    assert logs[0]["name"] == f"{STAGE}_pca_multiview_loss"
    # the diagnostic helpers of KeypointPCA (reproject / compute_reprojection_error) describe the same quantity the kernel reduces:
    # loss = mean(relu(error - epsilon)) over (sample, keypoint)
    rand = (torch.randn(10, 4, generator=torch.Generator().manual_seed(8)) * 5).to(dev)
    err = fn.pca.compute_reprojection_error(rand)
    assert err.shape == (10, 2) and torch.allclose(fn.pca.reproject(fn.pca.reproject(rand)), fn.pca.reproject(rand), atol=1e-5)
    fn.epsilon = torch.tensor(0.5)
    assert float(fn(rand, stage=STAGE)[0]) == pytest.approx(float(torch.relu(err - 0.5).mean()), rel=1e-4, abs=1e-5)


# ---------------------------------------------------------------------------------------------------------- losses/test_factory.py
def test_loss_factory_cases(dev):
    """TestLossFactory: zero on equal maps; heat-map losses ignore anneal_weight, others scale with it; None == 1; stage may be omitted"""
    from lightning_pose_amd.losses.factory import LossFactory, get_loss_classes
    from lightning_pose_amd.losses.losses import Loss

    assert all(issubclass(c, Loss) for c in get_loss_classes().values())
    maps = (torch.ones(2, 3, 16, 16) / 256).to(dev)
    fac = LossFactory({"heatmap_mse": {"log_weight": 0.0}}, None)
    tot, logs = fac(stage=STAGE, heatmaps_targ=maps, heatmaps_pred=maps.clone())
    assert tot == 0.0 and {d["name"] for d in logs} == {"train_heatmap_mse_loss", "heatmap_mse_weight", "train_heatmap_mse_loss_weighted"}
    noisy = maps + 0.01 * torch.randn(maps.shape, generator=torch.Generator().manual_seed(4)).to(dev)
    a, _ = fac(stage=STAGE, anneal_weight=1.0, heatmaps_targ=maps, heatmaps_pred=noisy)
    b, _ = fac(stage=STAGE, anneal_weight=0.25, heatmaps_targ=maps, heatmaps_pred=noisy)
    assert a > 0 and torch.isclose(a, b)
    tfac = LossFactory({"temporal": {"log_weight": 0.0, "epsilon": 0.0}}, None)
    kp = torch.rand(6, 8, generator=torch.Generator().manual_seed(5)).to(dev)
    full, _ = tfac(stage=STAGE, anneal_weight=1.0, keypoints_pred=kp)
    half, _ = tfac(stage=STAGE, anneal_weight=0.5, keypoints_pred=kp)
    none, _ = tfac(stage=STAGE, anneal_weight=None, keypoints_pred=kp)
    assert torch.isclose(half, 0.5 * full) and torch.isclose(none, full)
    nostage, logs = tfac(keypoints_pred=kp)
    assert torch.isclose(nostage, full) and logs[0]["name"] == "None_temporal_loss"


# ------------------------------------------------------------------------------------------------------------ data/test_heatmaps.py
KP_MIXED = torch.tensor([
    [[32.0, 32.0], [-10.0, 50.0], [500.0, 32.0], [32.0, 500.0]],          # valid, x < -1, x > W + 1, y > H + 1 (heat-map px)
    [[32.0, -10.0], [64.0, 64.0], [float("nan"), 32.0], [128.0, 128.0]],  # y < -1, valid, NaN, valid
])


def _gen(dev, kp, vis=None):
    from lightning_pose_amd.data.heatmaps import generate_heatmaps

    v = None if vis is None else vis.to(dev)
    return generate_heatmaps(kp.to(dev), height=256, width=256, output_shape=(64, 64), visibility=v).cpu()


def test_generate_heatmaps_oob_nan_and_visibility(dev):
    """TestGenerateHeatmaps::test_out_of_bounds_nan_indices + the visibility_* cases"""
    zeros, uniform = torch.zeros(64, 64), torch.ones(64, 64) / (64 * 64)
    h = _gen(dev, KP_MIXED)
    for b, k in ((0, 1), (0, 2), (0, 3), (1, 0), (1, 2)):
        assert torch.allclose(h[b, k], zeros), (b, k)
    for b, k in ((0, 0), (1, 1), (1, 3)):
        assert not torch.allclose(h[b, k], zeros) and float(h[b, k].sum()) == pytest.approx(1.0, abs=1e-5)
    h1 = _gen(dev, KP_MIXED, torch.ones(2, 4, dtype=torch.long))       # occluded: uniform, whatever the coordinates are
    assert torch.allclose(h1[0, 1], uniform) and torch.allclose(h1[1, 2], uniform) and torch.allclose(h1[0, 0], uniform)
    h0 = _gen(dev, KP_MIXED, torch.zeros(2, 4, dtype=torch.long))      # not labeled: zeros, valid coordinates included
    assert torch.allclose(h0[0, 0], zeros) and torch.allclose(h0[1, 1], zeros)
    h2 = _gen(dev, KP_MIXED, torch.full((2, 4), 2, dtype=torch.long))  # visible: Gaussian, but OOB / NaN fall back to zeros
    assert torch.allclose(h2[0, 1], zeros) and torch.allclose(h2[1, 2], zeros)
    assert torch.allclose(h2[0, 0], h[0, 0]) and torch.allclose(h2[1, 1], h[1, 1])
    mixed = torch.tensor([[2, 1, 0, 2], [0, 2, 1, 2]])
    hm = _gen(dev, KP_MIXED, mixed)
    assert torch.allclose(hm[0, 0], h[0, 0]) and torch.allclose(hm[0, 1], uniform) and torch.allclose(hm[0, 2], zeros)
    assert torch.allclose(hm[0, 3], zeros) and torch.allclose(hm[1, 0], zeros) and torch.allclose(hm[1, 1], h[1, 1])
    assert torch.allclose(hm[1, 2], uniform) and torch.allclose(hm[1, 3], h[1, 3])
    # the Gaussian peaks where the keypoint lands on the heat-map grid: (32, 32) px of 256 -> (8, 8) of 64
    assert int(h[0, 0].argmax()) == 8 * 64 + 8


def test_generate_heatmaps_extreme_coordinates_stay_finite(dev):
    """TestGenerateHeatmaps::test_extreme_keypoint_clamping (value part)"""
    big = 1e8
    kp = torch.tensor([[[-big, 32.0], [big, 32.0], [32.0, -big], [32.0, big]]])
    h = _gen(dev, kp)
    assert torch.isfinite(h).all() and h.shape == (1, 4, 64, 64) and not h.any()
    assert torch.allclose(_gen(dev, kp, torch.ones(1, 4, dtype=torch.long))[0, 0], torch.ones(64, 64) / 4096)
    assert not _gen(dev, kp, torch.full((1, 4), 2, dtype=torch.long)).any()


def test_generate_heatmaps_keep_gradients(dev):
    """TestGenerateHeatmaps::test_keep_gradients: detached by default; with keep_gradients=True the keypoints stay attached (lp_heatmap_gen_bwd)
    and the gradient equals autograd's through the restated reference expression - including the maps that carry none (NaN, out of bounds,
    visibility 0 / 1)."""
    from lightning_pose_amd.data.heatmaps import generate_heatmaps

    kp = torch.tensor([[[32.0, 64.0], [128.0, 96.0]]], device=dev, requires_grad=True)
    h = generate_heatmaps(kp, height=256, width=256, output_shape=(64, 64), keep_gradients=False)
    assert not h.requires_grad and kp.grad is None
    gen = torch.Generator().manual_seed(3)
    base = torch.tensor([[[32.0, 64.0], [128.0, 96.0], [200.0, 150.0], [100.0, 200.0]], [[64.0, 32.0], [160.0, 120.0], [3.5, 250.2], [float("nan"), 5.0]],
                         [[-3.0, 10.0], [300.0, 40.0], [255.9, 0.1], [17.3, 41.9]]])
    weights = torch.randn(3, 4, 48, 64, generator=gen)   # a loss that is not invariant to the map (sum(H) = 1 has zero gradient)
    for vis in (None, torch.tensor([[2, 2, 1, 2], [2, 0, 2, 2], [2, 2, 2, 2]])):
        a = base.clone().to(dev).requires_grad_(True)
        b = base.clone().requires_grad_(True)
        got = generate_heatmaps(a, height=192, width=256, output_shape=(48, 64), keep_gradients=True, visibility=None if vis is None else vis.to(dev))
        from oracle import restated as O
        want = O.generate_heatmaps(b, 192, 256, (48, 64), 1.25, vis, keep_gradients=True)
        torch.testing.assert_close(got.detach().cpu(), want.detach(), atol=2e-7, rtol=1e-5)
        (got * weights.to(dev)).sum().backward()
        (want * weights).sum().backward()
        assert a.grad is not None and torch.isfinite(a.grad).all()
        torch.testing.assert_close(a.grad.cpu(), torch.nan_to_num(b.grad), atol=2e-6, rtol=2e-4)
        assert float(a.grad[1, 3].abs().sum()) == 0 and float(a.grad[2, 1].abs().sum()) == 0   # NaN / out of bounds: constant maps


# --------------------------------------------------------------------------------------------------------------- data/test_utils.py
def _affine(kp, tf):
    """(S, K, 2) keypoints through (2, 3) or (S, 2, 3) matrices"""
    if tf.dim() == 2:
        return kp @ tf[:, :2].T + tf[:, -1]
    return torch.bmm(kp, tf[:, :, :2].transpose(2, 1)) + tf[:, :, -1].unsqueeze(1)


def test_undo_affine_transform_batch_cases(dev):
    """test_undo_affine_transform_batch: one matrix, one per frame, one shared by 3 views, one per view; and the (1,) sentinel"""
    from lightning_pose_amd.data.utils import undo_affine_transform_batch

    S, K = 5, 6
    g = torch.Generator().manual_seed(0)
    kp, tf = torch.randn(S, K, 2, generator=g), torch.randn(2, 3, generator=g)
    out = undo_affine_transform_batch(_affine(kp, tf).reshape(S, -1).to(dev), tf.to(dev), is_multiview=False)
    assert torch.allclose(kp.reshape(S, -1), out.cpu(), atol=1e-4)
    kp, tfs = torch.randn(S, K, 2, generator=g), torch.randn(S, 2, 3, generator=g)
    out = undo_affine_transform_batch(_affine(kp, tfs).reshape(S, -1).to(dev), tfs.to(dev), is_multiview=False)
    assert torch.allclose(kp.reshape(S, -1), out.cpu(), atol=1e-4)
    V = 3
    kp, tf = torch.randn(S, K * V, 2, generator=g), torch.randn(2, 3, generator=g)
    out = undo_affine_transform_batch(_affine(kp, tf).reshape(S, -1).to(dev), tf.repeat(V, 1, 1).to(dev), is_multiview=True)
    assert torch.allclose(kp.reshape(S, -1), out.cpu(), atol=1e-4)
    kps, augs, tfv = [], [], []
    for _ in range(V):
        kv, tv = torch.randn(S, K, 2, generator=g), torch.randn(2, 3, generator=g)
        kps.append(kv.reshape(S, -1))
        augs.append(_affine(kv, tv).reshape(S, -1))
        tfv.append(tv)
    out = undo_affine_transform_batch(torch.cat(augs, -1).to(dev), torch.stack(tfv).to(dev), is_multiview=True)
    assert torch.allclose(torch.cat(kps, -1), out.cpu(), atol=1e-4)
    plain = torch.randn(S, 2 * K, generator=g).to(dev)
    assert undo_affine_transform_batch(plain, torch.tensor([-1.0]).to(dev)) is plain  # "no augmentation": the input itself
    # test_undo_affine_transform: the (S, K, 2) form, one matrix and one per frame
    from lightning_pose_amd.data.utils import undo_affine_transform

    kp = torch.randn(S, K, 2, generator=g)
    for tf in (torch.randn(2, 3, generator=g), torch.randn(S, 2, 3, generator=g)):
        assert torch.allclose(kp, undo_affine_transform(_affine(kp, tf).to(dev), tf.to(dev)).cpu(), atol=1e-4)


# -------------------------------------------------------------------------------------------------------------- data/test_bboxes.py
def test_model_to_frame_batch_cases(dev):
    """TestModelToFrameBatch: crop consistency (single view), per-view boxes, context rows, 'frames' batches, view inference, mixed views"""
    from lightning_pose_amd.data.bboxes import model_to_frame_batch

    d = lambda t: t.to(dev)  # noqa: E731
    # single view: cropping the image by (25, 40) px and shrinking the box accordingly leaves the frame coordinates unchanged
    g = torch.Generator().manual_seed(6)
    kp = torch.rand(4, 10, generator=g) * 100 + 60
    kp[1, 2:4] = float("nan")
    bbox = torch.tensor([[0.0, 0.0, 406.0, 396.0]]).repeat(4, 1)
    full = model_to_frame_batch({"images": torch.zeros(4, 3, 256, 256), "bbox": d(bbox), "keypoints": d(kp)}, d(kp.clone()))
    xc, yc = 25, 40
    xp, yp = xc * bbox[:, 3] / 256, yc * bbox[:, 2] / 256
    small = bbox.clone()
    small[:, 0] += xp
    small[:, 1] += yp
    small[:, 2] -= 2 * yp
    small[:, 3] -= 2 * xp
    kp2 = kp.clone()
    kp2[:, 0::2] -= xc
    kp2[:, 1::2] -= yc
    crop = model_to_frame_batch({"images": torch.zeros(4, 3, 256 - 2 * yc, 256 - 2 * xc), "bbox": d(small)}, d(kp2))
    assert torch.allclose(full.cpu(), crop.cpu(), equal_nan=True, atol=1e-3)
    # multiview: each view's keypoints go through that view's [x, y, h, w]
    kp = torch.tensor([[0.0, 0.0, 0.0, 0.0], [10.0, 10.0, 10.0, 10.0]])
    bb = torch.tensor([[5.0, 6.0, 100.0, 101.0, 10.0, 11.0, 102.0, 103.0], [0.0, 0.0, 123.0, 124.0, 0.0, 0.0, 3.0, 4.0]])
    out = model_to_frame_batch({"images": torch.zeros(2, 2, 3, 10, 10), "bbox": d(bb), "num_views": torch.tensor([2, 2])}, d(kp)).cpu()
    assert out[0].tolist() == [5.0, 6.0, 10.0, 11.0] and out[1].tolist() == [124.0, 123.0, 4.0, 3.0]
    # context batch: the two edge rows on either side of the bbox tensor are skipped
    edge = torch.tensor([1.0, 2.0, 100.0, 101.0, 10.0, 11.0, 102.0, 103.0])
    bb6 = torch.stack([edge, edge, bb[0], bb[1], edge, edge])
    out = model_to_frame_batch({"images": torch.zeros(2, 2, 3, 10, 10), "bbox": d(bb6), "num_views": torch.full((6,), 2)}, d(kp)).cpu()
    assert out[0].tolist() == [5.0, 6.0, 10.0, 11.0] and out[1].tolist() == [124.0, 123.0, 4.0, 3.0]
    # unlabeled dict ('frames'), single view: (0, 0) -> box corner, (w, h) -> opposite corner
    kp1 = torch.tensor([[0.0, 0.0], [10.0, 10.0]])
    out = model_to_frame_batch({"frames": torch.zeros(2, 3, 10, 10), "bbox": d(bb[:, :4])}, d(kp1)).cpu()
    assert out[0].tolist() == [5.0, 6.0] and out[1].tolist() == [124.0, 123.0]
    # is_multiview: the number of views comes from the bbox width
    out = model_to_frame_batch({"frames": torch.zeros(2, 3, 10, 10), "bbox": d(bb), "is_multiview": True}, d(kp)).cpu()
    assert out[0].tolist() == [5.0, 6.0, 10.0, 11.0] and out[1].tolist() == [124.0, 123.0, 4.0, 3.0]
    with pytest.raises(ValueError):
        model_to_frame_batch({"images": torch.zeros(2, 2, 3, 10, 10), "bbox": d(bb), "num_views": torch.tensor([16, 2])}, d(kp))
    # the argument is not written through (documented deviation: the reference's in_place=True default does)
    assert kp.tolist() == [[0.0, 0.0, 0.0, 0.0], [10.0, 10.0, 10.0, 10.0]]


# ----------------------------------------------------------------------------------------------------- models/heads/test_heatmap.py
def test_subpixmaxima_known_answers(dev):
    """TestRunSubpixelmaxima: a delta at (2, 2) / (4, 4) of a 6x6 / 9x9-ish map decodes to 2^ds times that location with confidence 1;
    lower temperatures blur it (the reference's T = 100 / T = 10 expectations)"""
    from lightning_pose_amd import ops

    maps = torch.zeros(1, 2, 16, 16)
    maps[0, 0, 2, 2] = 1.0
    maps[0, 1, 4, 4] = 1.0
    for ds, want in ((1, [4.0, 4.0, 8.0, 8.0]), (2, [8.0, 8.0, 16.0, 16.0])):
        kp, _, conf = ops.decode(maps.to(dev), ds, 1000.0, ops.DecodeFrameMap(None, False, None, 1, 1, 1, 2))
        assert torch.allclose(kp.reshape(-1).cpu(), torch.tensor(want), atol=1e-4)
        assert torch.allclose(conf.reshape(-1).cpu(), torch.ones(2), atol=1e-5)
    _, _, c100 = ops.decode(maps.to(dev), 2, 100.0, ops.DecodeFrameMap(None, False, None, 1, 1, 1, 2))
    assert torch.allclose(c100.cpu(), torch.ones(1, 2), rtol=1e-3) and (c100.cpu() != 1.0).all()
    k10, _, c10 = ops.decode(maps.to(dev), 2, 10.0, ops.DecodeFrameMap(None, False, None, 1, 1, 1, 2))
    assert (k10.reshape(-1).cpu()[2:] != 16.0).all() and (c10.cpu() < 0.5).all()


# --------------------------------------------------------------------------------------------------------------- utils/test_pca.py
def test_pca_helper_classes_reference_cases():
    """TestEmpiricalEpsilon, TestComponentChooser, TestFormatMultiviewDataForPca, test_convert_dict_values_to_tensors and TestNaNPCA
    (host-side fit helpers of the PCA loss; scikit-learn's own PCA on the diabetes data is the yardstick, as in the reference)"""
    from sklearn.datasets import load_diabetes
    from sklearn.decomposition import PCA

    from lightning_pose_amd.utils.pca import (ComponentChooser, EmpiricalEpsilon, NaNPCA, convert_dict_values_to_tensors,
                                              format_multiview_data_for_pca)

    ramp = np.arange(101, dtype="float")
    assert EmpiricalEpsilon(percentile=90)(ramp) == 90 and EmpiricalEpsilon(percentile=90)(torch.tensor(ramp)) == 90
    holes = ramp.copy()
    holes[1::2] = np.nan
    assert EmpiricalEpsilon(percentile=90)(holes) == 90

    data = load_diabetes().data
    skl = PCA(svd_solver="full").fit(data)
    assert ComponentChooser(skl, 4)() == 4 and ComponentChooser(skl, 2)() < ComponentChooser(skl, 3)()
    with pytest.raises(ValueError):
        ComponentChooser(skl, 11)          # only 10 observation dimensions
    n95 = ComponentChooser(skl, 0.95)()
    assert 0 < n95 <= 10 and ComponentChooser(skl, 1.0)() == 10 and ComponentChooser(skl, 0.20)() < ComponentChooser(skl, 0.90)()
    for bad in (1.04, -0.2):
        with pytest.raises(ValueError):
            ComponentChooser(skl, bad)

    kp = torch.rand(12, 20, 2)
    for matches in ([[0, 1, 2, 3], [4, 5, 6, 7]], [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11]]):
        arr = format_multiview_data_for_pca(kp, matches)
        assert arr.shape == torch.Size([12 * 4, 2 * len(matches)])
    with pytest.raises(AssertionError):
        format_multiview_data_for_pca(kp, [[0, 1, 2, 3], [4, 5, 6]])
    conv = convert_dict_values_to_tensors({"a": 4.0, "b": 10.1, "c": 4}, device="cpu")
    assert all(isinstance(v, torch.Tensor) and v.dtype == torch.float32 for v in conv.values())

    # NaNPCA == sklearn on complete data ...
    mine = NaNPCA().fit(data)
    z = mine.transform(data)
    assert data.shape == (442, 10) and skl.noise_variance_ == mine.noise_variance_ and skl.n_samples_ == mine.n_samples_
    assert skl.n_components_ == mine.n_components_
    for attr in ("components_", "explained_variance_", "explained_variance_ratio_", "singular_values_"):
        assert np.allclose(getattr(skl, attr), getattr(mine, attr), rtol=1e-10), attr
    assert np.allclose(skl.transform(data), z, rtol=1e-10)
    # ... and degrades gracefully with missing entries: one NaN, a staircase of NaNs, a fully missing row
    one = data.copy()
    one[0, 0] = np.nan
    p1 = NaNPCA().fit(one)
    big = np.abs(mine.components_) > 0.05
    assert not np.allclose(mine.components_[big], p1.components_[big], rtol=1e-10)
    assert np.allclose(mine.components_[big], p1.components_[big], rtol=1e-1)
    for attr in ("explained_variance_", "explained_variance_ratio_", "singular_values_"):
        assert not np.allclose(getattr(mine, attr), getattr(p1, attr), rtol=1e-10) and np.allclose(getattr(mine, attr), getattr(p1, attr), rtol=1e-2)
    assert np.allclose(z[1:], p1.transform(one)[1:], atol=1e-2)
    rows, cols = list(range(13)), [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 8, 7, 6]
    many = data.copy()
    many[rows, cols] = np.nan
    p2 = NaNPCA().fit(many)
    assert np.allclose(p1.explained_variance_[:7], p2.explained_variance_[:7], rtol=1e-2)
    assert np.allclose(p1.singular_values_[:7], p2.singular_values_[:7], rtol=1e-2)
    assert np.allclose(z[13:], p2.transform(many)[13:], atol=1e-2)
    gone = data.copy()
    gone[0, :] = np.nan
    zg = NaNPCA().fit(gone).transform(gone)
    assert np.allclose(z[1:], zg[1:], atol=1e-2) and np.all(zg[0] == 0)
