"""The reference's own known-answer tests for the hot path, run against oracle.restated.

Each test names the reference test it restates (paths relative to the reference tree).  CPU only.
"""

import math

import numpy as np
import pytest
import torch

from oracle import restated as O


def _delta_maps():
    x = torch.zeros(1, 2, 8, 8)
    x[0, 0, 2, 2] = 1.0
    x[0, 1, 4, 4] = 1.0
    return x


@pytest.mark.parametrize("ds", [1, 2])
def test_subpixmax_delta_scales_by_2_pow_ds(ds):
    """tests/models/heads/test_heatmap.py:124-170 (TestRunSubpixelMaxima): delta at (2,2),(4,4)."""
    kp, conf = O.soft_argmax(_delta_maps(), ds, 1000.0)
    want = torch.tensor([[2.0, 2.0, 4.0, 4.0]]) * 2 ** ds
    assert torch.allclose(kp, want, atol=1e-4)
    assert torch.allclose(conf, torch.ones(1, 2), atol=1e-6)


def test_subpixmax_temperature_dependence():
    """tests/models/heads/test_heatmap.py:172-219: T=100 conf ~ 1 (rtol 1e-3); T=10 spreads mass."""
    kp, conf = O.soft_argmax(_delta_maps(), 2, 100.0)
    assert torch.allclose(conf, torch.ones(1, 2), rtol=1e-3)
    kp10, conf10 = O.soft_argmax(_delta_maps(), 2, 10.0)
    assert not torch.allclose(kp10[0, 2:], torch.tensor([16.0, 16.0]), atol=1e-2)
    assert (conf10 < 0.5).all()


def test_upsample_shape_and_peak():
    """tests/models/heads/test_heatmap.py:84-121 (TestUpsample)."""
    x = torch.zeros(1, 1, 16, 16)
    x[0, 0, 5, 9] = 1.0
    y = O.upsample2x(x)
    assert y.shape == (1, 1, 32, 32)
    iy, ix = divmod(int(y.flatten().argmax()), 32)
    assert abs(iy - 10.5) <= 1 and abs(ix - 18.5) <= 1


def test_temporal_known_answers():
    """tests/losses/test_losses.py:314-408 (TestTemporalLoss)."""
    s2 = math.sqrt(2.0)
    assert torch.allclose(O.temporal_loss(torch.tensor([[0.0, 0.0], [s2, s2]])), torch.tensor(2.0), atol=1e-6)
    kp = torch.tensor([[0.0, 0.0], [1.0, 1.0], [2.0, 3.0]])  # dists sqrt2, sqrt5
    got = O.temporal_loss(kp, None, 2.1)
    assert torch.allclose(got, torch.tensor((math.sqrt(5.0) - 2.1) / 2), atol=1e-6)
    conf = torch.tensor([[0.9], [0.01], [0.9]])
    assert float(O.temporal_loss(kp, conf, 0.0, 0.05)) == 0.0


def test_heatmap_mse_identity_and_scale():
    """tests/losses/test_losses.py:86-119: equal inputs -> 0; value = mse * h * w."""
    g = torch.Generator().manual_seed(0)
    t = torch.rand(2, 3, 8, 8, generator=g)
    assert float(O.heatmap_mse_loss(t, t)) == 0.0
    p = torch.rand(2, 3, 8, 8, generator=g)
    assert torch.allclose(O.heatmap_mse_loss(t, p), ((t - p) ** 2).mean() * 64, rtol=1e-6)


def test_heatmap_kl_js_zero_on_equal():
    """tests/losses/test_losses.py:122-217."""
    t = torch.softmax(torch.randn(2, 3, 64), -1).reshape(2, 3, 8, 8)
    assert abs(float(O.heatmap_kl_loss(t, t))) < 1e-6
    assert abs(float(O.heatmap_js_loss(t, t))) < 1e-6


def test_pca_in_subspace_is_zero():
    """tests/losses/test_losses.py:220-311 (TestPCALoss): data in the kept subspace -> ~0."""
    g = torch.Generator().manual_seed(1)
    basis = torch.linalg.qr(torch.randn(8, 3, generator=g))[0].T  # (3, 8) orthonormal rows
    mean = torch.randn(8, generator=g)
    data = torch.randn(20, 3, generator=g) @ basis + mean
    assert float(O.pca_loss(data, mean, basis, 0.0)) < 1e-5


def test_generate_heatmaps_semantics():
    """tests/data/test_heatmaps.py:203-454: sums to one, NaN/OOB -> zeros, visibility 0/1/2."""
    kp = torch.tensor([[[10.0, 20.0], [float("nan"), float("nan")], [500.0, 10.0]]])
    hm = O.generate_heatmaps(kp, 64, 64, (16, 16))
    assert torch.allclose(hm[0, 0].sum(), torch.tensor(1.0), atol=1e-5)
    assert float(hm[0, 1].abs().sum()) == 0.0 and float(hm[0, 2].abs().sum()) == 0.0
    iy, ix = divmod(int(hm[0, 0].flatten().argmax()), 16)
    assert (ix, iy) in {(2, 5), (3, 5)}
    vis = torch.tensor([[1, 0, 2]])
    hm = O.generate_heatmaps(kp, 64, 64, (16, 16), visibility=vis)
    assert torch.allclose(hm[0, 0], torch.full((16, 16), 1 / 256.0))
    assert float(hm[0, 1].abs().sum()) == 0.0 and float(hm[0, 2].abs().sum()) == 0.0


def test_confidence_window_interior_and_border():
    """tests/data/test_heatmaps.py:457-563."""
    p = torch.zeros(1, 1, 10, 10)
    p[0, 0, 4, 6] = 0.7
    p[0, 0, 0, 0] = 0.3
    assert torch.allclose(O.confidence_window(p, torch.tensor([[[6.4, 4.9]]])), torch.tensor([[0.7]]))
    assert torch.allclose(O.confidence_window(p, torch.tensor([[[0.0, 0.0]]])), torch.tensor([[0.3]]))


def test_undo_affine_round_trip():
    """tests/data/test_utils.py:120-143: apply then undo, atol 1e-4."""
    g = torch.Generator().manual_seed(2)
    kp = torch.rand(4, 6, generator=g) * 100
    A = torch.tensor([[0.9, -0.2, 4.0], [0.25, 1.1, -7.0]])
    aug = (kp.reshape(4, 3, 2) @ A[:, :2].T + A[:, 2]).reshape(4, 6)
    assert torch.allclose(O.undo_affine(aug, A), kp, atol=1e-4)


def test_factory_anneal_rules():
    """tests/losses/test_factory.py:132-274: heatmap losses ignore anneal; others scale; None == 1."""
    v = torch.tensor(3.0)
    assert float(O.factory_total({"heatmap_mse": (v, 0.0)}, 0.0)) == pytest.approx(1.5)
    assert float(O.factory_total({"temporal": (v, 0.0)}, 0.5)) == pytest.approx(0.75)
    assert float(O.factory_total({"temporal": (v, 0.0)}, None)) == pytest.approx(1.5)


def test_tracker_step_matches_reference_golden(golden):
    """Full semi-supervised step of the reference's SemiSupervisedHeatmapTracker (golden) vs OracleTracker."""
    g = golden("tracker_step")
    model = O.OracleTracker(num_keypoints=3, downsample_factor=2, torch_seed=7)
    assert float(model.backbone[0].weight.detach().double().sum()) == pytest.approx(float(g["w_conv1_sum"]), rel=1e-9)
    assert float(model.head.upsampling_layers[1].weight.detach().double().sum()) == pytest.approx(float(g["w_head1_sum"]), rel=1e-9)
    batch = {
        "labeled": {"images": g.t("images"), "keypoints": g.t("keypoints"), "heatmaps": g.t("heatmaps"), "bbox": g.t("bbox_l")},
        "unlabeled": {"frames": g.t("frames"), "transforms": g.t("A"), "bbox": g.t("bbox_u"), "is_multiview": False},
    }
    model.train()
    loss, logs = O.training_step(model, batch, {"temporal": {"log_weight": 2.0, "epsilon": 1.0, "prob_threshold": 0.0}}, 0.5)
    loss.backward()
    want = dict(zip([str(n) for n in g["log_names"]], g["log_values"]))
    assert set(want) == set(logs)
    for k, v in want.items():
        assert float(logs[k].detach()) == pytest.approx(float(v), rel=2e-4, abs=2e-5), k
    torch.testing.assert_close(model.head.upsampling_layers[2].weight.grad, g.t("g_head_last_w"), rtol=2e-3, atol=1e-7)
    assert float(model.backbone[0].weight.grad.norm()) == pytest.approx(float(g["g_conv1_norm"]), rel=5e-3)
