"""fp32 validation path (Fp32Engine on the lp_f32_* kernels) against the verbatim reference, which trains in fp32 only
(lightning_pose/train.py:411-428): BASELINE.json's north_star tolerance for fp32 is 1e-4.  Every logged scalar, the loss and the
parameter gradients of one semi-supervised step of the reference's own SemiSupervisedHeatmapTracker (tests/golden/tracker_step.npz)."""

import numpy as np
import pytest
import torch

FP32_TOL = 1e-4   # north_star: "within 1e-4 fp32"


def _batch(g, dev):
    d = lambda k: g.t(k).to(dev)  # noqa: E731
    return {
        "labeled": {"images": d("images"), "keypoints": d("keypoints"), "heatmaps": d("heatmaps"), "bbox": d("bbox_l"), "idxs": torch.arange(4)},
        "unlabeled": {"frames": d("frames"), "transforms": d("A"), "bbox": d("bbox_u"), "is_multiview": False},
    }


def test_fp32_training_step_vs_reference_golden(stack_backend, golden):
    dev = stack_backend
    from lightning_pose_amd.engine_fp32 import Fp32Engine
    from lightning_pose_amd.losses import LossFactory
    from lightning_pose_amd.models import SemiSupervisedHeatmapTracker

    g = golden("tracker_step")
    sup = LossFactory({"heatmap_mse": {"log_weight": 0.0}}, None)
    unsup = LossFactory({"temporal": {"log_weight": 2.0, "epsilon": 1.0, "prob_threshold": 0.0}}, None)
    model = SemiSupervisedHeatmapTracker(num_keypoints=3, loss_factory=sup, loss_factory_unsupervised=unsup, backbone="resnet50",
                                         pretrained=False, torch_seed=7, device=dev, precision="fp32")
    assert isinstance(model.net, Fp32Engine)
    model.total_unsupervised_importance = torch.tensor(0.5)
    model.train()
    opt = model.configure_optimizers()["optimizer"]
    opt.zero_grad()
    out = model.training_step(_batch(g, dev), 0)
    out["loss"].backward()
    want = dict(zip([str(n) for n in g["log_names"]], g["log_values"]))
    got = {k: float(v) for k, v in model.logged.items()}
    assert set(want) == set(got)
    for k, v in want.items():
        assert got[k] == pytest.approx(float(v), rel=FP32_TOL, abs=FP32_TOL), k
    assert float(out["loss"]) == pytest.approx(float(g["loss"]), rel=FP32_TOL, abs=FP32_TOL)
    # parameter gradients through the hand-written fp32 backward
    gw = getattr(model.head.upsampling_layers, "2").weight.grad.cpu()
    torch.testing.assert_close(gw, g.t("g_head_last_w"), atol=FP32_TOL * float(g.t("g_head_last_w").abs().max()), rtol=1e-3)
    gb = getattr(model.head.upsampling_layers, "1").bias.grad.cpu()
    torch.testing.assert_close(gb, g.t("g_head_first_b"), atol=FP32_TOL * float(g.t("g_head_first_b").abs().max()) + 1e-12, rtol=1e-3)
    assert float(getattr(model.backbone, "0").weight.grad.norm()) == pytest.approx(float(g["g_conv1_norm"]), rel=2e-3)
    l4 = getattr(getattr(model.backbone, "7"), "2").conv3.weight.grad
    assert float(l4.norm()) == pytest.approx(float(g["g_l4_last_norm"]), rel=2e-3)
    # the reference's train-mode forward AFTER the step's running-statistics updates (no optimiser step yet): heat-maps at 1e-4
    with torch.no_grad():
        heat = model.forward(_batch(g, dev)["labeled"]["images"]).cpu()
    torch.testing.assert_close(heat, g.t("heat_after_step_train_mode"), atol=FP32_TOL * float(g.t("heat_after_step_train_mode").max()), rtol=1e-3)


@pytest.mark.parametrize("M,C_", [(300, 72), (64, 64), (5000, 256), (1, 8)])
def test_fp32_ordered_batchnorm_sums(stack_backend, M, C_):
    """lp_f32_bn_stats_ordered (the validation executor's forward statistics): the sums of lp_f32_bn_stats without atomics - the same values
    up to summation order, and the same bits every time"""
    from lightning_pose_amd import _lib
    from lightning_pose_amd.ops import _p, _stream

    dev = stack_backend
    lib = _lib.lib()
    x = torch.randn(M, C_, generator=torch.Generator().manual_seed(M + C_)).to(dev)
    ref = torch.stack([x.double().sum(0), (x.double() ** 2).sum(0)]).reshape(-1)
    ws = torch.empty(int(lib.lp_f32_bn_stats_workspace_bytes(M, C_)), device=dev, dtype=torch.uint8)
    outs = []
    for _ in range(2):
        s = torch.full((2 * C_,), 0.5, device=dev)   # (accumulated into)
        assert lib.lp_f32_bn_stats_ordered(_p(x), M, C_, _p(s), _p(ws), ws.numel(), _stream()) == 0
        outs.append(s.cpu())
    assert torch.equal(outs[0], outs[1])
    torch.testing.assert_close(outs[0].double() - 0.5, ref.cpu(), rtol=1e-5, atol=5e-4)   # fp32 sums of up to 5000 terms
    a = torch.zeros(2 * C_, device=dev)
    assert lib.lp_f32_bn_stats(_p(x), M, C_, _p(a), _stream()) == 0
    torch.testing.assert_close(a.cpu(), outs[0] - 0.5, rtol=1e-5, atol=5e-4)
    assert lib.lp_f32_bn_stats_ordered(_p(x), M, C_, _p(a), _p(ws), max(ws.numel() - 1, 0), _stream()) != 0   # workspace too small
