"""Opt-in two-stream mode (LP_TWO_STREAMS=1): the labeled and the unlabeled pass of a semi-supervised step on their
own HIP streams.  Sorted last on purpose: its device test is the first time real concurrent streams touch the engine.

CPU: a subprocess on the deterministic (single-threaded) emulator runs one step in both modes and requires bit-identical gradients,
running statistics and logged scalars - the control flow changes no number.  Device: the same step in both modes on real streams."""

import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_stream_control_flow_is_numerically_neutral_on_the_emulator():
    env = dict(os.environ, HIPEMU_THREADS="1")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_two_stream_probe.py")], env=env, capture_output=True, text=True,
                         timeout=900)
    assert res.returncode == 0 and "TWO_STREAM_IDENTICAL" in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]


@pytest.mark.gpu
def test_two_stream_step_on_the_device(monkeypatch):
    """real streams: the losses of the step equal the single-stream step's (the forward is deterministic up to the fp32 atomics of the
    fused BatchNorm sums), every gradient is finite and non-zero, and the optimiser sees both passes' gradients"""
    from lightning_pose_amd.losses import LossFactory
    from lightning_pose_amd.models import SemiSupervisedHeatmapTracker
    from lightning_pose_amd import ops

    dev = torch.device("cuda:0")
    K, HW = 5, 128

    def run(two: bool):
        monkeypatch.setenv("LP_TWO_STREAMS", "1" if two else "0")
        sup = LossFactory({"heatmap_mse": {"log_weight": 0.0}}, None)
        unsup = LossFactory({"temporal": {"log_weight": 2.0, "epsilon": 0.0, "prob_threshold": 0.0},
                             "unimodal_mse": {"log_weight": 2.0, "prob_threshold": 0.0}}, None)
        model = SemiSupervisedHeatmapTracker(num_keypoints=K, loss_factory=sup, loss_factory_unsupervised=unsup, backbone="resnet50",
                                             pretrained=False, torch_seed=9, device=dev)
        g = torch.Generator().manual_seed(4)
        kp = (torch.rand(8, K, 2, generator=g) * HW).to(dev)
        box = torch.tensor([[0.0, 0.0, HW, HW]])
        batch = {"labeled": {"images": torch.randn(8, 3, HW, HW, generator=g).to(dev), "keypoints": kp.reshape(8, 2 * K),
                             "heatmaps": ops.generate_heatmaps(kp, HW, HW, (HW // 4, HW // 4)), "bbox": box.repeat(8, 1).to(dev),
                             "idxs": torch.arange(8)},
                 "unlabeled": {"frames": torch.randn(16, 3, HW, HW, generator=g).to(dev), "transforms": torch.tensor([-1.0]).to(dev),
                               "bbox": box.repeat(16, 1).to(dev), "is_multiview": False}}
        model.train()
        opt = model.configure_optimizers()["optimizer"]
        out = []
        for _ in range(3):
            opt.zero_grad()
            loss = model.training_step(batch, 0)["loss"]
            loss.backward()
            model.net.wait_pending()
            torch.cuda.synchronize()
            out.append(({k: float(v) for k, v in model.logged.items()}, float(model.net.G.abs().sum()), bool(torch.isfinite(model.net.G).all())))
            opt.step()
        assert model.net.two_streams_active() == two
        return out

    single, double = run(False), run(True)
    # backbone lr = 0 and the head's first Adam steps are +-lr: the FIRST step is the comparable one
    for k, v in single[0][0].items():
        assert double[0][0][k] == pytest.approx(v, rel=2e-3, abs=1e-6), k
    for logged, gsum, finite in double:
        assert finite and gsum > 0 and all(torch.isfinite(torch.tensor(list(logged.values()))))
    assert double[0][1] == pytest.approx(single[0][1], rel=0.05)
