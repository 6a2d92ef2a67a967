"""The optimisation step as one captured HIP graph (lightning_pose_amd/graph_step.py): lp_adam_step_dev reads the per-step scalars from
device memory so the captured launches never change; on the device a graphed run must reproduce the eager run."""

import ctypes as C

import numpy as np
import pytest
import torch

from tests.hipemu import emu


def test_adam_step_dev_equals_adam_step(kernel_backend):
    gen = torch.Generator().manual_seed(1)
    n = 5000
    p0, g = torch.randn(n, generator=gen).numpy(), torch.randn(n, generator=gen).numpy()
    m0, v0 = (torch.randn(n, generator=gen) * 0.1).numpy(), (torch.rand(n, generator=gen) * 0.01).numpy()
    for wd, dec in ((0.0, 0), (0.01, 1), (0.02, 0)):
        for step, lr in ((1, 1e-3), (7, 5e-4), (40, 0.0)):
            a = [emu.Buf(x.copy()) for x in (p0, g, m0, v0)]
            b = [emu.Buf(x.copy()) for x in (p0, g, m0, v0)]
            wa, wb = emu.Z(n, np.uint16), emu.Z(n, np.uint16)
            emu.ok(emu.lib().lp_adam_step(a[0].p, a[1].p, a[2].p, a[3].p, n, lr, 0.9, 0.999, 1e-8, wd, dec, step, 0.5, wa.p, emu.stream()))
            hyper = emu.Buf(np.array([lr, 1 - 0.9 ** step, (1 - 0.999 ** step) ** 0.5, 0], np.float32))
            emu.ok(emu.lib().lp_adam_step_dev(b[0].p, b[1].p, b[2].p, b[3].p, n, hyper.p, 0.9, 0.999, 1e-8, wd, dec, 0.5, wb.p, emu.stream()))
            for x, y in zip(a + [wa], b + [wb]):
                np.testing.assert_allclose(x.np().astype(np.float64), y.np().astype(np.float64), rtol=2e-6, atol=1e-7)


@pytest.mark.gpu
def test_graphed_steps_reproduce_eager_steps(monkeypatch):
    from lightning_pose_amd import ops
    from lightning_pose_amd.losses import LossFactory
    from lightning_pose_amd.models import SemiSupervisedHeatmapTracker
    from lightning_pose_amd.trainer import Trainer

    dev = torch.device("cuda:0")
    K, HW = 5, 128

    def run(graph: bool):
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
                 "unlabeled": {"frames": torch.randn(8, 3, HW, HW, generator=g).to(dev), "transforms": torch.tensor([-1.0]).to(dev),
                               "bbox": box.repeat(8, 1).to(dev), "is_multiview": False}}
        model.train()
        trainer = Trainer(max_epochs=1, data_parallel=False, hip_graph=graph)
        trainer.setup(model)
        for g_ in model.optimizers().param_groups:   # every group trains, so both refresh their weight copies inside the step
            g_["lr"] = 1e-4
        losses, hm = [], []
        for i in range(6):
            loss = trainer.training_batch(model, batch, i)
            torch.cuda.synchronize()
            losses.append(float(loss))
            hm.append(float(model.logged["train_heatmap_mse_loss"]))
        gs = trainer._graphed
        return model, losses, hm, gs

    m0, l0, h0, _ = run(False)
    m1, l1, h1, gs = run(True)
    assert gs is not None and gs.captures == 1 and gs.replays == 4 and gs.eager_steps == 2
    assert m1.global_step == m0.global_step == 6 and int(m1.net.nbt) == int(m0.net.nbt) == 12
    assert [g_["step"] for g_ in m1.optimizers().param_groups] == [6, 6]
    # same trajectory (fp32 atomics in the fused BatchNorm sums make two runs differ in the last bits, which the bf16 trunk amplifies)
    np.testing.assert_allclose(h1, h0, rtol=2e-2)
    np.testing.assert_allclose(l1, l0, rtol=0.2)
    a, b = m0.net.P[m0.net.plan.n_backbone:], m1.net.P[m1.net.plan.n_backbone:]
    assert float((a - b).abs().max()) <= 12.5 * 1e-4   # Adam moves a weight by <= lr per step: two runs are at most 2 x 6 x lr apart


@pytest.mark.gpu
def test_graphed_steps_keep_the_callers_batches_and_refresh_the_inference_copies():
    """(round-2 advisor findings)  A list of batches reused every epoch: the captured graph reads PRIVATE input buffers, so the caller's
    batch 0 still holds batch 0's data after later batches were loaded; and validation after replayed steps runs on weights folded from
    the CURRENT parameters / running statistics, as after eager steps."""
    from lightning_pose_amd import ops
    from lightning_pose_amd.losses import LossFactory
    from lightning_pose_amd.models import SemiSupervisedHeatmapTracker
    from lightning_pose_amd.trainer import Trainer

    dev = torch.device("cuda:0")
    K, HW = 5, 128
    box = torch.tensor([[0.0, 0.0, HW, HW]])

    def make_batch(seed):
        g = torch.Generator().manual_seed(seed)
        kp = (torch.rand(8, K, 2, generator=g) * HW).to(dev)
        return {"labeled": {"images": torch.randn(8, 3, HW, HW, generator=g).to(dev), "keypoints": kp.reshape(8, 2 * K),
                            "heatmaps": ops.generate_heatmaps(kp, HW, HW, (HW // 4, HW // 4)), "bbox": box.repeat(8, 1).to(dev),
                            "idxs": torch.arange(8)},
                "unlabeled": {"frames": torch.randn(8, 3, HW, HW, generator=g).to(dev), "transforms": torch.tensor([-1.0]).to(dev),
                              "bbox": box.repeat(8, 1).to(dev), "is_multiview": False}}

    def run(graph: bool):
        sup = LossFactory({"heatmap_mse": {"log_weight": 0.0}}, None)
        unsup = LossFactory({"temporal": {"log_weight": 2.0, "epsilon": 0.0, "prob_threshold": 0.0}}, None)
        model = SemiSupervisedHeatmapTracker(num_keypoints=K, loss_factory=sup, loss_factory_unsupervised=unsup, backbone="resnet50",
                                             pretrained=False, torch_seed=9, device=dev)
        batches = [make_batch(4), make_batch(5), make_batch(6)]
        keep = batches[0]["labeled"]["images"].clone(), batches[0]["unlabeled"]["frames"].clone()
        model.train()
        trainer = Trainer(max_epochs=1, data_parallel=False, hip_graph=graph)
        trainer.setup(model)
        vals, psums = [], []
        for epoch in range(3):
            for i, b in enumerate(batches):
                trainer.training_batch(model, b, i)
            vals.append(trainer.validate(model, [batches[0]["labeled"]])["val_supervised_loss"])
            psums.append(float(model.net.P.double().abs().sum()))
        torch.cuda.synchronize()
        assert torch.equal(batches[0]["labeled"]["images"], keep[0]) and torch.equal(batches[0]["unlabeled"]["frames"], keep[1])
        # the weights moved between the validations (checked on the master parameters: the validation loss of this random-init network
        # repeated to the last digit across epochs in 1 of 14 device runs, profiles/r03_flake6.log - a saturated soft-max does not see small updates)
        assert psums[0] != psums[-1]
        return vals, trainer._graphed

    v0, _ = run(False)
    v1, gs = run(True)
    assert gs is not None and gs.replays >= 5
    # ... and the graphed run's validation follows them.  (Two runs of the SAME eager code differ too - BatchNorm sums go through fp32 atomics and a
    # random-init ResNet at batch 8 amplifies that over 9 steps: 1 - 2 % usually, 3 % exceeded once in ~10 device runs, profiles/r03_final_pytest_gpu_run4.log.)
    np.testing.assert_allclose(v1, v0, rtol=1e-1)
