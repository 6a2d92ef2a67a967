"""End-to-end CPU test of the PRODUCT host stack (models / losses / engine / optimiser) with the kernel library
replaced by its CPU-emulated build: one semi-supervised training step against the golden step of the reference's own
SemiSupervisedHeatmapTracker (tests/golden/tracker_step.npz).  The GPU run of the same comparison is
tests/test_gpu_tracker.py."""

import numpy as np
import pytest
import torch

from oracle import restated as O
from tests.hipemu import emu


def _batch(g, dev):
    d = lambda k: g.t(k).to(dev)  # noqa: E731
    return {
        "labeled": {"images": d("images"), "keypoints": d("keypoints"), "heatmaps": d("heatmaps"), "bbox": d("bbox_l"),
                    "idxs": torch.arange(4)},
        "unlabeled": {"frames": d("frames"), "transforms": d("A"), "bbox": d("bbox_u"), "is_multiview": False},
    }


def test_tracker_training_step_vs_reference_golden(stack_backend, golden):
    dev = stack_backend
    from lightning_pose_amd.losses import LossFactory
    from lightning_pose_amd.models import SemiSupervisedHeatmapTracker

    g = golden("tracker_step")
    sup = LossFactory({"heatmap_mse": {"log_weight": 0.0}}, None)
    unsup = LossFactory({"temporal": {"log_weight": 2.0, "epsilon": 1.0, "prob_threshold": 0.0}}, None)
    model = SemiSupervisedHeatmapTracker(num_keypoints=3, loss_factory=sup, loss_factory_unsupervised=unsup, backbone="resnet50",
                                         pretrained=False, torch_seed=7, device=dev)
    # seeded initialisation equals the reference's, and the state_dict uses the reference's names / shapes
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    assert float(sd["backbone.0.weight"].double().sum()) == pytest.approx(float(g["w_conv1_sum"]), rel=1e-9)
    assert float(sd["head.upsampling_layers.1.weight"].double().sum()) == pytest.approx(float(g["w_head1_sum"]), rel=1e-9)
    ref = O.OracleTracker(3, 2, torch_seed=7).state_dict()
    assert set(sd) == set(ref)
    for k in ref:
        assert tuple(sd[k].shape) == tuple(ref[k].shape), k
    torch.testing.assert_close(sd["backbone.7.2.conv3.weight"].contiguous(), ref["backbone.7.2.conv3.weight"])

    model.total_unsupervised_importance = torch.tensor(0.5)
    model.train()
    opt = model.configure_optimizers()["optimizer"]
    opt.zero_grad()
    out = model.training_step(_batch(g, dev), 0)
    out["loss"].backward()
    want = dict(zip([str(n) for n in g["log_names"]], g["log_values"]))
    assert set(want) == set(model.logged)
    got = {k: float(v) for k, v in model.logged.items()}
    # bf16 trunk vs the fp32 reference: heat-map losses agree to ~1e-3 relative; keypoint-space quantities on the
    # nearly flat heat-maps of a random-init network are ill-conditioned under softmax(T=1000) and only sanity-checked
    for k in ("total_unsupervised_importance", "heatmap_mse_weight", "temporal_weight"):
        assert got[k] == pytest.approx(float(want[k]), rel=1e-6), k
    for k in ("train_heatmap_mse_loss", "train_heatmap_mse_loss_weighted", "train_supervised_loss"):
        assert got[k] == pytest.approx(float(want[k]), rel=5e-3), k
    print("\nTRACKER_STEP", {k: (round(got[k], 6), round(float(want[k]), 6)) for k in want})
    # keypoint-space scalars of this RANDOM-INIT step (flat heat-maps, chaotic trunk): the fixture's temporal loss is exactly 0 (every
    # frame-to-frame move is below epsilon = 1 px) and stays 0; RMSE and the total follow the heat-map loss.  Their VALUES are pinned at
    # 1e-4 by the fp32 path on this same fixture (tests/test_fp32_parity.py) and, for both precisions on fitted heat-maps at the
    # BASELINE configs, by tests/test_step_parity.py
    assert got["train_temporal_loss"] == pytest.approx(float(want["train_temporal_loss"]), abs=1e-3)
    assert got["train_supervised_rmse"] == pytest.approx(float(want["train_supervised_rmse"]), rel=5e-2)
    assert got["total_loss"] == pytest.approx(float(want["total_loss"]), rel=5e-3)
    # gradients reached every parameter group through the hand-written backward (their VALUES are checked block by
    # block in tests/test_emu_engine.py: a random-init 50-layer BatchNorm net at batch 4 is chaotic end to end)
    gw = getattr(model.head.upsampling_layers, "2").weight.grad
    g_conv1 = getattr(model.backbone, "0").weight.grad
    assert torch.isfinite(gw).all() and float(gw.norm()) > 0
    assert float(g_conv1.norm()) == pytest.approx(float(g["g_conv1_norm"]), rel=0.5)
    # optimiser: backbone lr = 0 keeps weights, head moves
    w_bb = getattr(model.backbone, "0").weight.detach().clone()
    w_hd = getattr(model.head.upsampling_layers, "2").weight.detach().clone()
    opt.step()
    assert torch.equal(getattr(model.backbone, "0").weight.detach(), w_bb)
    assert not torch.equal(getattr(model.head.upsampling_layers, "2").weight.detach(), w_hd)
    assert opt.param_groups[0]["name"] == "backbone" and opt.param_groups[0]["lr"] == 0


def test_checkpoint_interchange_with_reference_layout(stack_backend, tmp_path):
    """a checkpoint written here has the reference's Lightning layout (the oracle's verbatim-named tracker loads its
    state_dict strictly), round-trips bit-exactly into a fresh model, honours the legacy key remap and the backbone-only retry."""
    dev = stack_backend
    from lightning_pose_amd import checkpoint as ck
    from lightning_pose_amd.models import HeatmapTracker

    model = HeatmapTracker(num_keypoints=3, backbone="resnet50", pretrained=False, torch_seed=3, device=dev)
    assert model.hparams["num_keypoints"] == 3 and model.hparams["backbone"] == "resnet50" and "loss_factory" not in model.hparams
    model.current_epoch, model.global_step = 4, 17
    path = ck.save_checkpoint(model, str(tmp_path / "run" / "tb_logs" / "version_0" / "checkpoints" / "epoch=4-step=17-best.ckpt"))
    raw = torch.load(path, map_location="cpu")  # weights_only=True: plain tensors and python containers only
    assert {"state_dict", "hyper_parameters", "epoch", "global_step", "pytorch-lightning_version"} <= set(raw)
    # the reference-side module (same names / shapes as lightning_pose.models.HeatmapTracker) accepts it strictly
    ref = O.OracleTracker(3, 2, torch_seed=0)
    ref.load_state_dict(raw["state_dict"], strict=True)
    torch.testing.assert_close(ref.state_dict()["backbone.4.0.conv1.weight"], raw["state_dict"]["backbone.4.0.conv1.weight"], atol=0, rtol=0)
    # ... and a checkpoint in the reference's layout loads here: directory form, different seed -> identical weights afterwards
    other = ck.load_model_from_checkpoint(str(tmp_path / "run"), device=dev)
    assert other.current_epoch == 4 and other.global_step == 17 and other.num_keypoints == 3
    a, b = model.state_dict(), other.state_dict()
    assert set(a) == set(b)
    for k in a:
        assert torch.equal(a[k].cpu(), b[k].cpu()), k
    # the bf16 operand copies the kernels read were refreshed too
    assert torch.equal(model.net.Wb.cpu(), other.net.Wb.cpu())
    # legacy checkpoints kept the head under "upsampling_layers.*"
    legacy = dict(raw)
    legacy["state_dict"] = {(k[len("head."):] if k.startswith("head.") else k): v for k, v in raw["state_dict"].items()}
    torch.save(legacy, str(tmp_path / "legacy.ckpt"))
    assert set(ck.read_state_dict(str(tmp_path / "legacy.ckpt"))) == set(raw["state_dict"])
    # a model with another number of keypoints keeps the checkpoint's backbone and its own head
    five = HeatmapTracker(num_keypoints=5, backbone="resnet50", pretrained=False, torch_seed=9, device=dev)
    head_before = five.state_dict()["head.upsampling_layers.2.weight"].cpu().clone()
    ck.load_weights(five, path)
    assert torch.equal(five.state_dict()["backbone.7.2.conv3.weight"].cpu(), a["backbone.7.2.conv3.weight"].cpu())
    assert torch.equal(five.state_dict()["head.upsampling_layers.2.weight"].cpu(), head_before)


def test_checkpoint_semi_supervised_vit_round_trip(stack_backend, tmp_path):
    """the same interchange for the ViT tracker and the semi-supervised class: hyper-parameters (backbone, image_size, ...) rebuild the
    model, loss factories come back in through the overrides, HF parameter names survive the trip"""
    dev = stack_backend
    from lightning_pose_amd import checkpoint as ck
    from lightning_pose_amd.losses import LossFactory
    from lightning_pose_amd.models import SemiSupervisedHeatmapTracker

    sup = LossFactory({"heatmap_mse": {"log_weight": 0.0}}, None)
    unsup = LossFactory({"temporal": {"log_weight": 5.0, "epsilon": 5.0}}, None)
    model = SemiSupervisedHeatmapTracker(num_keypoints=2, loss_factory=sup, loss_factory_unsupervised=unsup, backbone="vits_dino",
                                         pretrained=False, torch_seed=1, image_size=64, device=dev)
    assert model.hparams["backbone"] == "vits_dino" and model.hparams["image_size"] == 64
    path = ck.save_checkpoint(model, str(tmp_path / "vit.ckpt"), optimizer=model.configure_optimizers()["optimizer"])
    raw = torch.load(path, map_location="cpu", weights_only=False)
    assert any(k.startswith("backbone.vision_encoder.") for k in raw["state_dict"]) and "lp_amd_optimizer_state" in raw
    again = ck.load_model_from_checkpoint(path, loss_factory=sup, loss_factory_unsupervised=unsup, device=dev)
    assert type(again) is SemiSupervisedHeatmapTracker and again.loss_factory_unsup is unsup
    a, b = model.state_dict(), again.state_dict()
    assert set(a) == set(b) and all(torch.equal(a[k].cpu(), b[k].cpu()) for k in a)
    with pytest.raises(ValueError):
        ck.load_model_from_checkpoint(None)


def test_reference_default_backbone_name_and_mmpose_checkpoint(stack_backend, tmp_path):
    """the reference's shipped default is backbone='resnet50_animal_ap10k' (config_default.yaml:121): same ResNet-50, weights from an
    mmpose checkpoint whose keys carry a 'backbone.' prefix (models/backbones/factory.py:253-266) - offline, that file is handed in
    as backbone_checkpoint"""
    dev = stack_backend
    from lightning_pose_amd.models import HeatmapTracker
    from lightning_pose_amd.models.backbones.factory import backbone_features

    assert backbone_features("resnet50_animal_ap10k") == 2048
    with pytest.raises(ValueError):
        backbone_features("resnet51")
    with pytest.raises(RuntimeError):  # pretrained weights cannot be downloaded here: fail loudly, never fall back to random init
        HeatmapTracker(num_keypoints=3, backbone="resnet50_animal_ap10k", pretrained=True, device=dev)
    ref = O.OracleTracker(3, 2, torch_seed=11).state_dict()
    tv_names = {"backbone.0": "conv1", "backbone.1": "bn1", "backbone.4": "layer1", "backbone.5": "layer2", "backbone.6": "layer3",
                "backbone.7": "layer4"}
    mm = {}
    for k, v in ref.items():
        for ours, tv in tv_names.items():
            if k.startswith(ours + "."):
                mm["backbone." + tv + k[len(ours):]] = v.clone()
    mm["keypoint_head.final_layer.weight"] = torch.zeros(3, 2048, 1, 1)  # mmpose heads are ignored
    torch.save({"state_dict": mm, "meta": {}}, str(tmp_path / "res50_ap10k.pth"))
    model = HeatmapTracker(num_keypoints=3, backbone="resnet50_animal_ap10k", pretrained=True, torch_seed=5,
                           backbone_checkpoint=str(tmp_path / "res50_ap10k.pth"), device=dev)
    sd = model.state_dict()
    for k in ("backbone.0.weight", "backbone.4.0.conv1.weight", "backbone.7.2.bn3.running_var", "backbone.6.3.conv2.weight"):
        assert torch.equal(sd[k].cpu().contiguous(), ref[k]), k
    assert not torch.equal(sd["head.upsampling_layers.2.weight"].cpu(), ref["head.upsampling_layers.2.weight"])  # own seed (5 vs 11)


def test_reference_default_config_drives_the_registry(stack_backend, monkeypatch):
    """scripts/configs/config_default.yaml of the reference (read here when /root/reference is present, else its relevant keys restated)
    goes through get_loss_factories / get_model unchanged: backbone 'resnet50_animal_ap10k', heatmap_loss_type 'mse', the `losses`
    hyper-parameter block, losses_to_use switching the semi-supervised class on."""
    import os

    import yaml

    from lightning_pose_amd.losses.factory import get_loss_factories
    from lightning_pose_amd.models import HeatmapTracker, SemiSupervisedHeatmapTracker, heatmap_tracker
    from lightning_pose_amd.models.factory import get_model

    dev = stack_backend
    monkeypatch.setattr(heatmap_tracker, "_default_device", lambda: dev)
    path = "/root/reference/scripts/configs/config_default.yaml"
    if os.path.exists(path):
        cfg = yaml.safe_load(open(path))
    else:
        cfg = {"data": {"mirrored_column_matches": None, "columns_for_singleview_pca": None},
               "training": {"rng_seed_model_pt": 0, "optimizer": "Adam", "optimizer_params": {"learning_rate": 1e-3},
                            "lr_scheduler": "multisteplr", "lr_scheduler_params": {"multisteplr": {"milestones": [150, 200, 250], "gamma": 0.5}}},
               "model": {"losses_to_use": [], "backbone": "resnet50_animal_ap10k", "model_type": "heatmap", "heatmap_loss_type": "mse",
                         "checkpoint": None},
               "losses": {"temporal": {"log_weight": 11.0, "epsilon": 20.0, "prob_threshold": 0.05}}}
    assert cfg["model"]["backbone"] == "resnet50_animal_ap10k"
    cfg["data"].update(image_resize_dims={"height": 128, "width": 128}, num_keypoints=3, keypoint_names=["a", "b", "c"])
    cfg["model"]["backbone_pretrained"] = False  # (the default True needs the mmpose weights: see the checkpoint test above)
    factories = get_loss_factories(cfg, None)
    assert list(factories["supervised"].loss_instance_dict) == ["heatmap_mse"] and not factories["unsupervised"].loss_instance_dict
    model = get_model(cfg, None, factories)
    assert type(model) is HeatmapTracker and model.backbone_arch == "resnet50_animal_ap10k" and model.downsample_factor == 2
    opt = model.configure_optimizers()
    assert opt["monitor"] == "val_supervised_loss" and [g["name"] for g in opt["optimizer"].param_groups] == ["backbone", "head"]
    cfg["model"]["losses_to_use"] = ["temporal"]
    factories = get_loss_factories(cfg, None)
    t = factories["unsupervised"].loss_instance_dict["temporal"]
    assert float(t.epsilon) == 20.0 and float(t.prob_threshold) == pytest.approx(0.05) and float(t.weight) == pytest.approx(0.5 / np.exp(11.0))
    assert type(get_model(cfg, None, factories)) is SemiSupervisedHeatmapTracker
