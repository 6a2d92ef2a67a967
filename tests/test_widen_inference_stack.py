"""Inference path of the engines (late-sorting file: written after round 1's last device run): Engine.forward_infer (BatchNorm folded,
one launch per trunk layer), ViTEngine.forward_infer (no attention probabilities, no tape), and the tracker taking them under
eval + no_grad."""

import torch

from oracle import restated as O
from tests.test_emu_engine import close


def test_engine_forward_infer_matches_eval_forward(stack_backend):
    """Inference path (BatchNorm folded into the convolutions, one launch per layer) vs the eval-mode training-path forward and vs the
    fp32 reference-architecture oracle in eval mode, with non-trivial running statistics / affine parameters"""
    dev = stack_backend
    from lightning_pose_amd.engine import Engine
    from lightning_pose_amd.models.backbones._init import seeded_state_dict

    K = 3
    torch.manual_seed(11)
    sd = seeded_state_dict(K, 2)
    gen = torch.Generator().manual_seed(5)
    for k in list(sd):
        if k.startswith("head") and k.endswith("weight"):
            sd[k] = sd[k] * 60
        if k.endswith("running_mean"):
            sd[k] = torch.randn(sd[k].shape, generator=gen) * 0.1
        if k.endswith("running_var"):
            sd[k] = 0.5 + torch.rand(sd[k].shape, generator=gen)
        if ".bn" in k or k.startswith("backbone.1.") or "downsample.1" in k:
            if k.endswith(".weight"):
                sd[k] = 0.7 + 0.6 * torch.rand(sd[k].shape, generator=gen)
            if k.endswith(".bias"):
                sd[k] = torch.randn(sd[k].shape, generator=gen) * 0.1
    eng = Engine(K, 2, dev)
    eng.load_state_dict(sd, strict=False)
    images = torch.randn(3, 3, 64, 64, generator=gen)
    want, _ = eng.forward(images.to(dev), training=False)
    got = eng.forward_infer(images.to(dev))
    assert got.shape == want.shape == (3, K, 16, 16)
    got, want = got.cpu(), want.cpu()
    close("infer vs eval forward", got, want, cos_min=0.999, ratio_tol=0.01)
    ref = O.OracleTracker(K, 2, torch_seed=11)
    ref.load_state_dict(sd, strict=False)
    ref.eval()
    with torch.no_grad():
        oracle = ref(images)
    close("infer vs fp32 oracle", got, oracle, cos_min=0.995, ratio_tol=0.03)
    close("eval forward vs fp32 oracle", want, oracle, cos_min=0.995, ratio_tol=0.03)
    torch.testing.assert_close(got.sum((-1, -2)), torch.ones(3, K), atol=1e-4, rtol=0)
    # the folded copies are cached, and dropped when the weights move
    assert eng._fold is not None
    fold_id = id(eng._fold)
    eng.forward_infer(images.to(dev))
    assert id(eng._fold) == fold_id
    eng.refresh_weight_copies()
    assert eng._fold is None


def test_tracker_eval_no_grad_takes_the_inference_path(stack_backend, monkeypatch):
    from lightning_pose_amd.models import HeatmapTracker

    dev = stack_backend
    model = HeatmapTracker(num_keypoints=3, backbone="resnet50", pretrained=False, torch_seed=3, device=dev)
    calls = []
    real = model.net.forward_infer
    monkeypatch.setattr(model.net, "forward_infer", lambda x: (calls.append(tuple(x.shape)), real(x))[1])
    images = torch.randn(2, 3, 64, 64).to(dev)
    model.eval()
    with torch.no_grad():
        heat = model(images)
    assert calls == [(2, 3, 64, 64)] and tuple(heat.shape) == (2, 3, 16, 16)
    batch = {"images": images, "bbox": torch.tensor([[0.0, 0.0, 64.0, 64.0]]).repeat(2, 1).to(dev)}
    model.predict_step(batch, 0)
    assert len(calls) == 1  # with autograd on, the taped path runs (a caller may differentiate through it)
    from lightning_pose_amd.utils.predictions import predict_batches
    predict_batches(model, [batch])
    assert len(calls) == 2  # the prediction loop (eval + no_grad, what pl.Trainer.predict does) takes the inference path
    model.train()
    with torch.no_grad():
        model(images)
    assert len(calls) == 2  # training mode never folds: it normalises with batch statistics


def test_vit_engine_forward_infer_is_bit_identical(stack_backend):
    from lightning_pose_amd.models.backbones._init import head_state_dict, vit_seeded_state_dict
    from lightning_pose_amd.vit_engine import ViTEngine

    dev = stack_backend
    K = 4
    eng = ViTEngine(K, 2, dev, hidden=128, depth=2, heads=2, mlp=256, patch=16, pretrain_grid=3)
    torch.manual_seed(4)
    sd = vit_seeded_state_dict(128, 2, 2, 256, 16, 3)
    sd.update(head_state_dict(128, K, 1))
    eng.load_state_dict(sd, strict=False)
    images = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(2)).to(dev)
    heat, _tape = eng.forward(images, True)
    torch.testing.assert_close(eng.forward_infer(images).cpu(), heat.cpu(), atol=0, rtol=0)
