"""HeatmapHead on its own, with every constructor option (reference models/heads/heatmap.py:147-212).

The reference's own head tests (tests/models/heads/test_heatmap.py::TestHeatmapHead) build the head outside a tracker - with
``deconv_out_channels=32``, with ``final_softmax=False``, for every ``downsample_factor`` - and call it on random features.  The same
constructions here, on the HIP head (``HeadEngine``), compared with the VERBATIM reference class holding the same weights: values, the
gradient of the features and of every layer, in the product's bf16-mixed arithmetic (tolerances below)."""

import pytest
import torch

from tests.conftest import needs_reference

# bf16 operands, fp32 accumulation: relative to the largest magnitude of the compared tensor
TOL_VALUE = 2e-2
TOL_GRAD = 3e-2


def _reference_head(**kw):
    from oracle import ref_loader

    return ref_loader.load("models.heads.heatmap").HeatmapHead(**kw)


def _pair(device, gain=30.0, **kw):
    """product head + verbatim reference head with the same (enlarged: the initial gain of 0.01 leaves flat maps) weights"""
    from lightning_pose_amd.models.heads.heatmap import HeatmapHead

    torch.manual_seed(5)
    ours = HeatmapHead(**kw, device=device)
    ref = _reference_head(**kw)
    sd = {}
    for k, v in ref.state_dict().items():
        sd[k] = (v * gain + (0.05 * torch.randn_like(v) if k.endswith("bias") else 0)).detach().clone()
    ref.load_state_dict(sd)
    ours.load_state_dict({k: v.to(device) for k, v in sd.items()})
    assert set(ours.state_dict()) == set(ref.state_dict())
    assert all(ours.state_dict()[k].shape == v.shape for k, v in ref.state_dict().items())
    return ours, ref


def _rel(a: torch.Tensor, b: torch.Tensor) -> float:
    return float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max() / b.detach().cpu().double().abs().max().clamp_min(1e-30))


CASES = [
    # (the constructions of the reference's TestHeatmapHead, and a stride-16 backbone with three layers)
    dict(backbone_arch="resnet50", in_channels=256, out_channels=17, deconv_out_channels=32, downsample_factor=2, final_softmax=True),
    dict(backbone_arch="resnet50", in_channels=128, out_channels=10, downsample_factor=2, final_softmax=False),
    dict(backbone_arch="resnet50", in_channels=256, out_channels=17, deconv_out_channels=96, downsample_factor=1, final_softmax=True),
    dict(backbone_arch="vits_dino", in_channels=384, out_channels=5, deconv_out_channels=160, downsample_factor=1, final_softmax=False),
]


@needs_reference
@pytest.mark.parametrize("kw", CASES, ids=["deconv32", "no_softmax", "deconv96_ds1", "vit_deconv160_no_softmax"])
def test_head_alone_vs_the_verbatim_class(stack_backend, kw):
    dev = stack_backend
    ours, ref = _pair(dev, **kw)
    torch.manual_seed(11)
    feat = torch.randn(2, kw["in_channels"], 4, 6)
    f_ref = feat.clone().requires_grad_(True)
    f_our = feat.to(dev).requires_grad_(True)
    y_ref = ref(f_ref)
    y_our = ours(f_our)
    assert y_our.shape == y_ref.shape and y_our.dtype == torch.float32
    assert len(ours.upsampling_layers) == len(ref.upsampling_layers)
    if kw["final_softmax"]:
        sums = y_our.sum(dim=(2, 3)).cpu()
        assert torch.allclose(sums, torch.ones_like(sums), atol=1e-5)
    assert _rel(y_our, y_ref) < TOL_VALUE, _rel(y_our, y_ref)
    g = torch.randn_like(y_ref)
    if kw["final_softmax"]:
        g = g * y_ref.detach()   # (a soft-max output's gradient only matters where the map has mass)
    y_ref.backward(g)
    for p in ours.parameters():
        p.grad.zero_()
    y_our.backward(g.to(dev))
    assert _rel(f_our.grad, f_ref.grad) < TOL_GRAD, _rel(f_our.grad, f_ref.grad)
    named = dict(ours.named_parameters())
    last_bias = f"upsampling_layers.{len(ref.upsampling_layers) - 1}.bias"
    for k, p in ref.named_parameters():
        if kw["final_softmax"] and k == last_bias:
            # a constant added to a map leaves its soft-max unchanged: this gradient is zero in exact arithmetic - fp32 rounding noise in the
            # reference, the sum of the bf16 roundings of the logits' gradients here
            scale = float(named[k.replace("bias", "weight")].grad.abs().max())
            assert float(named[k].grad.abs().max()) < 0.1 * scale and float(p.grad.abs().max()) < 1e-3 * scale
            continue
        assert _rel(named[k].grad, p.grad) < TOL_GRAD, (k, _rel(named[k].grad, p.grad))


def test_head_alone_trains(stack_backend):
    """parameters are real leaves: an optimiser step on them changes the next forward pass (the bf16 operand copies follow the masters)"""
    from lightning_pose_amd.models.heads.heatmap import HeatmapHead

    dev = stack_backend
    torch.manual_seed(2)
    head = HeatmapHead("resnet50", 128, 6, deconv_out_channels=16, downsample_factor=2, final_softmax=False, device=dev)
    feat = torch.randn(1, 128, 4, 4, device=dev)
    target = torch.randn(1, 6, 32, 32, device=dev)
    opt = torch.optim.SGD(head.parameters(), lr=5.0)
    losses = []
    for _ in range(3):
        for p in head.parameters():
            p.grad.zero_()
        loss = ((head(feat) - target) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[2] < losses[1] < losses[0], losses


def test_tracker_heads_keep_the_default_configuration():
    """inside a tracker the module only names engine parameters; the reference's trackers never pass the options (heatmap_tracker.py:89-94)"""
    from lightning_pose_amd.models.heads.heatmap import HeatmapHead

    h = HeatmapHead("resnet50", 2048, 17, _bound=True)
    assert h.net is None and len(h.upsampling_layers) == 3
    with pytest.raises(RuntimeError):
        h(torch.zeros(1, 2048, 2, 2))
    with pytest.raises(NotImplementedError):
        HeatmapHead("resnet50", 2048, 17, deconv_out_channels=32, _bound=True)
    for ds in (1, 2, 3):   # reference test_different_downsample_factors
        assert len(HeatmapHead("resnet50", 256, 17, downsample_factor=ds, _bound=True).upsampling_layers) == 4 - ds + 1
