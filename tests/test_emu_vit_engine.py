"""ViT backbone + head (lightning_pose_amd.vit_engine.ViTEngine) against the third-party module the reference calls -
HuggingFace ``ViTModel`` with ``interpolate_pos_encoding=True`` (models/backbones/vit.py:16-49) - plus the reference head,
in fp32 on the CPU.  A small configuration (2 layers, 2 heads of 64, 3x3 pretraining grid resized to 4x4) keeps it emulator-sized;
the bf16-mixed policy bounds the agreement."""

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

transformers = pytest.importorskip("transformers")


def _oracle(K, hidden, depth, heads, mlp, grid0, seed):
    from transformers import ViTConfig, ViTModel
    torch.manual_seed(seed)
    cfg = ViTConfig(hidden_size=hidden, num_hidden_layers=depth, num_attention_heads=heads, intermediate_size=mlp, image_size=16 * grid0,
                    patch_size=16, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    vit = ViTModel(cfg, add_pooling_layer=False).eval()
    with torch.no_grad():  # make every parameter non-trivial (HF initialises biases / LayerNorm to 0 / 1)
        for n, p in vit.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
        vit.embeddings.cls_token.normal_(std=0.5)
        vit.embeddings.position_embeddings.normal_(std=0.5)
    head = nn.Sequential(nn.PixelShuffle(2), nn.ConvTranspose2d(hidden // 4, K, 3, 2, 1, 1))
    with torch.no_grad():
        head[1].weight.normal_(std=0.3)
        head[1].bias.normal_(std=0.1)
    return vit, head


def _oracle_forward(vit, head, images):
    hs = vit(images, interpolate_pos_encoding=True).last_hidden_state[:, 1:]
    n = int(hs.shape[1] ** 0.5)
    feat = hs.reshape(images.shape[0], n, n, -1).permute(0, 3, 1, 2)
    logits = head(feat)
    b, k, h, w = logits.shape
    return torch.softmax(logits.reshape(b, k, -1), -1).reshape(b, k, h, w)


@pytest.mark.parametrize("hidden,depth,heads,mlp", [(128, 2, 2, 256)])
def test_vit_engine_forward_backward_vs_hf(stack_backend, hidden, depth, heads, mlp):
    check_vit_engine_vs_hf(stack_backend, hidden, depth, heads, mlp)


@pytest.mark.parametrize("batch,size", [(2, 64), (3, 96)])
def test_fp32_vit_engine_vs_hf(stack_backend, batch, size):
    """the fp32 VALIDATION executor (vit_engine_fp32.Fp32ViTEngine on csrc/vit_f32.hip + lp_f32_conv_*): the same comparison at fp32 bars"""
    check_vit_engine_vs_hf(stack_backend, 128, 2, 2, 256, batch=batch, size=size, fp32=True)


def check_vit_engine_vs_hf(stack_backend, hidden, depth, heads, mlp, batch=2, size=64, fp32=False):
    """(ViT-B's width and 12 heads run this from tests/test_widen_vitb_width.py)"""
    from lightning_pose_amd.vit_engine import ViTEngine
    if fp32:
        from lightning_pose_amd.vit_engine_fp32 import Fp32ViTEngine as ViTEngine  # noqa: F811

    dev = stack_backend
    K, grid0 = 5, 3
    vit, head = _oracle(K, hidden, depth, heads, mlp, grid0, seed=0)
    eng = ViTEngine(K, 2, dev, hidden=hidden, depth=depth, heads=heads, mlp=mlp, patch=16, pretrain_grid=grid0)
    sd = {f"backbone.vision_encoder.{k}": v for k, v in vit.state_dict().items()}
    sd["head.upsampling_layers.1.weight"] = head[1].weight.detach()
    sd["head.upsampling_layers.1.bias"] = head[1].bias.detach()
    eng.load_state_dict(sd, strict=True)
    # the state_dict round-trips under the reference's names and shapes
    for k, v in eng.state_dict().items():
        torch.testing.assert_close(v.cpu(), sd[k].reshape(v.shape), atol=0, rtol=0)

    gen = torch.Generator().manual_seed(1)
    images = torch.randn(batch, 3, size, size, generator=gen)
    heat, tape = eng.forward(images.to(dev), True)
    want = _oracle_forward(vit, head, images)
    assert heat.shape == want.shape == (batch, K, size // 4, size // 4)
    # bf16 operands vs fp32; the 768-wide head sums 192 products per logit, its sharper soft-max doubles the relative error of a peak
    if fp32:
        torch.testing.assert_close(heat.cpu(), want.detach(), atol=1e-6, rtol=1e-4)
    else:
        torch.testing.assert_close(heat.cpu(), want.detach(), atol=2e-3, rtol=5e-2 if hidden == 128 else 1e-1)

    g = torch.randn(want.shape, generator=gen)
    (want * g).sum().backward()
    eng.zero_grad()
    eng.backward(tape, g.to(dev))
    grads = eng.grad_views()
    ref = {f"backbone.vision_encoder.{k}": p.grad for k, p in vit.named_parameters()}
    ref["head.upsampling_layers.1.weight"] = head[1].weight.grad
    ref["head.upsampling_layers.1.bias"] = head[1].bias.grad
    for k, gr in ref.items():
        got = grads[k].cpu().reshape(gr.shape)
        if gr.norm() < 1e-5:
            # analytically zero (soft-max is invariant to the key bias and to the head's per-channel bias): only rounding noise
            assert got.norm() < (1e-5 if fp32 else 5e-3), (k, got.norm().item())
            continue
        cos = F.cosine_similarity(got.flatten(), gr.flatten(), dim=0).item()
        rel = ((got - gr).norm() / gr.norm()).item()
        if fp32:
            assert rel < 1e-4, (k, cos, rel)
        else:
            assert cos > 0.999 and rel < 0.03, (k, cos, rel)  # torch.autocast(bf16) of the same model: cos 0.9999, rel 0.010-0.015


@pytest.mark.parametrize("batch,size", [(3, 96), (1, 160)])
def test_vit_engine_odd_batches_and_token_counts(stack_backend, batch, size):
    """37 and 101 tokens (6 x 6 and 10 x 10 patch grids: the key axis is padded to 64 / 128 inside the attention kernels, the position table
    is interpolated from the 3 x 3 pretraining grid), batches of 3 and 1"""
    check_vit_engine_vs_hf(stack_backend, 128, 2, 2, 256, batch=batch, size=size)


def test_vit_tracker_trains_through_the_reference_surface(stack_backend, monkeypatch):
    """backbone="vits_dino" through the registry surface: construction, state_dict names, one semi-supervised style step with the
    fused optimiser (loss goes down on a fixed batch).  A 2-layer / 2-head ViT is substituted for speed."""
    from lightning_pose_amd.losses import LossFactory
    from lightning_pose_amd.models import get_model_class
    from lightning_pose_amd.models.backbones import factory as bf
    from lightning_pose_amd.trainer import Trainer

    dev = stack_backend
    monkeypatch.setitem(bf.VIT_CONFIGS, "vits_dino", (128, 2, 2, 256, 16, 3))
    monkeypatch.setitem(bf._IMPLEMENTED, "vits_dino", 128)
    cls = get_model_class("heatmap", False)
    K = 3
    model = cls(num_keypoints=K, loss_factory=LossFactory({"heatmap_mse": {"log_weight": 0.0}}, None), backbone="vits_dino",
                pretrained=False, torch_seed=0, device=dev, optimizer="AdamW", optimizer_params={"learning_rate": 1e-3})
    sd = model.state_dict()
    assert "backbone.vision_encoder.embeddings.cls_token" in sd and "backbone.vision_encoder.layers.1.mlp.fc2.weight" in sd
    assert "head.upsampling_layers.1.weight" in sd and sd["head.upsampling_layers.1.weight"].shape == (128 // 4, K, 3, 3)
    names = {n for n, _ in model.named_parameters()}
    assert "backbone.vision_encoder.embeddings.position_embeddings" in names
    gen = torch.Generator().manual_seed(0)
    from lightning_pose_amd import ops
    kp = torch.rand(2, K, 2, generator=gen) * 60 + 2
    batch = {"images": torch.randn(2, 3, 64, 64, generator=gen).to(dev), "keypoints": kp.reshape(2, -1).to(dev),
             "heatmaps": ops.generate_heatmaps(kp.to(dev), 64, 64, (16, 16)), "bbox": torch.tensor([[0.0, 0.0, 64.0, 64.0]] * 2).to(dev)}
    tr = Trainer(data_parallel=False)
    tr.setup(model)
    model.train()
    for g in model.optimizers().param_groups:  # unfreeze the backbone for this check
        g["lr"] = 1e-3
    losses = [float(tr.training_batch(model, batch, i)) for i in range(5)]
    assert all(l == l for l in losses) and losses[-1] < losses[0], losses


def test_gelu_backward_inside_fc2s_data_gradient_changes_no_bit(stack_backend, monkeypatch):
    """round 6: fc2's data gradient leaves as the gradient of fc1's OUTPUT (lp_gemm_nt_gelu_bwd) - every gradient equals the two-pass form
    (lp_gemm_nt, then lp_gelu_bwd_colsum) bit for bit, except fc1's bias gradient, whose column sums are now fixed-point totals"""
    from lightning_pose_amd.vit_engine import ViTEngine

    dev = stack_backend

    def grads(fused: bool):
        monkeypatch.setenv("LP_VIT_GELU_FUSED", "1" if fused else "0")
        torch.manual_seed(3)
        eng = ViTEngine(5, 2, dev, hidden=128, depth=2, heads=2, mlp=256, patch=16, pretrain_grid=3)
        sd = {k: torch.randn_like(v) * (0.05 if v.dim() > 1 else 0.1) + (1.0 if "layernorm" in k and k.endswith("weight") else 0.0)
              for k, v in eng.state_dict().items()}
        eng.load_state_dict(sd)
        x = torch.randn(2, 3, 64, 64, device=dev)
        heat, tape = eng.forward(x, training=True)
        eng.zero_grad()
        eng.backward(tape, torch.randn(heat.shape, generator=torch.Generator().manual_seed(4)).to(dev) * heat)
        return {k: v.detach().cpu().clone() for k, v in eng.grad_views().items()}

    a, b = grads(True), grads(False)
    assert a.keys() == b.keys()
    exact = 0
    for k in a:
        if k.endswith("mlp.fc1.bias"):   # (a sum of ~1e-3-sized terms that nearly cancel: the two-pass form rounds its fp32 partial sums)
            torch.testing.assert_close(a[k], b[k], atol=1e-2 * float(b[k].abs().max()), rtol=0)
            assert float(b[k].abs().max()) > 0
        elif a[k].dim() == 1:            # LayerNorm parameters, the other biases: fp32 atomics in arrival order - equal to their own run-to-run noise
            torch.testing.assert_close(a[k], b[k], atol=1e-6 * float(b[k].abs().max()), rtol=0)
        else:                            # every weight gradient (and so every tensor that flows through the backward pass)
            assert torch.equal(a[k], b[k]), k
            exact += 1
    assert exact >= 10
