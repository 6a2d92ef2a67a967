"""Seeded parameter initialisation identical to the reference's (CPU, once, at construction).

The reference seeds torch (models/heatmap_tracker.py:69-70), builds ``torchvision.models.resnet50(weights=None)``
(models/backbones/factory.py:322) and then the head (models/heads/heatmap.py:20-83).  To reproduce its weights bit
for bit the same number of RNG draws must happen in the same order, so this module instantiates bare torch layers of
the same shapes in the same order (no forward is ever run on them) and returns their tensors under the reference's
``state_dict`` keys.  torchvision semantics: SURVEY.md Appendix A.
"""

from __future__ import annotations

import torch
import torch.nn as nn


def seeded_state_dict(num_keypoints: int, n_head_layers: int) -> dict[str, torch.Tensor]:
    convs: list[tuple[str, nn.Conv2d]] = []
    bns: list[tuple[str, int]] = []

    def conv(name, cin, cout, k, stride=1, pad=0):
        m = nn.Conv2d(cin, cout, k, stride=stride, padding=pad, bias=False)  # default init draws from the RNG
        return name, m

    stem = conv("backbone.0", 3, 64, 7, 2, 3)
    ordered: list[tuple[str, nn.Conv2d]] = [stem]       # kaiming order = module traversal order
    bns.append(("backbone.1", 64))
    inplanes = 64
    for li, (planes, nblk, stride) in enumerate(((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))):
        for bi in range(nblk):
            pre = f"backbone.{4 + li}.{bi}"
            st = stride if bi == 0 else 1
            down = None
            if bi == 0 and (st != 1 or inplanes != planes * 4):
                down = conv(f"{pre}.downsample.0", inplanes, planes * 4, 1, st)   # constructed before the block
            c1 = conv(f"{pre}.conv1", inplanes, planes, 1)
            c2 = conv(f"{pre}.conv2", planes, planes, 3, st, 1)
            c3 = conv(f"{pre}.conv3", planes, planes * 4, 1)
            ordered += [c1, c2, c3]
            bns += [(f"{pre}.bn1", planes), (f"{pre}.bn2", planes), (f"{pre}.bn3", planes * 4)]
            if down is not None:
                ordered.append(down)                                              # but traversed after conv3/bn3
                bns.append((f"{pre}.downsample.1", planes * 4))
            inplanes = planes * 4
    nn.Linear(2048, 1000)  # resnet.fc: dropped by the tracker, but its default init advances the RNG
    for _, m in ordered:
        nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    sd: dict[str, torch.Tensor] = {}
    for name, m in ordered:
        sd[f"{name}.weight"] = m.weight.detach()
    for name, c in bns:
        sd[f"{name}.weight"] = torch.ones(c)
        sd[f"{name}.bias"] = torch.zeros(c)
        sd[f"{name}.running_mean"] = torch.zeros(c)
        sd[f"{name}.running_var"] = torch.ones(c)

    # head: construct every ConvTranspose2d first, then xavier-initialise them in order (reference :43-83)
    layers = []
    cin = 2048 // 4
    for _ in range(n_head_layers):
        layers.append(nn.ConvTranspose2d(cin, num_keypoints, 3, stride=2, padding=1, output_padding=1))
        cin = num_keypoints
    for i, ct in enumerate(layers):
        nn.init.xavier_uniform_(ct.weight, gain=0.01)
        nn.init.zeros_(ct.bias)
        sd[f"head.upsampling_layers.{i + 1}.weight"] = ct.weight.detach()
        sd[f"head.upsampling_layers.{i + 1}.bias"] = ct.bias.detach()
    return sd


def head_state_dict(in_channels: int, num_keypoints: int, n_head_layers: int) -> dict[str, torch.Tensor]:
    """HeatmapHead initialisation alone (reference models/heads/heatmap.py:43-83): every ConvTranspose2d constructed first,
    then xavier-uniform(gain 0.01) weights and zero biases in order."""
    layers = []
    cin = in_channels // 4
    for _ in range(n_head_layers):
        layers.append(nn.ConvTranspose2d(cin, num_keypoints, 3, stride=2, padding=1, output_padding=1))
        cin = num_keypoints
    sd: dict[str, torch.Tensor] = {}
    for i, ct in enumerate(layers):
        nn.init.xavier_uniform_(ct.weight, gain=0.01)
        nn.init.zeros_(ct.bias)
        sd[f"head.upsampling_layers.{i + 1}.weight"] = ct.weight.detach()
        sd[f"head.upsampling_layers.{i + 1}.bias"] = ct.bias.detach()
    return sd


def vit_seeded_state_dict(hidden: int, depth: int, heads: int, mlp: int, patch: int, grid: int) -> dict[str, torch.Tensor]:
    """Random ViT weights under the reference's ``backbone.vision_encoder.*`` names.  The reference only ever loads pretrained
    weights (``ViTModel.from_pretrained``, models/backbones/vit.py:27); without network access ``pretrained=False`` uses the
    initialisation of ``transformers.ViTModel(config)`` (truncated normal 0.02 / zeros / ones) - through transformers when it is
    installed, so the values are what that constructor would give under the current seed."""
    try:
        from transformers import ViTConfig, ViTModel
        cfg = ViTConfig(hidden_size=hidden, num_hidden_layers=depth, num_attention_heads=heads, intermediate_size=mlp,
                        image_size=patch * grid, patch_size=patch)
        sd = ViTModel(cfg, add_pooling_layer=False).state_dict()
        return {f"backbone.vision_encoder.{k}": v.detach() for k, v in sd.items()}
    except ImportError:
        pre = "backbone.vision_encoder"
        tn = lambda *shape: nn.init.trunc_normal_(torch.empty(*shape), std=0.02)  # noqa: E731
        sd = {f"{pre}.embeddings.cls_token": tn(1, 1, hidden), f"{pre}.embeddings.position_embeddings": tn(1, 1 + grid * grid, hidden),
              f"{pre}.embeddings.patch_embeddings.projection.weight": tn(hidden, 3, patch, patch),
              f"{pre}.embeddings.patch_embeddings.projection.bias": torch.zeros(hidden)}
        for i in range(depth):
            p = f"{pre}.layers.{i}"
            for nm, (n, k) in (("attention.q_proj", (hidden, hidden)), ("attention.k_proj", (hidden, hidden)), ("attention.v_proj", (hidden, hidden)),
                               ("attention.o_proj", (hidden, hidden)), ("mlp.fc1", (mlp, hidden)), ("mlp.fc2", (hidden, mlp))):
                sd[f"{p}.{nm}.weight"], sd[f"{p}.{nm}.bias"] = tn(n, k), torch.zeros(n)
            for nm in ("layernorm_before", "layernorm_after"):
                sd[f"{p}.{nm}.weight"], sd[f"{p}.{nm}.bias"] = torch.ones(hidden), torch.zeros(hidden)
        sd[f"{pre}.layernorm.weight"], sd[f"{pre}.layernorm.bias"] = torch.ones(hidden), torch.zeros(hidden)
        return sd
