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
