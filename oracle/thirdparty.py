"""Pure-torch restatements of third-party arithmetic the reference calls on the hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference (paninski-lab/lightning-pose v2.4.0) does not vendor these; versions are
unpinned in its pyproject.toml (``kornia``, ``torchvision``).  None of them is installed in
the build image, so their *published* semantics are restated here and pinned against the
reference's own known-answer tests (tests/models/heads/test_heatmap.py:124-219 in the
reference tree), see tests/test_oracle_kats.py.

Call sites in the reference:
  kornia.geometry.subpix.spatial_softmax2d / spatial_expectation2d
        lightning_pose/models/heads/heatmap.py:126-127,211
  kornia.filters.filter2d + kornia...pyramid._get_pyramid_gaussian_kernel
        lightning_pose/models/heads/heatmap.py:93-99
  kornia.losses.kl_div_loss_2d / js_div_loss_2d
        lightning_pose/losses/losses.py:358,374-378,402,418-422
  torchvision.models.resnet50
        lightning_pose/models/backbones/factory.py:322
"""

from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# kornia
# --------------------------------------------------------------------------------------


def spatial_softmax2d(x: torch.Tensor, temperature: torch.Tensor | float = 1.0) -> torch.Tensor:
    """softmax over H*W of ``x * temperature`` (temperature MULTIPLIES)."""
    b, c, h, w = x.shape
    t = temperature if torch.is_tensor(temperature) else torch.tensor(float(temperature))
    flat = x.reshape(b, c, h * w) * t.to(device=x.device, dtype=x.dtype)
    return F.softmax(flat, dim=-1).reshape(b, c, h, w)


def spatial_expectation2d(p: torch.Tensor, normalized_coordinates: bool = True) -> torch.Tensor:
    """Expected (x, y) of a 2-D distribution; pixel-index grid when not normalised."""
    b, c, h, w = p.shape
    if normalized_coordinates:
        xs = torch.linspace(-1.0, 1.0, w, device=p.device, dtype=p.dtype)
        ys = torch.linspace(-1.0, 1.0, h, device=p.device, dtype=p.dtype)
    else:
        xs = torch.linspace(0.0, w - 1.0, w, device=p.device, dtype=p.dtype)
        ys = torch.linspace(0.0, h - 1.0, h, device=p.device, dtype=p.dtype)
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    flat = p.reshape(b, c, h * w)
    ex = (flat * gx.reshape(1, 1, -1)).sum(-1, keepdim=True)
    ey = (flat * gy.reshape(1, 1, -1)).sum(-1, keepdim=True)
    return torch.cat([ex, ey], dim=-1)


def get_pyramid_gaussian_kernel() -> torch.Tensor:
    """[1,5,5] binomial kernel outer([1,4,6,4,1])/256."""
    v = torch.tensor([1.0, 4.0, 6.0, 4.0, 1.0])
    return (torch.outer(v, v) / 256.0).unsqueeze(0)


def filter2d(x: torch.Tensor, kernel: torch.Tensor, border_type: str = "reflect") -> torch.Tensor:
    """Depthwise cross-correlation with a single [1,kh,kw] kernel, same-size output."""
    b, c, h, w = x.shape
    kh, kw = kernel.shape[-2:]
    ph, pw = kh // 2, kw // 2
    mode = {"constant": "constant", "reflect": "reflect", "replicate": "replicate",
            "circular": "circular"}[border_type]
    xp = F.pad(x, (pw, pw, ph, ph), mode=mode)
    wgt = kernel.to(x).reshape(1, 1, kh, kw).expand(c, 1, kh, kw)
    return F.conv2d(xp, wgt, groups=c)


def kl_div_loss_2d(pred: torch.Tensor, target: torch.Tensor, reduction: str = "mean") -> torch.Tensor:
    """KL(target || pred) summed over H,W per (b,c)."""
    val = (target * (torch.log(target) - torch.log(pred))).sum(dim=(-2, -1))
    return _reduce(val, reduction)


def js_div_loss_2d(pred: torch.Tensor, target: torch.Tensor, reduction: str = "mean") -> torch.Tensor:
    m = 0.5 * (pred + target)
    val = 0.5 * (target * (torch.log(target) - torch.log(m))).sum(dim=(-2, -1)) \
        + 0.5 * (pred * (torch.log(pred) - torch.log(m))).sum(dim=(-2, -1))
    return _reduce(val, reduction)


def _reduce(val: torch.Tensor, reduction: str) -> torch.Tensor:
    if reduction == "none":
        return val
    if reduction == "mean":
        return val.mean()
    if reduction == "sum":
        return val.sum()
    raise NotImplementedError(reduction)


# --------------------------------------------------------------------------------------
# torchvision ResNet-50 (Bottleneck v1.5: stride on the 3x3)
# --------------------------------------------------------------------------------------


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int = 1, downsample: nn.Module | None = None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        idt = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            idt = self.downsample(x)
        return self.relu(out + idt)


class ResNet50(nn.Module):
    """children order: conv1, bn1, relu, maxpool, layer1..4, avgpool, fc (as torchvision)."""

    def __init__(self, num_classes: int = 1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, 3, 1)
        self.layer2 = self._make_layer(128, 4, 2)
        self.layer3 = self._make_layer(256, 6, 2)
        self.layer4 = self._make_layer(512, 3, 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(2048, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0.0)

    def _make_layer(self, planes: int, blocks: int, stride: int) -> nn.Sequential:
        down = None
        if stride != 1 or self.inplanes != planes * 4:
            down = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * 4),
            )
        layers = [Bottleneck(self.inplanes, planes, stride, down)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(Bottleneck(self.inplanes, planes))
        return nn.Sequential(*layers)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


def resnet50(weights=None, **kwargs) -> ResNet50:
    if weights is not None:
        raise RuntimeError("no network in this environment: pretrained weights unavailable")
    return ResNet50(**kwargs)
