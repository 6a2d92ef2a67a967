"""Backbone registry for the MI355X path (reference: lightning_pose/models/backbones/factory.py:98-168,170-236,238-348)."""

from __future__ import annotations

BACKBONE_STRIDES: dict[str, int] = {"resnet50": 32, "vits_dino": 16, "vitb_dino": 16}

# name -> number of output features
_IMPLEMENTED = {"resnet50": 2048, "vits_dino": 384, "vitb_dino": 768}

# ViT variants: (hidden, depth, heads, mlp, patch, pretraining grid) of facebook/dino-vit{s,b}16
VIT_CONFIGS = {"vits_dino": (384, 12, 6, 1536, 16, 14), "vitb_dino": (768, 12, 12, 3072, 16, 14)}


def backbone_features(backbone_arch: str) -> int:
    if backbone_arch not in _IMPLEMENTED:
        raise ValueError(f'"{backbone_arch}" is not a valid backbone; allowed backbones: {sorted(_IMPLEMENTED)}')
    return _IMPLEMENTED[backbone_arch]
