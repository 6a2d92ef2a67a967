"""Backbone registry for the MI355X path (reference: lightning_pose/models/backbones/factory.py:98-168,238-348)."""

from __future__ import annotations

BACKBONE_STRIDES: dict[str, int] = {"resnet50": 32, "vits_dino": 16}

# name -> number of output features.  ViT-S/16 ("vits_dino") is the next row of the scope table.
_IMPLEMENTED = {"resnet50": 2048}


def backbone_features(backbone_arch: str) -> int:
    if backbone_arch not in _IMPLEMENTED:
        if backbone_arch in BACKBONE_STRIDES:
            raise NotImplementedError(f'backbone "{backbone_arch}" is on the roadmap of the MI355X path but not built yet')
        raise ValueError(f'"{backbone_arch}" is not a valid backbone; allowed backbones: {sorted(_IMPLEMENTED)}')
    return _IMPLEMENTED[backbone_arch]
