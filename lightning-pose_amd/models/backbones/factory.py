"""Backbone registry for the MI355X path (reference: lightning_pose/models/backbones/factory.py:98-168,170-236,238-348)."""

from __future__ import annotations

# The reference's ResNet-50 variants differ only in where their pretrained weights come from (torchvision ImageNet for "resnet50",
# mmpose checkpoints for the animal / human ones - reference :253-283; its shipped default is "resnet50_animal_ap10k",
# config_default.yaml:121): one architecture here, the weights arrive through ``backbone_checkpoint``.
RESNET50_VARIANTS = ("resnet50", "resnet50_animal_apose", "resnet50_animal_ap10k", "resnet50_human_jhmdb", "resnet50_human_res_rle",
                     "resnet50_human_top_res", "resnet50_human_hand")

BACKBONE_STRIDES: dict[str, int] = {**{n: 32 for n in RESNET50_VARIANTS}, "vits_dino": 16, "vitb_dino": 16}

# name -> number of output features
_IMPLEMENTED = {**{n: 2048 for n in RESNET50_VARIANTS}, "vits_dino": 384, "vitb_dino": 768}

# ViT variants: (hidden, depth, heads, mlp, patch, pretraining grid) of facebook/dino-vit{s,b}16
VIT_CONFIGS = {"vits_dino": (384, 12, 6, 1536, 16, 14), "vitb_dino": (768, 12, 12, 3072, 16, 14)}


def backbone_features(backbone_arch: str) -> int:
    if backbone_arch not in _IMPLEMENTED:
        raise ValueError(f'"{backbone_arch}" is not a valid backbone; allowed backbones: {sorted(_IMPLEMENTED)}')
    return _IMPLEMENTED[backbone_arch]
