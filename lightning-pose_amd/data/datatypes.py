"""Batch-dict contracts consumed by the training step (reference: lightning_pose/data/datatypes.py:163-250)."""

from __future__ import annotations

from typing import TypedDict

import torch


class HeatmapLabeledBatchDict(TypedDict):
    images: torch.Tensor      # (B, 3, H, W) or (B, V, 3, H, W)
    keypoints: torch.Tensor   # (B, 2K)
    heatmaps: torch.Tensor    # (B, K, h, w)
    bbox: torch.Tensor        # (B, 4) or (B, 4V)   rows [x, y, h, w]
    idxs: torch.Tensor


class UnlabeledBatchDict(TypedDict):
    frames: torch.Tensor      # (S, 3, H, W)
    transforms: torch.Tensor  # (2, 3) | (S, 2, 3) | (1,) sentinel
    bbox: torch.Tensor        # (S, 4)
    is_multiview: bool


class MultiviewUnlabeledBatchDict(TypedDict):
    frames: torch.Tensor      # (S, V, 3, H, W)
    transforms: torch.Tensor  # (V, 2, 3) | (V, 1, 1)
    bbox: torch.Tensor        # (S, 4V)
    is_multiview: bool


class SemiSupervisedHeatmapBatchDict(TypedDict):
    labeled: HeatmapLabeledBatchDict
    unlabeled: UnlabeledBatchDict
