"""Output contracts of ``get_loss_inputs_*`` (reference: lightning_pose/models/datatypes.py:40-57).
``models.factory._validate_loss_model_compatibility`` reads these annotations."""

from __future__ import annotations

from typing import TypedDict

import torch


class HeatmapTrackerLabeledOutputsDict(TypedDict):
    heatmaps_targ: torch.Tensor
    heatmaps_pred: torch.Tensor
    keypoints_targ: torch.Tensor
    keypoints_pred: torch.Tensor
    confidences: torch.Tensor


class HeatmapTrackerUnlabeledOutputsDict(TypedDict):
    heatmaps_pred: torch.Tensor
    keypoints_pred: torch.Tensor
    keypoints_pred_augmented: torch.Tensor
    confidences: torch.Tensor
