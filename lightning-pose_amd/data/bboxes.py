"""``model_to_frame_batch`` (reference: lightning_pose/data/bboxes.py:222-288, norm_to_frame :74-105)."""

from __future__ import annotations

import torch

from .. import ops


def batch_num_views(batch_dict: dict) -> int:
    """Number of camera views encoded in a batch (reference bboxes.py:254-271)."""
    if "num_views" in batch_dict and (int(batch_dict["num_views"].max()) > 1 or batch_dict.get("is_multiview", False)):
        unique = torch.unique(batch_dict["num_views"])
        if unique.numel() != 1:
            raise ValueError(f"each batch element must contain the same number of views; found elements with {unique} views")
        return int(unique)
    if batch_dict.get("is_multiview", False):
        return batch_dict["bbox"].shape[1] // 4
    return 1


def model_dims(batch_dict: dict) -> tuple[int, int]:
    img = batch_dict["images"] if "images" in batch_dict else batch_dict["frames"]
    return img.shape[-2], img.shape[-1]


def model_to_frame_batch(batch_dict: dict, model_keypoints: torch.Tensor, in_place: bool = True) -> torch.Tensor:
    """(B, 2K) keypoints in network-input px -> original-frame px using ``batch_dict['bbox']`` rows [x, y, h, w].

    Unlike the reference this never writes through its argument (the reference's default ``in_place=True`` rewrites
    the decode output through a reshape view - SURVEY.md section 8b "aliasing"); the returned values are identical.
    """
    mh, mw = model_dims(batch_dict)
    views = batch_num_views(batch_dict)
    k = model_keypoints.shape[1] // 2
    bbox = batch_dict["bbox"]
    if bbox.shape[0] != model_keypoints.shape[0]:  # context batch: no predictions for the first / last two frames (norm_to_frame :99-104)
        bbox = bbox[2:-2]
    fm = ops.DecodeFrameMap(None, False, bbox, views, mh, mw, k)
    return ops.frame_map_apply(model_keypoints, fm)
