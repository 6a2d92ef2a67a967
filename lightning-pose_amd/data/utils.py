"""``undo_affine_transform_batch`` (reference: lightning_pose/data/utils.py:142-234) on the lp_hip frame-map kernel."""

from __future__ import annotations

import torch

from .. import ops


def undo_affine_transform_batch(keypoints_augmented: torch.Tensor, transforms: torch.Tensor, is_multiview: bool = False) -> torch.Tensor:
    """Undo the augmentation affine on (S, 2K) keypoints.  ``transforms``: (2,3) single, (S,2,3) per frame,
    (V,2,3) per view when ``is_multiview``; any tensor whose last dim is not 3 is the "no augmentation" sentinel,
    in which case the input tensor itself is returned (as the reference does, data/utils.py:231-232)."""
    if transforms.shape[-1] != 3:
        return keypoints_augmented
    k = keypoints_augmented.shape[1] // 2
    views = transforms.shape[0] if is_multiview else 1
    fm = ops.DecodeFrameMap(transforms, is_multiview, None, views, 1, 1, k)
    return ops.frame_map_apply(keypoints_augmented, fm)


def undo_affine_transform(keypoints: torch.Tensor, transform: torch.Tensor) -> torch.Tensor:
    """(S, K, 2) keypoints and one (2, 3) matrix or one per frame (S, 2, 3) -> the un-augmented (S, K, 2) keypoints
    (reference data/utils.py:142-188); same kernel as the batch form."""
    s, k = keypoints.shape[0], keypoints.shape[1]
    return undo_affine_transform_batch(keypoints.reshape(s, 2 * k), transform, is_multiview=False).reshape(s, k, 2)
