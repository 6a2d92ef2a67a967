"""``undo_affine_transform_batch`` (reference: lightning_pose/data/utils.py:142-234) on the lp_hip frame-map kernel."""

from __future__ import annotations

import math

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


def split_sizes_from_probabilities(total_number: int, train_probability: float, val_probability: float | None = None,
                                   test_probability: float | None = None) -> list[int]:
    """[train, val, test] example counts (reference data/utils.py:17-64): unspecified val / test share the remainder equally (rounded to 5
    decimals), counts are floored, what flooring loses goes to the test set - or to the training set when fewer than 5 examples are left -
    and an empty validation set takes one example from the training set."""
    if test_probability is None and val_probability is None:
        rest = 1.0 - train_probability
        val_probability = test_probability = round(rest / 2, 5)
    elif test_probability is None:
        test_probability = 1.0 - train_probability - val_probability
    if test_probability + train_probability + val_probability != 1.0:
        raise AssertionError("split probabilities must add to one")
    train_number = int(math.floor(train_probability * total_number))
    val_number = int(math.floor(val_probability * total_number))
    leftover = total_number - train_number - val_number
    if leftover < 5:
        train_number, test_number = train_number + leftover, 0
    else:
        test_number = leftover
    if val_number == 0:  # at least one validation example (reference :63-68)
        train_number -= 1
        val_number += 1
        if train_number < 1:
            raise ValueError("Must have at least two labeled frames, one train and one validation")
    assert train_number + val_number + test_number == total_number
    return [train_number, val_number, test_number]


def compute_num_train_frames(len_train_dataset: int, train_frames: int | float | None = None) -> int:
    """``train_frames`` > 1: that many frames; in (0, 1): that fraction; 1, None or more than available: all (reference :104-139)."""
    if train_frames is None or train_frames >= len_train_dataset or train_frames == 1:
        return len_train_dataset
    if train_frames > 1:
        return int(train_frames)
    if train_frames > 0:
        return int(train_frames * len_train_dataset)
    raise ValueError("train_frames must be >0")
