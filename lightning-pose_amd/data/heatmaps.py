"""Heat-map target generation (reference: lightning_pose/data/heatmaps.py:11-87) on the lp_hip kernels.

``evaluate_heatmaps_at_location`` (reference :90-142) is exported in its standalone form; inside the training step the same
5x5 confidence window is the epilogue of the fused decode kernel (``lightning_pose_amd.ops.decode``), which is where
the reference calls it (models/heads/heatmap.py:129).
"""

from __future__ import annotations

import math

import torch

from .. import ops


def generate_heatmaps(
    keypoints: torch.Tensor,
    height: int,
    width: int,
    output_shape: tuple[int, int],
    sigma: float = 1.25,
    keep_gradients: bool = False,
    visibility: torch.Tensor | None = None,
) -> torch.Tensor:
    """2-D Gaussian targets, (B, K, 2) image-px keypoints -> (B, K, h, w).  Same semantics as the reference:
    sigma in heat-map px, maps normalised to sum 1, NaN / out-of-bounds -> zeros, visibility 0 -> zeros,
    1 -> uniform, 2 -> Gaussian."""
    if keep_gradients and keypoints.requires_grad:   # reference :37-40: the keypoints stay attached (lp_heatmap_gen_bwd)
        return ops.generate_heatmaps_with_grad(keypoints, height, width, tuple(output_shape), sigma, visibility)
    return ops.generate_heatmaps(keypoints, height, width, tuple(output_shape), sigma, visibility)


def evaluate_heatmaps_at_location(heatmaps: torch.Tensor, locs: torch.Tensor, sigma: float = 1.25, num_stds: int = 2) -> torch.Tensor:
    """(B, K, h, w) heat-maps and (B, K, 2) = (x, y) locations -> (B, K) confidence: the sum of all pixels within
    ``floor(sigma * num_stds)`` of ``int64(loc)``, the map zero-padded by that margin (reference :90-142)."""
    return ops.heatmap_confidence(heatmaps, locs.to(torch.float32), int(math.floor(sigma * num_stds)))
