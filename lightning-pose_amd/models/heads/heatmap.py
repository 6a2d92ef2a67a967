"""HeatmapHead (reference: lightning_pose/models/heads/heatmap.py:147-227): parameter holder for the
PixelShuffle + ConvTranspose2d stack executed by ``lightning_pose_amd.engine.Engine`` and owner of the fused decode."""

from __future__ import annotations

import math

import torch
from torch import nn

from ... import ops
from ..backbones import BACKBONE_STRIDES


class _Holder(nn.Module):
    """Names parameters/buffers that live in the engine's flat buffers; it has no arithmetic of its own."""

    def forward(self, *args, **kwargs):  # pragma: no cover
        raise RuntimeError("this module only names parameters; the network runs in lightning_pose_amd.engine.Engine")


def run_subpixelmaxima(heatmaps: torch.Tensor, downsample_factor: int, temperature: torch.Tensor | float,
                       frame_map: "ops.DecodeFrameMap | None" = None):
    """Soft-argmax decode (reference :103-144) -> (keypoints (B, 2K) model px, confidences (B, K)).

    One fused kernel: up-sampling, softmax(T), expectation, 5x5 confidence and the sub-pixel offset.
    """
    b, k, h, w = heatmaps.shape
    if frame_map is None:
        frame_map = ops.DecodeFrameMap(None, False, None, 1, h << downsample_factor, w << downsample_factor, k)
    kp_aug, _kp_frame, conf = ops.decode(heatmaps, downsample_factor, float(temperature), frame_map)
    return kp_aug, conf


class HeatmapHead(_Holder):
    def __init__(self, backbone_arch: str, in_channels: int, out_channels: int, deconv_out_channels: int | None = None,
                 downsample_factor: int = 2, final_softmax: bool = True) -> None:
        super().__init__()
        if deconv_out_channels not in (None, out_channels) or not final_softmax:
            raise NotImplementedError("the MI355X head implements the default HeatmapHead configuration")
        self.backbone_arch = backbone_arch
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.downsample_factor = downsample_factor
        self.final_softmax = final_softmax
        self.temperature = torch.tensor(1000.0)  # soft-argmax temperature (reference :187)
        stride = BACKBONE_STRIDES.get(backbone_arch, 32)
        self.n_layers = int(math.log2(stride)) - downsample_factor - 1
        self.upsampling_layers = _Holder()

    def run_subpixelmaxima(self, heatmaps: torch.Tensor, frame_map: "ops.DecodeFrameMap | None" = None):
        return run_subpixelmaxima(heatmaps, self.downsample_factor, self.temperature, frame_map)
