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


class _HeadFn(torch.autograd.Function):
    """features -> heat-maps through HeadEngine.forward; backward returns the feature gradient and accumulates the layers' gradients
    into the engine's flat buffer (the tensors in ``param.grad`` are views of it)."""

    @staticmethod
    def forward(ctx, anchor: torch.Tensor, features: torch.Tensor, net):
        heat, tape = net.forward(features)
        ctx.net, ctx.tape = net, tape
        return heat

    @staticmethod
    def backward(ctx, g_heat: torch.Tensor):
        d = ctx.net.backward(ctx.tape, g_heat)
        ctx.tape = None
        return None, d, None


class _Layers(_Holder):
    """``upsampling_layers``: entry 0 is the PixelShuffle, entries 1 .. n name the ConvTranspose2d parameters (reference :20-71)."""

    def __init__(self, n_layers: int):
        super().__init__()
        self._n = n_layers + 1

    def __len__(self) -> int:
        return self._n


class HeatmapHead(_Holder):
    """Reference models/heads/heatmap.py:147-227, every constructor option.  Inside a tracker (``_bound=True``) the module only names
    the parameters of the tracker's engine; on its own it owns a ``HeadEngine`` and ``forward(features)`` runs the HIP head:
    PixelShuffle(2) -> ConvTranspose2d x n (``deconv_out_channels`` wide in between) -> spatial soft-max unless ``final_softmax=False``."""

    def __init__(self, backbone_arch: str, in_channels: int, out_channels: int, deconv_out_channels: int | None = None,
                 downsample_factor: int = 2, final_softmax: bool = True, *, _bound: bool = False,
                 device: torch.device | str | None = None) -> None:
        super().__init__()
        self.backbone_arch = backbone_arch
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.deconv_out_channels = deconv_out_channels
        self.downsample_factor = downsample_factor
        self.final_softmax = final_softmax
        self.temperature = torch.tensor(1000.0)  # soft-argmax temperature (reference :187)
        stride = BACKBONE_STRIDES.get(backbone_arch, 32)
        self.n_layers = int(math.log2(stride)) - downsample_factor - 1
        self.upsampling_layers = _Layers(self.n_layers)
        self.net = None
        if _bound:
            if deconv_out_channels not in (None, out_channels) or not final_softmax:
                raise NotImplementedError("the trackers build the default HeatmapHead configuration (reference heatmap_tracker.py:89-94)")
            return
        import os

        from ...engine import HeadEngine

        dev = torch.device(device) if device is not None else torch.device(f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}")
        self.net = HeadEngine(in_channels, stride, out_channels, downsample_factor, deconv_out_channels, final_softmax, dev)
        init = {}
        for c in self.net.plan.head:   # initialize_upsampling_layers (reference :74-83): xavier_uniform(gain 0.01) weights, zero biases
            w = torch.empty(c.cin, c.cout, 3, 3)
            nn.init.xavier_uniform_(w, gain=0.01)
            init[f"{c.name}.weight"], init[f"{c.name}.bias"] = w, torch.zeros(c.cout)
        self.net.load_state_dict(init)
        for c in self.net.plan.head:
            mod = _Holder()
            self.upsampling_layers.add_module(c.name.rsplit(".", 1)[1], mod)
            for leaf in ("weight", "bias"):
                p = nn.Parameter(self.net.param_view(c, leaf), requires_grad=True)
                p.grad = self.net.param_view(c, leaf, buf=self.net.G)
                mod.register_parameter(leaf, p)
        self._anchor = torch.zeros(1, device=dev, requires_grad=True)
        self._stale = False

    def forward(self, features: torch.Tensor) -> torch.Tensor:
        """(B, in_channels, h, w) -> (B, out_channels, h 2^(n+1), w 2^(n+1)) (reference :203-212)."""
        if self.net is None:
            raise RuntimeError("this head names the parameters of its tracker's engine; call the tracker")
        self.net.refresh_weight_copies()   # (the fp32 masters may have been stepped or loaded since the last call: a few small tensors)
        if torch.is_grad_enabled():
            return _HeadFn.apply(self._anchor, features, self.net)
        return self.net.forward(features)[0]

    def _apply(self, fn, *a, **k):  # parameters are views of device-resident flat buffers: moving them would detach them
        if self.net is None:
            return super()._apply(fn, *a, **k)
        probe = fn(torch.zeros(1, device=self.net.device))
        if probe.device != self.net.device or probe.dtype != torch.float32:
            raise NotImplementedError(f"this head lives on {self.net.device} in fp32 master precision; construct it with device=...")
        return self

    def run_subpixelmaxima(self, heatmaps: torch.Tensor, frame_map: "ops.DecodeFrameMap | None" = None):
        return run_subpixelmaxima(heatmaps, self.downsample_factor, self.temperature, frame_map)
