"""HeatmapTracker / SemiSupervisedHeatmapTracker (reference: lightning_pose/models/heatmap_tracker.py:34-286) on the
MI355X engine.

Same constructor arguments, methods, attributes and ``state_dict`` keys as the reference classes;
what differs is underneath: ``self.backbone`` / ``self.head`` only *name* parameters that live in the engine's flat
buffers, ``forward`` runs the hand-written HIP trunk (one autograd node for the whole network), and keypoints come out
of one fused decode kernel that also applies the affine-undo and bounding-box maps.
"""

from __future__ import annotations

import os
from typing import Any, Literal

import torch
from torch import nn

from .. import ops
from ..data.bboxes import batch_num_views, model_dims, model_to_frame_batch
from ..engine import Engine
from ..losses.losses import RegressionRMSELoss
from .backbones import backbone_features
from .backbones._init import head_state_dict, seeded_state_dict, vit_seeded_state_dict
from .backbones.factory import VIT_CONFIGS
from .base import BaseSupervisedTracker, SemiSupervisedTrackerMixin
from .datatypes import HeatmapTrackerLabeledOutputsDict, HeatmapTrackerUnlabeledOutputsDict
from .heads.heatmap import HeatmapHead, _Holder


class _NetworkFn(torch.autograd.Function):
    """images -> heat-maps through Engine.forward; backward hands the heat-map gradient to Engine.backward, which
    accumulates parameter gradients straight into the flat gradient buffer (the tensors in ``param.grad`` are views of it)."""

    @staticmethod
    def forward(ctx, anchor: torch.Tensor, images: torch.Tensor, net: Engine, training: bool, images2: torch.Tensor | None = None):
        # images2: a second batch sharing the pass (joint labeled + unlabeled forward, two BatchNorm segments)
        heat, tape = net.forward(images if images2 is None else (images, images2), training=training)
        ctx.net, ctx.tape = net, tape
        return heat

    @staticmethod
    def backward(ctx, g_heat: torch.Tensor):
        ctx.net.backward(ctx.tape, g_heat)
        ctx.tape = None
        return None, None, None, None, None


def _default_device() -> torch.device:
    return torch.device(f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}")


class HeatmapTracker(BaseSupervisedTracker):
    """Base model that produces heatmaps of keypoints from images."""

    def __init__(self, num_keypoints: int, num_targets: int | None = None, loss_factory: Any = None, backbone: str = "resnet50",
                 downsample_factor: Literal[1, 2, 3] = 2, pretrained: bool = True, torch_seed: int = 123, optimizer: str = "Adam",
                 optimizer_params: Any = None, lr_scheduler: str = "multisteplr", lr_scheduler_params: Any = None, **kwargs: Any) -> None:
        self.torch_seed = torch_seed
        torch.manual_seed(torch_seed)  # reproducible weight initialisation, as the reference (:69-70)
        super().__init__(optimizer=optimizer, optimizer_params=optimizer_params, lr_scheduler=lr_scheduler,
                         lr_scheduler_params=lr_scheduler_params)
        self.backbone_arch = backbone
        self.num_fc_input_features = backbone_features(backbone)
        self.do_context = bool(kwargs.get("do_context", False))
        if self.do_context:
            raise NotImplementedError("context (MHCRNN) models are outside the MI355X heatmap-tracker path")
        self.num_keypoints = num_keypoints
        self.num_targets = num_keypoints * 2 if num_targets is None else num_targets
        self.downsample_factor = downsample_factor

        device = torch.device(kwargs.get("device") or _default_device())
        # "bf16-mixed" (the product path) or "fp32" (validation against the fp32-only reference, train.py:411-428); LP_PRECISION overrides
        self.precision = {"32": "fp32", "32-true": "fp32", "fp32": "fp32"}.get(
            str(os.environ.get("LP_PRECISION") or kwargs.get("precision") or "bf16-mixed"), "bf16-mixed")
        self.head = HeatmapHead(backbone_arch=backbone, in_channels=self.num_fc_input_features, out_channels=num_keypoints,
                                downsample_factor=downsample_factor, _bound=True)
        self.backbone = _Holder()
        checkpoint = kwargs.get("backbone_checkpoint")
        if backbone in VIT_CONFIGS:
            from ..vit_engine import ViTEngine
            hidden, depth, heads, mlp, patch, grid = VIT_CONFIGS[backbone]
            engine_cls = ViTEngine
            if self.precision == "fp32":  # validation mode (vit_engine_fp32.py)
                from ..vit_engine_fp32 import Fp32ViTEngine as engine_cls
            self.net = engine_cls(num_keypoints, downsample_factor, device, hidden=hidden, depth=depth, heads=heads, mlp=mlp, patch=patch,
                                 pretrain_grid=grid)
            init = vit_seeded_state_dict(hidden, depth, heads, mlp, patch, grid)
            init.update(head_state_dict(self.num_fc_input_features, num_keypoints, self.head.n_layers))
            if pretrained:
                if checkpoint is None:
                    raise RuntimeError("pretrained=True needs the DINO weights, which cannot be downloaded here; pass "
                                       "backbone_checkpoint=<state_dict / safetensors file of facebook/dino-vits16> or pretrained=False")
                if str(checkpoint).endswith(".safetensors"):
                    import safetensors.torch
                    hf = safetensors.torch.load_file(checkpoint, device="cpu")
                else:
                    hf = torch.load(checkpoint, map_location="cpu")
                    hf = hf.get("state_dict", hf)
                for k, v in hf.items():
                    k = k[len("vit."):] if k.startswith("vit.") else k
                    key = ViTEngine.canonical_key(f"backbone.vision_encoder.{k}")
                    if key in init and init[key].shape == v.shape:
                        init[key] = v
        else:
            if self.precision == "fp32":  # validation mode: every activation and contraction in fp32 (engine_fp32.py)
                from ..engine_fp32 import Fp32Engine
                self.net = Fp32Engine(num_keypoints, downsample_factor, device)
            else:
                self.net = Engine(num_keypoints, downsample_factor, device)
            init = seeded_state_dict(num_keypoints, self.head.n_layers)
            if pretrained:
                if checkpoint is None:
                    raise RuntimeError("pretrained=True needs ImageNet weights, which cannot be downloaded here; pass "
                                       "backbone_checkpoint=<state_dict file with torchvision resnet50 keys, or the mmpose checkpoint of "
                                       "the resnet50_animal_* / resnet50_human_* variants> or pretrained=False")
                tv = torch.load(checkpoint, map_location="cpu")
                tv = tv.get("state_dict", tv)
                names = {"conv1": "backbone.0", "bn1": "backbone.1", "layer1": "backbone.4", "layer2": "backbone.5",
                         "layer3": "backbone.6", "layer4": "backbone.7"}
                for k, v in tv.items():
                    if k.startswith("backbone."):  # mmpose checkpoints (resnet50_animal_* / resnet50_human_*, reference :262-266)
                        k = k[len("backbone."):]
                    top, _, rest = k.partition(".")
                    if top in names and f"{names[top]}.{rest}" in init:
                        init[f"{names[top]}.{rest}"] = v
        if kwargs.get("residual_fp32") is not None and hasattr(self.net, "residual_fp32"):
            self.net.residual_fp32 = bool(kwargs["residual_fp32"])   # (optional policy; default: LP_RESIDUAL_FP32, off)
        self.net.load_state_dict(init, strict=False)
        self._bind_parameters()
        self._anchor = torch.zeros(1, device=device, requires_grad=True)

        self.loss_factory = loss_factory
        self.rmse_loss = RegressionRMSELoss()
        self.save_hyperparameters(ignore=["loss_factory", "loss_factory_unsupervised"])

    # ------------------------------------------------------------------------------------------------ plumbing
    def _bind_parameters(self) -> None:
        """Expose the engine's flat buffers as nn.Parameters / buffers under the reference's state_dict names."""
        net = self.net
        grads = {k: v for k, v in self._grad_views().items()}
        for key, view in net.state_dict().items():
            path = key.split(".")
            mod: nn.Module = self
            for part in path[:-1]:
                if not hasattr(mod, part):
                    mod.add_module(part, _Holder())
                mod = getattr(mod, part)
            leaf = path[-1]
            if key in grads:  # trainable: weights, biases, the ViT [CLS] token and position table
                p = nn.Parameter(view, requires_grad=True)
                p.grad = grads[key]
                mod.register_parameter(leaf, p)
            else:
                mod.register_buffer(leaf, view)

    def _grad_views(self) -> dict[str, torch.Tensor]:
        net, out = self.net, {}
        if hasattr(net, "grad_views"):
            return net.grad_views()
        for c in net.plan.convs:
            out[f"{c.name}.weight"] = net.param_view(c, "weight", buf=net.G)
            if c.kind == "convT":
                out[f"{c.name}.bias"] = net.param_view(c, "bias", buf=net.G)
        for b in net.plan.bns:
            out[f"{b.name}.weight"] = net.param_view(b, "weight", buf=net.G)
            out[f"{b.name}.bias"] = net.param_view(b, "bias", buf=net.G)
        return out

    def load_state_dict(self, state_dict, strict: bool = True, **kw):  # type: ignore[override]
        result = super().load_state_dict(state_dict, strict=strict, **kw)
        self.net.refresh_weight_copies()
        return result

    def _apply(self, fn, *a, **k):  # parameters are views of device-resident flat buffers: moving them would detach them
        probe = fn(torch.zeros(1, device=self.net.device))
        if probe.device != self.net.device or probe.dtype != torch.float32:
            raise NotImplementedError(f"this model lives on {self.net.device} in fp32 master precision; construct it with "
                                      "device=... instead of moving/casting it")
        return self

    @property
    def device(self) -> torch.device:  # type: ignore[override]
        return self.net.device

    # ------------------------------------------------------------------------------------------------ forward
    def joint_forward(self, images_a: torch.Tensor, images_b: torch.Tensor) -> bool:
        """Run the labeled and the unlabeled images of a semi-supervised step through the network in ONE pass (every layer one
        launch over both batches, which keep their own BatchNorm statistics as in the reference's two forward calls,
        models/base.py:682-695) and park the two heat-map stacks for the ``forward`` calls that follow.  False (nothing done) when
        the engine cannot split the pass on tile boundaries, the image sizes differ, or LP_JOINT_FORWARD=0."""
        if os.environ.get("LP_JOINT_FORWARD", "1") == "0" or not (self.training and torch.is_grad_enabled()):
            return False
        sa, sb = images_a.shape, images_b.shape
        if sa[-3:] != sb[-3:] or not hasattr(self.net, "can_segment"):
            return False
        xa, xb = images_a.reshape(-1, *sa[-3:]), images_b.reshape(-1, *sb[-3:])
        if not self.net.can_segment(xa.shape[0], sa[-2], sa[-1]):
            return False
        heat = _NetworkFn.apply(self._anchor, xa, self.net, True, xb)
        ha, hb = heat[:xa.shape[0]], heat[xa.shape[0]:]
        if len(sa) > 4:
            ha = ha.reshape(sa[0], -1, heat.shape[-2], heat.shape[-1])
        if len(sb) > 4:
            hb = hb.reshape(sb[0], -1, heat.shape[-2], heat.shape[-1])
        self._joint = {id(images_a): ha, id(images_b): hb}
        return True

    def forward(self, images: torch.Tensor) -> torch.Tensor:
        """(B,3,H,W) -> (B,K,h,w); (B,V,3,H,W) -> (B,K*V,h,w) (reference :107-133)."""
        parked = getattr(self, "_joint", None)
        if parked and id(images) in parked:
            return parked.pop(id(images))
        shape = images.shape
        x = images.reshape(-1, shape[-3], shape[-2], shape[-1]) if len(shape) > 4 else images
        if torch.is_grad_enabled():
            heat = _NetworkFn.apply(self._anchor, x, self.net, self.training)
        elif not self.training and hasattr(self.net, "forward_infer"):
            heat = self.net.forward_infer(x)  # eval + no_grad (predict / validation): BatchNorm folded, nothing kept
        else:
            heat, _ = self.net.forward(x, training=self.training)
        if len(shape) > 4:
            heat = heat.reshape(shape[0], -1, heat.shape[-2], heat.shape[-1])
        return heat

    def _decode(self, heat: torch.Tensor, batch_dict: dict, transforms: torch.Tensor | None, is_multiview: bool):
        mh, mw = model_dims(batch_dict)
        views = batch_num_views(batch_dict)
        fm = ops.DecodeFrameMap(transforms, is_multiview, batch_dict["bbox"], views, mh, mw, heat.shape[1])
        if getattr(self, "_decode_prune", None) is None:
            self._decode_prune = ops._DecodePruneAuto()   # this model's own plain / pruned decode choice (made from ITS maps)
        return ops.decode(heat, self.downsample_factor, float(self.head.temperature), fm, self._decode_prune)

    def get_loss_inputs_labeled(self, batch_dict: dict) -> HeatmapTrackerLabeledOutputsDict:
        predicted_heatmaps = self.forward(batch_dict["images"])
        _kp_model, predicted_keypoints, confidence = self._decode(predicted_heatmaps, batch_dict, None, False)
        target_keypoints = model_to_frame_batch(batch_dict, batch_dict["keypoints"])
        return {
            "heatmaps_targ": batch_dict["heatmaps"],
            "heatmaps_pred": predicted_heatmaps,
            "keypoints_targ": target_keypoints,
            "keypoints_pred": predicted_keypoints,
            "confidences": confidence,
        }

    def predict_step(self, batch_dict: dict, batch_idx: int, return_heatmaps: bool | None = False):
        images = batch_dict["images"] if "images" in batch_dict else batch_dict["frames"]
        predicted_heatmaps = self.forward(images)
        _kp_model, predicted_keypoints, confidence = self._decode(predicted_heatmaps, batch_dict, None, False)
        if return_heatmaps:
            return predicted_keypoints, confidence, predicted_heatmaps
        return predicted_keypoints, confidence

    def get_parameters(self) -> list[dict]:
        """Group order matters: UnfreezeBackbone requires [backbone, head] (reference :193-205)."""
        return [
            {"params": list(self.backbone.parameters()), "lr": 0, "name": "backbone"},
            {"params": list(self.head.parameters()), "name": "head"},
        ]


class SemiSupervisedHeatmapTracker(SemiSupervisedTrackerMixin, HeatmapTracker):
    """Model produces heatmaps of keypoints from labeled/unlabeled images."""

    def __init__(self, num_keypoints: int, loss_factory: Any = None, loss_factory_unsupervised: Any = None, backbone: str = "resnet50",
                 downsample_factor: Literal[1, 2, 3] = 2, pretrained: bool = True, torch_seed: int = 123, optimizer: str = "Adam",
                 optimizer_params: Any = None, lr_scheduler: str = "multisteplr", lr_scheduler_params: Any = None, **kwargs: Any) -> None:
        super().__init__(num_keypoints=num_keypoints, loss_factory=loss_factory, backbone=backbone,
                         downsample_factor=downsample_factor, pretrained=pretrained, torch_seed=torch_seed, optimizer=optimizer,
                         optimizer_params=optimizer_params, lr_scheduler=lr_scheduler, lr_scheduler_params=lr_scheduler_params,
                         **kwargs)
        self.loss_factory_unsup = loss_factory_unsupervised
        # modified by the AnnealWeight callback during training; deliberately not a buffer (reference :260-262)
        self.total_unsupervised_importance = torch.tensor(1.0)

    def get_loss_inputs_unlabeled(self, batch_dict: dict) -> HeatmapTrackerUnlabeledOutputsDict:
        pred_heatmaps = self.forward(batch_dict["frames"])
        pred_keypoints_augmented, pred_keypoints, confidence = self._decode(
            pred_heatmaps, batch_dict, batch_dict["transforms"], bool(batch_dict.get("is_multiview", False)))
        return {
            "heatmaps_pred": pred_heatmaps,                          # if augmented, augmented heatmaps
            "keypoints_pred": pred_keypoints,                        # if augmented, original keypoints (frame px)
            "keypoints_pred_augmented": pred_keypoints_augmented,    # match pred_heatmaps (model px)
            "confidences": confidence,
        }
