"""Model registry (reference: lightning_pose/models/factory.py:67-113,116-192,195-319)."""

from __future__ import annotations

import inspect
from typing import Any, get_type_hints

from .base import check_if_semi_supervised
from .heatmap_tracker import HeatmapTracker, SemiSupervisedHeatmapTracker


def get_model_class(model_type: str, semi_supervised: bool) -> type:
    if model_type != "heatmap":
        raise NotImplementedError(f"{model_type} is an invalid model_type for a {'semi' if semi_supervised else 'fully'}-supervised "
                                  "model on the MI355X path (only 'heatmap' is implemented)")
    return SemiSupervisedHeatmapTracker if semi_supervised else HeatmapTracker


def _required_keys(loss: Any) -> set[str]:
    """Keyword tensors a loss needs = named parameters of its __call__ without defaults (reference :116-136)."""
    sig = inspect.signature(loss.__call__)
    return {n for n, p in sig.parameters.items()
            if p.kind in (p.POSITIONAL_OR_KEYWORD, p.KEYWORD_ONLY) and p.default is p.empty and n not in ("self", "stage")}


def _produced_keys(fn: Any) -> set[str]:
    """Keys a get_loss_inputs_* method produces = fields of its TypedDict return annotation (reference :139-155)."""
    ret = get_type_hints(fn).get("return")
    return set(getattr(ret, "__annotations__", {}).keys())


def _validate_loss_model_compatibility(model_class: type, loss_factories: dict[str, Any]) -> None:
    pairs = [("supervised", "get_loss_inputs_labeled"), ("unsupervised", "get_loss_inputs_unlabeled")]
    for which, method in pairs:
        factory = loss_factories.get(which)
        if factory is None or not hasattr(model_class, method) or not factory.loss_instance_dict:
            continue
        produced = _produced_keys(getattr(model_class, method))
        for name, loss in factory.loss_instance_dict.items():
            missing = _required_keys(loss) - produced
            if missing:
                raise ValueError(f"loss '{name}' requires {sorted(missing)} but {model_class.__name__}.{method} only provides "
                                 f"{sorted(produced)}")


def _get(cfg: Any, key: str, default: Any = None) -> Any:
    try:
        return cfg[key]
    except (KeyError, TypeError, AttributeError):
        return getattr(cfg, key, default)


def get_model(cfg: Any, data_module: Any, loss_factories: dict[str, Any]):
    """Build the tracker named by ``cfg.model`` (hydra-style mapping with data / training / model sections)."""
    model, data, training = cfg["model"], cfg["data"], cfg["training"]
    semi = check_if_semi_supervised(_get(model, "losses_to_use", None))
    cls = get_model_class(str(model["model_type"]), semi)
    _validate_loss_model_compatibility(cls, loss_factories)
    dims = data["image_resize_dims"]
    if "vit" in str(model["backbone"]) and dims["height"] != dims["width"]:
        raise RuntimeError("ViT backbones require square images")
    lr = _get(_get(training, "optimizer_params", {}) or {}, "learning_rate", 1e-3)
    kwargs = dict(
        num_keypoints=int(data["num_keypoints"]),
        loss_factory=loss_factories["supervised"],
        backbone=str(model["backbone"]),
        pretrained=bool(_get(model, "backbone_pretrained", True)),
        torch_seed=int(_get(training, "rng_seed_model_pt", 0)),
        optimizer=str(_get(training, "optimizer", "Adam")),
        optimizer_params={"learning_rate": lr},
        lr_scheduler=str(_get(training, "lr_scheduler", "multisteplr")),
        lr_scheduler_params=_get(_get(training, "lr_scheduler_params", {}) or {}, "multisteplr", None),
        image_size=int(dims["height"]),
        downsample_factor=int(_get(data, "downsample_factor", 2)),
        backbone_checkpoint=_get(model, "backbone_checkpoint", None),
    )
    if semi:
        kwargs["loss_factory_unsupervised"] = loss_factories["unsupervised"]
    net = cls(**kwargs)
    ckpt = _get(model, "checkpoint", None)
    if ckpt:  # warm start from a trained model (reference :299-317)
        from ..checkpoint import load_weights
        load_weights(net, str(ckpt))
    return net
