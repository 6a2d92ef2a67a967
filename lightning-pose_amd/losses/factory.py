"""Loss registry + LossFactory, the surface of lightning_pose/losses/factory.py for the heatmap-tracker path."""

from __future__ import annotations

from typing import Any, Literal

import numpy as np
import torch

from .. import ops
from .losses import HeatmapJSLoss, HeatmapKLLoss, HeatmapMSELoss, Loss, PCALoss, TemporalHeatmapLoss, TemporalLoss, UnimodalLoss

_HEATMAP_LOSSES = ("heatmap_mse", "heatmap_kl", "heatmap_js")


def get_loss_classes() -> dict[str, type[Loss]]:
    """Loss name -> class (reference :73-91).  Names outside the heatmap-tracker hot path (regression, the 3-D supervised losses)
    are not registered here; ``unimodal_mse`` is added."""
    return {
        HeatmapMSELoss.loss_name: HeatmapMSELoss,
        HeatmapKLLoss.loss_name: HeatmapKLLoss,
        HeatmapJSLoss.loss_name: HeatmapJSLoss,
        PCALoss.LOSS_NAME_MULTIVIEW: PCALoss,
        PCALoss.LOSS_NAME_SINGLEVIEW: PCALoss,
        TemporalLoss.loss_name: TemporalLoss,
        TemporalHeatmapLoss.LOSS_NAME_MSE: TemporalHeatmapLoss,
        TemporalHeatmapLoss.LOSS_NAME_KL: TemporalHeatmapLoss,
        UnimodalLoss.loss_name: UnimodalLoss,
    }


class LossFactory:
    """Holds one instance per configured loss and sums ``anneal * weight * loss`` (reference :197-285)."""

    def __init__(self, losses_params_dict: dict[str, dict], data_module: Any) -> None:
        self.losses_params_dict = losses_params_dict
        self.data_module = data_module
        self.loss_instance_dict: dict[str, Loss] = {}
        classes = get_loss_classes()
        for name, params in losses_params_dict.items():
            if name not in classes:
                raise NotImplementedError(f'loss "{name}" is not part of the MI355X heatmap-tracker path; '
                                          f"available: {sorted(classes)}")
            self.loss_instance_dict[name] = classes[name](data_module=data_module, **params)

    def __call__(self, stage: Literal["train", "val", "test"] | None = None, anneal_weight: float | torch.Tensor | None = 1.0,
                 **kwargs: Any) -> tuple[torch.Tensor, list[dict]]:
        log_list_all: list[dict] = []
        names, values, weights, anneals, logs = [], [], [], [], []
        for loss_name, loss_instance in self.loss_instance_dict.items():
            curr_loss, log_list = loss_instance(stage=stage, **kwargs)
            names.append(loss_name)
            values.append(curr_loss)
            weights.append(float(loss_instance.weight))
            anneals.append(1.0 if (anneal_weight is None or loss_name in _HEATMAP_LOSSES) else float(anneal_weight))
            logs.append(log_list)
        if not values:
            return torch.tensor(0.0), log_list_all
        # weighted[i] = weight_i * loss_i (logged), total = sum anneal_i * weighted[i] in registry order: ONE launch (ops.loss_combine)
        weighted, tot_loss = ops.loss_combine(values, weights, anneals)
        for i, (loss_name, log_list) in enumerate(zip(names, logs)):
            log_list_all += log_list + [{"name": f"{stage}_{loss_name}_loss_weighted", "value": weighted[i]}]
        return tot_loss, log_list_all


def _get(cfg: Any, key: str, default: Any = None) -> Any:
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default) if not hasattr(cfg, "get") else cfg.get(key, default)


def get_loss_factories(cfg: Any, data_module: Any) -> dict[str, LossFactory]:
    """{'supervised': LossFactory, 'unsupervised': LossFactory} from a hydra-style config (reference :94-194)."""
    model, data, losses = cfg["model"], cfg["data"], cfg["losses"]
    params: dict[str, dict[str, dict]] = {"supervised": {}, "unsupervised": {}}
    if str(model["model_type"]).find("heatmap") == -1:
        raise NotImplementedError("only heatmap trackers are implemented on the MI355X path")
    params["supervised"]["heatmap_" + str(_get(model, "heatmap_loss_type", "mse"))] = {"log_weight": 0.0}
    losses_to_use = _get(model, "losses_to_use", None)
    for loss_name in (losses_to_use or []):
        if loss_name == "":
            continue
        p = dict(losses[loss_name])
        p["loss_name"] = loss_name
        if loss_name == "pca_multiview":
            mcm = _get(data, "mirrored_column_matches")
            views = _get(data, "view_names", None)
            if views and len(views) > 1 and isinstance(mcm[0], int):
                nk = int(_get(data, "num_keypoints"))
                mcm = [(v * nk + np.array(mcm, dtype=int)).tolist() for v in range(len(views))]
            p["mirrored_column_matches"] = mcm
        elif loss_name == "pca_singleview":
            views = _get(data, "view_names", None)
            if views and len(views) > 1:
                raise NotImplementedError("The Pose PCA loss is currently not implemented for multiview data.")
            p["columns_for_singleview_pca"] = _get(data, "columns_for_singleview_pca", None)
        elif loss_name == "unimodal_mse":
            dims = _get(data, "image_resize_dims", None)
            if dims is not None:
                p.setdefault("original_image_height", int(dims["height"]))
                p.setdefault("original_image_width", int(dims["width"]))
        params["unsupervised"][loss_name] = p
    return {
        "supervised": LossFactory(losses_params_dict=params["supervised"], data_module=data_module),
        "unsupervised": LossFactory(losses_params_dict=params["unsupervised"], data_module=data_module),
    }
