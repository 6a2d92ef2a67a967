"""Loss classes of the heatmap-tracker step, same registry surface as lightning_pose/losses/losses.py, evaluated by
lp_hip kernels (lightning_pose_amd.ops).

Pipeline contract kept from the reference (losses/losses.py:10-17): each loss is a callable taking keyword tensors
(names discoverable from the ``__call__`` signature, which ``models.factory._validate_loss_model_compatibility``
introspects) plus ``stage`` and returning ``(scalar, [ {name, value[, prog_bar]}, {name: "<loss>_weight", value} ])``;
``weight = 1 / (2 exp(log_weight))`` (:89-100).  The reductions differ in mechanism, not in value: masked means stay
on the device (no boolean-index host sync), and value + gradient come out of one launch.
"""

from __future__ import annotations

import os
from typing import Any, Literal

import torch

from .. import ops
from ..utils.pca import KeypointPCA

__all__ = ["Loss", "HeatmapLoss", "HeatmapMSELoss", "TemporalLoss", "TemporalHeatmapLoss", "PCALoss", "UnimodalLoss", "RegressionRMSELoss"]

_DEFAULT_TORCH_DEVICE = "cpu"
if torch.cuda.is_available():
    _DEFAULT_TORCH_DEVICE = f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}"


class Loss:
    """Parent class: holds epsilon / log_weight, provides ``weight`` and ``log_loss`` (reference :60-198)."""

    loss_name: str

    def __init__(self, data_module: Any = None, epsilon: float | list[float] = 0.0, log_weight: float = 0.0, **kwargs: Any) -> None:
        self.data_module = data_module
        self.epsilon = torch.tensor(epsilon, dtype=torch.float)
        self.log_weight = torch.tensor(log_weight, dtype=torch.float)

    @property
    def weight(self) -> torch.Tensor:
        return 1.0 / (2.0 * torch.exp(self.log_weight))

    def rectify_epsilon(self, loss: torch.Tensor) -> torch.Tensor:
        """``relu(loss - epsilon)``: errors below epsilon cost nothing (reference :117-133; the fused kernels apply the same rule)."""
        return torch.relu(loss - self.epsilon.to(loss.device))

    def reduce_loss(self, loss: torch.Tensor, method: str = "mean") -> torch.Tensor:
        """reference :135-160"""
        if method == "mean":
            return torch.mean(loss)
        if method == "sum":
            return torch.sum(loss)
        raise NotImplementedError(f"reduction method {method!r}")

    def remove_nans(self, **kwargs: Any):
        raise NotImplementedError

    def compute_loss(self, **kwargs: Any):
        raise NotImplementedError

    def log_loss(self, loss: torch.Tensor, stage: Literal["train", "val", "test"] | None) -> list[dict]:
        return [
            {"name": f"{stage}_{self.loss_name}_loss", "value": loss, "prog_bar": True},
            {"name": f"{self.loss_name}_weight", "value": self.weight},
        ]

    def __call__(self, *args: Any, **kwargs: Any):
        raise NotImplementedError


class HeatmapLoss(Loss):
    """Losses on (heatmaps_targ, heatmaps_pred): all-zero target maps are ignored (reference :201-290)."""

    def __init__(self, data_module: Any = None, log_weight: float = 0.0, **kwargs: Any) -> None:
        super().__init__(data_module=data_module, log_weight=log_weight)

    def compute(self, targets: torch.Tensor, predictions: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    def __call__(self, heatmaps_targ: torch.Tensor, heatmaps_pred: torch.Tensor,
                 stage: Literal["train", "val", "test"] | None = None, **kwargs: Any):
        scalar_loss = self.compute(heatmaps_targ, heatmaps_pred)
        return scalar_loss, self.log_loss(loss=scalar_loss, stage=stage)


class HeatmapMSELoss(HeatmapLoss):
    """mean over labelled maps and pixels of (t - p)^2 * h * w (reference :293-335)."""

    loss_name = "heatmap_mse"

    def compute(self, targets: torch.Tensor, predictions: torch.Tensor) -> torch.Tensor:
        return ops.heatmap_mse(targets, predictions)


class HeatmapKLLoss(HeatmapLoss):
    """mean over labelled maps of KL(target || prediction) summed over the map, both + 1e-10 (reference :338-379)."""

    loss_name = "heatmap_kl"

    def compute(self, targets: torch.Tensor, predictions: torch.Tensor) -> torch.Tensor:
        return ops.heatmap_kl(targets, predictions)


class HeatmapJSLoss(HeatmapLoss):
    """mean over labelled maps of the Jensen-Shannon divergence of target and prediction (reference :382-423)."""

    loss_name = "heatmap_js"

    def compute(self, targets: torch.Tensor, predictions: torch.Tensor) -> torch.Tensor:
        return ops.heatmap_js(targets, predictions)


class TemporalLoss(Loss):
    """mean over (S-1) x K of relu(||kp[t+1] - kp[t]|| - eps_k), zeroed next to low-confidence frames (reference :576-703)."""

    loss_name = "temporal"

    def __init__(self, data_module: Any = None, epsilon: float | list[float] = 0.0, prob_threshold: float = 0.0,
                 log_weight: float = 0.0, **kwargs: Any) -> None:
        super().__init__(data_module=data_module, epsilon=epsilon, log_weight=log_weight)
        self.prob_threshold = torch.tensor(prob_threshold, dtype=torch.float)

    def __call__(self, keypoints_pred: torch.Tensor, confidences: torch.Tensor | None = None,
                 stage: Literal["train", "val", "test"] | None = None, **kwargs: Any):
        scalar_loss = ops.temporal_loss(keypoints_pred, confidences, self.epsilon, float(self.prob_threshold))
        return scalar_loss, self.log_loss(loss=scalar_loss, stage=stage)


class TemporalHeatmapLoss(Loss):
    """mean over (S-1) x K of relu(d(heatmap[t], heatmap[t+1]) - eps_k), d = pixel-mean squared difference
    ("temporal_heatmap_mse") or KL(heatmap[t+1] || heatmap[t]) ("temporal_heatmap_kl"), zeroed next to low-confidence frames
    (reference :706-869)."""

    LOSS_NAME_MSE = "temporal_heatmap_mse"
    LOSS_NAME_KL = "temporal_heatmap_kl"

    def __init__(self, loss_name: Literal["temporal_heatmap_mse", "temporal_heatmap_kl"], data_module: Any = None,
                 epsilon: float | list[float] = 0.0, prob_threshold: float = 0.0, log_weight: float = 0.0, **kwargs: Any) -> None:
        super().__init__(data_module=data_module, epsilon=epsilon, log_weight=log_weight)
        if loss_name not in (self.LOSS_NAME_MSE, self.LOSS_NAME_KL):
            raise ValueError(f"Invalid loss_name: {loss_name}")
        self.loss_name = loss_name
        self.prob_threshold = torch.tensor(prob_threshold, dtype=torch.float)

    @property
    def _kind(self) -> int:
        from .._lib import HM_KL, HM_MSE

        return HM_MSE if self.loss_name == self.LOSS_NAME_MSE else HM_KL

    def compute_loss(self, predictions: torch.Tensor) -> torch.Tensor:
        """(S, K, h, w) -> (S-1, K) distances between consecutive heat-maps (reference :793-829); diagnostic, not differentiable."""
        return ops.temporal_heatmap_distances(predictions, self._kind)

    def remove_nans(self, confidences: torch.Tensor, loss: torch.Tensor) -> torch.Tensor:
        """Zero the entries next to a frame whose confidence is below ``prob_threshold``, in place (reference :766-791)."""
        ignore = confidences < self.prob_threshold.to(confidences.device)
        loss[torch.logical_or(ignore[:-1], ignore[1:]).to(loss.device)] = 0.0
        return loss

    def rectify_epsilon(self, loss: torch.Tensor) -> torch.Tensor:
        """per-keypoint epsilon broadcast over the (S-1) rows (reference :754-764)"""
        return torch.relu(loss - self.epsilon.to(loss.device).reshape(1, -1))

    def __call__(self, heatmaps_pred: torch.Tensor, confidences: torch.Tensor, stage: Literal["train", "val", "test"] | None = None,
                 **kwargs: Any):
        kind = self._kind
        scalar_loss = ops.temporal_heatmap_loss(heatmaps_pred, confidences, self.epsilon, float(self.prob_threshold), kind)
        return scalar_loss, self.log_loss(loss=scalar_loss, stage=stage)


class PCALoss(Loss):
    """Penalise keypoints outside a low-dimensional subspace fitted on the labelled data (reference :426-573)."""

    LOSS_NAME_MULTIVIEW = "pca_multiview"
    LOSS_NAME_SINGLEVIEW = "pca_singleview"

    def __init__(self, loss_name: Literal["pca_singleview", "pca_multiview"], components_to_keep: int | float = 0.95,
                 empirical_epsilon_percentile: float = 99.0, epsilon: float | None = None, empirical_epsilon_multiplier: float = 1.0,
                 mirrored_column_matches=None, columns_for_singleview_pca=None, data_module: Any = None, log_weight: float = 0.0,
                 device: str | torch.device = _DEFAULT_TORCH_DEVICE, centering_method: str | None = None, **kwargs: Any) -> None:
        super().__init__(data_module=data_module, log_weight=log_weight)
        self.device = device
        if loss_name not in (self.LOSS_NAME_MULTIVIEW, self.LOSS_NAME_SINGLEVIEW):
            raise ValueError(f"Invalid loss_name: {loss_name}")
        self.loss_name = loss_name
        if loss_name == "pca_multiview" and mirrored_column_matches is None:
            raise ValueError("must provide mirrored_column_matches in data config")
        assert data_module is not None or kwargs.get("data_arr") is not None, "PCALoss requires a data_module to fit PCA"
        self.pca = KeypointPCA(loss_type=loss_name, data_module=data_module, components_to_keep=components_to_keep,
                               empirical_epsilon_percentile=empirical_epsilon_percentile,
                               mirrored_column_matches=mirrored_column_matches,
                               columns_for_singleview_pca=columns_for_singleview_pca, device=device,
                               centering_method=centering_method, data_arr=kwargs.get("data_arr"))
        self.pca()
        if epsilon is not None:
            self.epsilon = torch.tensor(epsilon, dtype=torch.float, device=self.device)
        else:
            self.epsilon = self.pca.parameters["epsilon"] * empirical_epsilon_multiplier
        self._index = None

    def __call__(self, keypoints_pred: torch.Tensor, stage: Literal["train", "val", "test"] | None = None, **kwargs: Any):
        assert keypoints_pred.device == torch.device(self.device), (keypoints_pred.device, torch.device(self.device))
        if self._index is None:
            self._index = torch.from_numpy(self.pca.index_table(keypoints_pred.shape[1] // 2)).to(keypoints_pred.device)
        if getattr(self, "_eps_host", None) is None or self._eps_src is not self.epsilon:
            # epsilon lives on the device (the reference's PCA parameters do): read it back ONCE - float() of a device tensor is a
            # host synchronisation, which stalls the launch queue every step and is illegal while a HIP graph is being captured
            self._eps_host, self._eps_src = float(self.epsilon), self.epsilon
        scalar_loss = ops.pca_loss(keypoints_pred, self._index, self.pca.parameters["mean"],
                                   self.pca.parameters["kept_eigenvectors"], self._eps_host)
        return scalar_loss, self.log_loss(loss=scalar_loss, stage=stage)


class UnimodalLoss(Loss):
    """"unimodal_mse": predicted heat-maps should look like ONE Gaussian at their own soft-argmax.

    NOT in the reference snapshot - parity unpinned; defined by ``oracle/restated.py unimodal_mse_loss``:
    ideal = generate_heatmaps(keypoints_pred_augmented) (detached); keep (s, k) with confidence >= prob_threshold and an
    in-bounds keypoint; mean over kept maps and pixels of (ideal - pred)^2 * h * w; 0 if nothing is kept.
    """

    loss_name = "unimodal_mse"

    def __init__(self, data_module: Any = None, original_image_height: int | None = None, original_image_width: int | None = None,
                 prob_threshold: float = 0.0, sigma: float = 1.25, log_weight: float = 0.0, **kwargs: Any) -> None:
        super().__init__(data_module=data_module, log_weight=log_weight)
        self.original_image_height = original_image_height
        self.original_image_width = original_image_width
        self.prob_threshold = torch.tensor(prob_threshold, dtype=torch.float)
        self.sigma = sigma

    def __call__(self, keypoints_pred_augmented: torch.Tensor, heatmaps_pred: torch.Tensor, confidences: torch.Tensor,
                 stage: Literal["train", "val", "test"] | None = None, **kwargs: Any):
        s, k, h, w = heatmaps_pred.shape
        # heat-maps are 1 / 2^downsample_factor of the network input; the config can pin the size explicitly
        ds = kwargs.get("downsample_factor", 2)
        img_h = self.original_image_height or h * 2 ** ds
        img_w = self.original_image_width or w * 2 ** ds
        scalar_loss = ops.unimodal_mse(keypoints_pred_augmented.detach().reshape(s, k, 2), heatmaps_pred, confidences, img_h, img_w,
                                       self.sigma, float(self.prob_threshold))
        return scalar_loss, self.log_loss(loss=scalar_loss, stage=stage)


class RegressionRMSELoss(Loss):
    """Always-on pixel-error diagnostic (reference :949-996): mean Euclidean error / sqrt(2) over labelled keypoints."""

    loss_name = "rmse"

    def __call__(self, keypoints_targ: torch.Tensor, keypoints_pred: torch.Tensor,
                 stage: Literal["train", "val", "test"] | None = None, **kwargs: Any):
        scalar_loss = ops.rmse(keypoints_targ, keypoints_pred)
        return scalar_loss, self.log_loss(loss=scalar_loss, stage=stage)
