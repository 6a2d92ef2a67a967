"""PCA subspace for the pca_singleview / pca_multiview losses.

Reference: lightning_pose/utils/pca.py - ``KeypointPCA`` :30-328 (data formatting :97-190, fit driver :205-264,
reprojection :266-309), ``NaNPCA._fit_full`` :419-564 (NaN-masked covariance + eigh + sklearn sign convention),
``ComponentChooser`` :639-738, ``EmpiricalEpsilon`` :611-636.  The fit runs ONCE on the host in float64 numpy (as in
the reference, which calls scikit-learn on the CPU); only its result (mean, kept eigenvectors, epsilon) goes to the
device, where ``lp_pca_fwd_bwd`` evaluates the loss.
"""

from __future__ import annotations

import warnings

import numpy as np
import torch


def format_multiview_data_for_pca(data_arr: np.ndarray, mirrored_column_matches: list[list[int]]) -> np.ndarray:
    """(N, K, 2) -> (N * J, 2 * V): one row per (frame, matched keypoint), columns [x_v0, y_v0, x_v1, y_v1, ...]."""
    n_kp = len(mirrored_column_matches[0])
    cols = []
    for view in mirrored_column_matches:
        if len(view) != n_kp:
            raise ValueError("every view in mirrored_column_matches must list the same number of keypoints")
        cols.append(data_arr[:, np.asarray(view)].reshape(-1, 2))
    return np.concatenate(cols, axis=1)


class KeypointPCA:
    def __init__(self, loss_type: str, data_module=None, components_to_keep: int | float | None = 0.99,
                 empirical_epsilon_percentile: float = 99.0, mirrored_column_matches=None, columns_for_singleview_pca=None,
                 device: str | torch.device = "cpu", centering_method: str | None = None, data_arr=None):
        if loss_type not in ("pca_singleview", "pca_multiview"):
            raise NotImplementedError(loss_type)
        if centering_method is not None:
            raise NotImplementedError("centering_method is not part of the hot path covered here")
        self.loss_type = loss_type
        self.data_module = data_module
        self.components_to_keep = components_to_keep
        self.empirical_epsilon_percentile = empirical_epsilon_percentile
        self.mirrored_column_matches = [list(map(int, m)) for m in mirrored_column_matches] if mirrored_column_matches is not None else None
        self.columns_for_singleview_pca = list(map(int, columns_for_singleview_pca)) if columns_for_singleview_pca is not None else None
        self.device = torch.device(device)
        self._data_arr = data_arr
        self.parameters: dict[str, torch.Tensor] = {}

    # -- data -------------------------------------------------------------------------------------------------
    def _get_data(self) -> np.ndarray:
        if self._data_arr is not None:
            arr = self._data_arr
        elif self.data_module is not None and hasattr(self.data_module, "pca_keypoints"):
            arr = self.data_module.pca_keypoints()   # (N, 2K) un-augmented training keypoints, NaN where unlabeled
        else:
            raise AssertionError("PCALoss requires a data_module exposing pca_keypoints() (or an explicit data_arr) to fit PCA")
        arr = arr.detach().cpu().numpy() if torch.is_tensor(arr) else np.asarray(arr)
        return arr.astype(np.float64)

    def _format_data(self, data_arr: np.ndarray) -> np.ndarray:
        kp = data_arr.reshape(data_arr.shape[0], -1, 2)
        if self.loss_type == "pca_multiview":
            return format_multiview_data_for_pca(kp, self.mirrored_column_matches)
        if self.columns_for_singleview_pca is not None:
            kp = kp[:, np.asarray(self.columns_for_singleview_pca)]
        return kp.reshape(kp.shape[0], -1)

    def index_table(self, num_keypoints: int) -> np.ndarray:
        """(rows, points) keypoint ids per PCA sample, the layout lp_pca_fwd_bwd consumes."""
        if self.loss_type == "pca_multiview":
            return np.ascontiguousarray(np.asarray(self.mirrored_column_matches, dtype=np.int32).T)
        cols = self.columns_for_singleview_pca if self.columns_for_singleview_pca is not None else list(range(num_keypoints))
        return np.asarray(cols, dtype=np.int32).reshape(1, -1)

    # -- fit ----------------------------------------------------------------------------------------------------
    def __call__(self) -> None:
        x = self._format_data(self._get_data())
        if x.shape[0] < x.shape[1]:
            raise ValueError(f"cannot fit PCA with {x.shape[0]} samples < {x.shape[1]} observation dimensions")
        mean = np.nanmean(x, axis=0)
        cov = np.ma.cov(np.ma.masked_invalid(x), rowvar=False).data
        evals, evecs = np.linalg.eigh(cov)
        evals, vt = evals[::-1].copy(), evecs[:, ::-1].T.copy()
        evals[evals < 0.0] = 0.0
        # sklearn svd_flip(u_based_decision=False): largest-magnitude entry of every component is positive
        pivot = np.argmax(np.abs(vt), axis=1)
        sign = np.sign(vt[np.arange(vt.shape[0]), pivot])
        sign[sign == 0] = 1.0
        vt *= sign[:, None]
        ncomp = min(x.shape)
        vt, evals = vt[:ncomp], evals[:ncomp]
        ratio = evals / evals.sum()
        self.explained_variance_ratio_ = ratio
        if self.loss_type == "pca_multiview":
            keep = 3
            if self.components_to_keep != 3:
                warnings.warn(f"for pca_multiview loss, you specified {self.components_to_keep} components_to_keep, "
                              "but we will instead keep 3 components", stacklevel=2)
        elif type(self.components_to_keep) is int:
            if self.components_to_keep > ncomp:
                raise ValueError(f"components_to_keep was set to {self.components_to_keep}, exceeding the maximum value of {ncomp} "
                                 "observation dims")
            keep = self.components_to_keep
        elif type(self.components_to_keep) is float:
            if not 0.0 <= self.components_to_keep <= 1.0:
                raise ValueError(f"components_to_keep was set to {self.components_to_keep} while it has to be between 0.0 and 1.0")
            keep = ncomp if self.components_to_keep == 1.0 else int(np.where(np.cumsum(ratio) >= self.components_to_keep)[0][0]) + 1
        else:
            raise TypeError(f"components_to_keep must be int or float, got {type(self.components_to_keep)}")
        self._n_components_kept = keep
        mean32, kept32 = mean.astype(np.float32), vt[:keep].astype(np.float32)
        # empirical epsilon: percentile of the training-data reprojection error, evaluated in fp32 like the reference
        xc = x.astype(np.float32) - mean32
        resid = xc - (xc @ kept32.T) @ kept32
        err = np.sqrt((resid.reshape(resid.shape[0], -1, 2) ** 2).sum(-1))
        eps = float(np.nanpercentile(err.flatten(), self.empirical_epsilon_percentile, axis=0))
        self.parameters = {
            "mean": torch.tensor(mean32, device=self.device),
            "kept_eigenvectors": torch.tensor(kept32, device=self.device),
            "discarded_eigenvectors": torch.tensor(vt[keep:].astype(np.float32), device=self.device),
            "epsilon": torch.tensor(eps, dtype=torch.float, device=self.device),
        }
