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


def format_multiview_data_for_pca(data_arr, mirrored_column_matches: list[list[int]]):
    """(N, K, 2) -> (N * J, 2 * V): one row per (frame, matched keypoint), columns [x_v0, y_v0, x_v1, y_v1, ...] (reference :759-792).
    numpy in -> numpy out, torch in -> torch out (same device)."""
    n_kp = len(mirrored_column_matches[0])
    cols = []
    for view in mirrored_column_matches:
        assert len(view) == n_kp, "every view in mirrored_column_matches must list the same number of keypoints"
        idx = list(map(int, view))
        cols.append(data_arr[:, idx].reshape(-1, 2))
    return torch.cat(cols, dim=1) if torch.is_tensor(data_arr) else np.concatenate(cols, axis=1)


def convert_dict_values_to_tensors(param_dict: dict, device: str | torch.device) -> dict[str, torch.Tensor]:
    """every value as a float32 tensor on ``device`` (reference :795-811)"""
    return {k: torch.tensor(v, dtype=torch.float, device=device) for k, v in param_dict.items()}


class EmpiricalEpsilon:
    """percentile of a per-term loss (NaNs ignored), used as the epsilon of the epsilon-insensitive PCA loss (reference :611-636)"""

    def __init__(self, percentile: float) -> None:
        self.percentile = percentile

    def __call__(self, loss) -> float:
        flat = loss.detach().flatten().cpu().numpy() if torch.is_tensor(loss) else np.asarray(loss).flatten()
        return float(np.nanpercentile(flat, self.percentile, axis=0))


class ComponentChooser:
    """number of principal components to keep: an integer count, or the smallest count explaining a fraction of the variance
    (reference :639-738; same validation, performed at construction)"""

    def __init__(self, fitted_pca_params, components_to_keep: int | float | None) -> None:
        self.fitted_pca_params = fitted_pca_params
        self.components_to_keep = components_to_keep
        n_max = int(fitted_pca_params.n_components_)
        if type(components_to_keep) is int:
            if components_to_keep > n_max:
                raise ValueError(f"components_to_keep was set to {components_to_keep}, exceeding the maximum value of {n_max} "
                                 "observation dims")
        elif type(components_to_keep) is float:
            if not 0.0 <= components_to_keep <= 1.0:
                raise ValueError(f"components_to_keep was set to {components_to_keep} while it has to be between 0.0 and 1.0")
        elif components_to_keep is not None:
            raise TypeError(f"components_to_keep must be int or float, got {type(components_to_keep)}")

    @property
    def cumsum_explained_variance(self) -> np.ndarray:
        return np.cumsum(self.fitted_pca_params.explained_variance_ratio_)

    def __call__(self) -> int:
        n_max = int(self.fitted_pca_params.n_components_)
        if self.components_to_keep is None:
            return n_max
        if type(self.components_to_keep) is int:
            return self.components_to_keep
        if self.components_to_keep == 1.0:
            return n_max
        return int(np.where(self.cumsum_explained_variance >= self.components_to_keep)[0][0]) + 1


class NaNPCA:
    """PCA by eigen-decomposition of a covariance estimated from the observed entries only (NaN = missing), with scikit-learn's
    attribute names and sign convention; equals ``sklearn.decomposition.PCA(svd_solver="full")`` on complete data
    (reference :331-609, the ``covariance_eigh`` path of sklearn with masked covariance).  Plain numpy, float64."""

    def __init__(self, n_components: int | None = None, whiten: bool = False, **_unused) -> None:
        self.n_components = n_components
        self.whiten = whiten

    def fit(self, X: np.ndarray) -> "NaNPCA":
        X = np.asarray(X, dtype=np.float64)
        n_samples, n_features = X.shape
        n_components = min(X.shape) if self.n_components is None else int(self.n_components)
        self.mean_ = np.nanmean(X, axis=0)
        cov = np.ma.cov(np.ma.masked_invalid(X), rowvar=False).data
        evals, evecs = np.linalg.eigh(cov)
        evals, vt = evals[::-1].copy(), evecs[:, ::-1].T.copy()
        evals[evals < 0.0] = 0.0
        # sklearn svd_flip(u_based_decision=False): the largest-magnitude entry of every component is positive
        pivot = np.argmax(np.abs(vt), axis=1)
        sign = np.sign(vt[np.arange(vt.shape[0]), pivot])
        sign[sign == 0] = 1.0
        vt *= sign[:, None]
        total_var = evals.sum()
        self.n_samples_, self.n_features_in_, self.n_components_ = n_samples, n_features, n_components
        self.components_ = vt[:n_components]
        self.explained_variance_ = evals[:n_components]
        self.explained_variance_ratio_ = self.explained_variance_ / total_var
        self.singular_values_ = np.sqrt(self.explained_variance_ * (n_samples - 1))
        self.noise_variance_ = float(evals[n_components:].mean()) if n_components < min(n_features, n_samples) else 0.0
        return self

    def transform(self, X: np.ndarray) -> np.ndarray:
        """projection onto the components; rows with missing entries get the least-squares latent of their observed entries
        (z = (W^T M W)^-1 W^T M x, M = diag(observed)), fully missing rows zeros"""
        X = np.asarray(X, dtype=np.float64)
        valid = ~np.isnan(X)
        Xc = np.where(valid, X - self.mean_, 0.0)
        W = self.components_.T
        out = np.zeros((X.shape[0], self.n_components_))
        full = valid.all(axis=1)
        out[full] = Xc[full] @ W
        for i in np.where(~full)[0]:
            if not valid[i].any():
                continue
            Wm = W * valid[i][:, None]
            try:
                out[i] = np.linalg.inv(W.T @ Wm) @ (Wm.T @ Xc[i])
            except np.linalg.LinAlgError:
                out[i] = 0.0
        if self.whiten:
            out /= np.sqrt(self.explained_variance_)
        return out


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

    # -- diagnostics (the training loss itself is lp_pca_fwd_bwd; these serve metrics / inspection, reference :266-309) ---------
    def reproject(self, data_arr: torch.Tensor) -> torch.Tensor:
        """(N, obs) formatted samples -> their projection onto the kept subspace, back in observation space"""
        evecs, mean = self.parameters["kept_eigenvectors"], self.parameters["mean"].unsqueeze(0)
        assert data_arr.shape[1] == evecs.shape[1] == mean.shape[1] and data_arr.shape[1] % 2 == 0
        return ((data_arr - mean) @ evecs.T) @ evecs + mean

    def compute_reprojection_error(self, data_arr: torch.Tensor) -> torch.Tensor:
        """(N, obs) -> (N, obs / 2): Euclidean distance of every 2-D keypoint to its reprojection"""
        diff = data_arr - self.reproject(data_arr)
        return torch.linalg.norm(diff.reshape(diff.shape[0], diff.shape[1] // 2, 2), dim=2)

    # -- fit ----------------------------------------------------------------------------------------------------
    def __call__(self) -> None:
        x = self._format_data(self._get_data())
        if x.shape[0] < x.shape[1]:
            raise ValueError(f"cannot fit PCA with {x.shape[0]} samples < {x.shape[1]} observation dimensions")
        pca = NaNPCA().fit(x)
        self.pca_object = pca
        mean, vt, ratio, ncomp = pca.mean_, pca.components_, pca.explained_variance_ratio_, pca.n_components_
        self.explained_variance_ratio_ = ratio
        if self.loss_type == "pca_multiview":
            keep = 3
            if self.components_to_keep != 3:
                warnings.warn(f"for pca_multiview loss, you specified {self.components_to_keep} components_to_keep, "
                              "but we will instead keep 3 components", stacklevel=2)
        else:
            keep = ComponentChooser(pca, self.components_to_keep)()
        self._n_components_kept = keep
        mean32, kept32 = mean.astype(np.float32), vt[:keep].astype(np.float32)
        # empirical epsilon: percentile of the training-data reprojection error, evaluated in fp32 like the reference
        xc = x.astype(np.float32) - mean32
        resid = xc - (xc @ kept32.T) @ kept32
        err = np.sqrt((resid.reshape(resid.shape[0], -1, 2) ** 2).sum(-1))
        eps = EmpiricalEpsilon(self.empirical_epsilon_percentile)(err)
        self.parameters = {
            "mean": torch.tensor(mean32, device=self.device),
            "kept_eigenvectors": torch.tensor(kept32, device=self.device),
            "discarded_eigenvectors": torch.tensor(vt[keep:].astype(np.float32), device=self.device),
            "epsilon": torch.tensor(eps, dtype=torch.float, device=self.device),
        }
