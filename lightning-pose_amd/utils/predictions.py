"""Inference path: ``predict_step`` batches -> DLC-style prediction tables.

Mirror of ``lightning_pose/utils/predictions.py`` for the part that sits directly after the hot path:
``PredictionHandler`` (:41-330), ``make_dlc_pandas_index`` (:551-570) and the prediction loop that
``predict_dataset`` / ``predict_video`` (:332-548) delegate to ``pl.Trainer.predict``.  Same names, argument meaning,
column layout and error behaviour; the arithmetic (trunk, head, fused decode incl. the bounding-box map) is the HIP
path behind ``HeatmapTracker.predict_step`` - nothing here computes keypoints on the host.

Not mirrored (outside the hot path): video readers (DALI / pynvvc / OpenCV), labeled-video rendering, metric files.
``frame_count`` therefore takes the number of frames from the caller (``frame_count=...``) or from a reader-provided
``count_frames`` callable instead of opening the video with OpenCV.
"""

from __future__ import annotations

import os
from typing import Any, Callable, Iterable

import numpy as np
import pandas as pd
import torch


def _get(cfg: Any, key: str, default: Any = None) -> Any:
    """cfg may be an omegaconf DictConfig, a plain dict, or any attribute container."""
    if hasattr(cfg, "get"):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


def make_dlc_pandas_index(cfg: Any, keypoint_names: list[str]) -> pd.MultiIndex:
    """Three-level (scorer, bodyparts, coords) column index; scorer = ``<model_type>_tracker`` (reference :551-570)."""
    model_type = _get(_get(cfg, "model"), "model_type")
    return pd.MultiIndex.from_product([[f"{model_type}_tracker"], list(keypoint_names), ["x", "y", "likelihood"]],
                                      names=["scorer", "bodyparts", "coords"])


class PredictionHandler:
    """Convert batches of model outputs into a prediction dataframe (reference :41-330)."""

    def __init__(self, cfg: Any, data_module: Any = None, video_file: str | None = None, *, frame_count: int | None = None,
                 count_frames: Callable[[str], int] | None = None) -> None:
        if data_module is None and video_file is None:
            raise ValueError("must pass either data_module or video_file")
        if _get(_get(cfg, "data"), "keypoint_names", None) is None:
            raise ValueError("must include `keypoint_names` field in cfg.data")
        self.cfg = cfg
        self.data_module = data_module
        self.video_file = video_file
        self._frame_count = frame_count
        self._count_frames = count_frames

    @property
    def frame_count(self) -> int:
        """Number of frames in the video or in the labeled dataset (reference :67-74)."""
        if self.video_file is not None:
            if self._frame_count is not None:
                return int(self._frame_count)
            if self._count_frames is not None:
                return int(self._count_frames(self.video_file))
            raise RuntimeError("video prediction needs frame_count=<n> or count_frames=<callable>: this package does not "
                               "open video files (the reference uses OpenCV, data/utils.py count_frames)")
        return len(self.data_module.dataset)

    @property
    def keypoint_names(self) -> list[str]:
        return list(_get(_get(self.cfg, "data"), "keypoint_names"))

    @property
    def do_context(self) -> bool:
        if self.data_module:
            return bool(self.data_module.dataset.do_context)
        return _get(_get(self.cfg, "model"), "model_type") == "heatmap_mhcrnn"

    def unpack_preds(self, preds: list[tuple[torch.Tensor, torch.Tensor]]) -> tuple[torch.Tensor, torch.Tensor]:
        """Stack per-batch (keypoints, confidences); for video loaders drop the padded rows of the last sequence and undo
        the two-frame shift of context models (reference :97-144)."""
        stacked_preds = torch.vstack([pred[0] for pred in preds])
        stacked_confs = torch.vstack([pred[1] for pred in preds])
        if self.video_file is not None:
            num_rows_to_discard = stacked_preds.shape[0] - self.frame_count
            if num_rows_to_discard > 0:
                stacked_preds = stacked_preds[:-num_rows_to_discard]
                stacked_confs = stacked_confs[:-num_rows_to_discard]
            if self.do_context:
                stacked_preds = self.fix_context_preds_confs(stacked_preds)
                zero_pad = _get(_get(self.cfg, "model"), "model_type") != "heatmap_mhcrnn"
                stacked_confs = self.fix_context_preds_confs(stacked_confs, zero_pad_confidence=zero_pad)
        return stacked_preds, stacked_confs

    def fix_context_preds_confs(self, stacked_preds: torch.Tensor, zero_pad_confidence: bool = False) -> torch.Tensor:
        """Row 0 of a context loader belongs to frame 2: shift by two, replicate the edges (reference :146-177)."""
        head = torch.tile(stacked_preds[0], (2, 1))
        combined = torch.vstack([head, stacked_preds[0:-2]])
        if combined.shape[0] == self.frame_count:
            combined[-2:, :] = combined[-3, :]
        else:
            n_pad = self.frame_count - combined.shape[0]
            combined = torch.vstack([combined, torch.tile(combined[0], (n_pad, 1))])
        if zero_pad_confidence:
            combined[:2, :] = 0.0
            combined[-2:, :] = 0.0
        return combined

    @staticmethod
    def make_pred_arr_undo_resize(keypoints_np: np.ndarray, confidence_np: np.ndarray) -> np.ndarray:
        """(n, 2K) + (n, K) -> (n, 3K) with columns (x, y, likelihood) per keypoint (reference :179-206)."""
        assert keypoints_np.shape[0] == confidence_np.shape[0]
        assert keypoints_np.shape[1] == confidence_np.shape[1] * 2
        num_joints = confidence_np.shape[-1]
        predictions = np.zeros((keypoints_np.shape[0], num_joints * 3))
        predictions[:, 0::3] = keypoints_np[:, 0::2]
        predictions[:, 1::3] = keypoints_np[:, 1::2]
        predictions[:, 2::3] = confidence_np
        return predictions

    def make_dlc_pandas_index(self, keypoint_names: list | None = None) -> pd.MultiIndex:
        return make_dlc_pandas_index(cfg=self.cfg, keypoint_names=keypoint_names or self.keypoint_names)

    def add_split_indices_to_df(self, df: pd.DataFrame) -> pd.DataFrame:
        """Column ("set", "", "") = train / validation / test / unused per labeled frame (reference :222-239)."""
        df["set"] = np.array(["unused"] * df.shape[0])
        splits = {"train": self.data_module.train_dataset.indices, "validation": self.data_module.val_dataset.indices,
                  "test": self.data_module.test_dataset.indices}
        for key, val in splits.items():
            df.loc[val, ("set", "", "")] = np.repeat(key, len(val))
        return df

    def __call__(self, preds: list[tuple[torch.Tensor, torch.Tensor]], is_multiview_video: bool = False):
        """Prediction table of one video / labeled dataset; a dict of tables keyed by view name for multiview models
        (reference :264-330)."""
        stacked_preds, stacked_confs = self.unpack_preds(preds=preds)
        view_names = _get(_get(self.cfg, "data"), "view_names", None)
        if view_names and len(view_names) > 1 and (self.video_file is None or is_multiview_video):
            num_keypoints = len(self.keypoint_names)
            view_to_df = {}
            for view_idx, view_name in enumerate(view_names):
                beg, end = view_idx * num_keypoints, (view_idx + 1) * num_keypoints
                pred_arr = self.make_pred_arr_undo_resize(stacked_preds[:, beg * 2:end * 2].cpu().numpy(),
                                                          stacked_confs[:, beg:end].cpu().numpy())
                df = pd.DataFrame(pred_arr, columns=self.make_dlc_pandas_index(self.keypoint_names))
                view_to_df[view_name] = df
                if self.video_file is None:
                    df = self.add_split_indices_to_df(df)
                    df.index = self.data_module.dataset.dataset[view_name].image_names
            return view_to_df
        pred_arr = self.make_pred_arr_undo_resize(stacked_preds.cpu().numpy(), stacked_confs.cpu().numpy())
        df = pd.DataFrame(pred_arr, columns=self.make_dlc_pandas_index())
        if self.video_file is None:
            df = self.add_split_indices_to_df(df)
            df.index = self.data_module.dataset.image_names
        return df


def predict_batches(model: Any, batches: Iterable[dict], return_heatmaps: bool = False) -> list[tuple[torch.Tensor, ...]]:
    """What ``pl.Trainer(...).predict(model, dataloaders=..., return_predictions=True)`` does for this module: eval mode
    (BatchNorm running statistics), no autograd tape, ``predict_step`` per batch.  Outputs stay on the device."""
    was_training = model.training
    model.eval()
    out = []
    try:
        with torch.no_grad():
            for batch_idx, batch in enumerate(batches):
                out.append(model.predict_step(batch, batch_idx, return_heatmaps=return_heatmaps))
    finally:
        model.train(was_training)
    return out


def predict_dataset(model: Any, data_module: Any, preds_file: str | list[str], cfg: Any = None):
    """Predict every labeled frame and save the table(s) (reference :332-393).  ``model`` is the tracker itself or an object
    with ``.model`` / ``.config.cfg`` like the reference's API wrapper."""
    module = getattr(model, "model", model)
    cfg_eff = cfg if cfg is not None else model.config.cfg
    preds = predict_batches(module, data_module.full_labeled_dataloader())
    handler = PredictionHandler(cfg=cfg_eff, data_module=data_module, video_file=None)
    df = handler(preds=[(p[0], p[1]) for p in preds])
    if isinstance(df, dict):
        if isinstance(preds_file, str):
            for view_name, d in df.items():
                d.to_csv(preds_file.replace(".csv", f"_{view_name}.csv"))
        else:
            assert list(df.keys()) == list(_get(_get(cfg_eff, "data"), "view_names"))
            if len(preds_file) != len(df):
                raise ValueError("preds_file must have one entry per view")
            for d, f in zip(df.values(), preds_file):
                d.to_csv(f)
    else:
        assert isinstance(preds_file, str), "preds_file must be a str for single-view predictions"
        df.to_csv(preds_file)
    return df


def predict_video(video_file: str | list[str], model: Any, predict_loader: Iterable[dict], output_pred_file: str | list[str] | None = None,
                  *, cfg: Any = None, frame_count: int | None = None, count_frames: Callable[[str], int] | None = None):
    """Predict one video (or one video per view) from a loader of ``{"frames", "bbox", ...}`` batches and save the table(s)
    (reference :416-548).  The loader is the caller's: building it from a file is the video producer's job (8f N1)."""
    is_multiview = not isinstance(video_file, str)
    module = getattr(model, "model", model)
    cfg_eff = cfg if cfg is not None else model.config.cfg
    if is_multiview:
        if output_pred_file is not None and not isinstance(output_pred_file, list):
            raise ValueError("for multiview prediction, 'output_pred_file' should be a list corresponding to view_names")
        view_names = list(_get(_get(cfg_eff, "data"), "view_names"))
        if len(view_names) != len(video_file):
            raise ValueError("expected video_file to correspond 1-1 with cfg.data.view_names")
        for f, view_name in zip(video_file, view_names):
            assert view_name in os.path.splitext(os.path.basename(f))[0], \
                "expected video_file to correspond 1-1 with cfg.data.view_name"
    handler = PredictionHandler(cfg=cfg_eff, video_file=video_file[0] if is_multiview else video_file, frame_count=frame_count,
                                count_frames=count_frames)
    preds = predict_batches(module, predict_loader)
    df = handler(preds=[(p[0], p[1]) for p in preds], is_multiview_video=is_multiview)
    if isinstance(df, dict):
        df = [df[v] for v in _get(_get(cfg_eff, "data"), "view_names")]
    if output_pred_file is not None:
        if is_multiview:
            if len(output_pred_file) != len(df):
                raise ValueError("output_pred_file must have one entry per view")
            for d, f in zip(df, output_pred_file):
                os.makedirs(os.path.dirname(f) or ".", exist_ok=True)
                d.to_csv(f)
        else:
            df.to_csv(output_pred_file)
    return df
