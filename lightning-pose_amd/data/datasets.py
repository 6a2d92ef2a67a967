"""Labeled dataset for the heatmap trackers: DLC-style label files + image files in, device-built labeled batches out.

Mirror of the part of ``lightning_pose/data/datasets.py`` (``BaseTrackingDataset`` :78-376, ``HeatmapDataset`` :380-550) and
``lightning_pose/utils/io.py`` (``parse_label_csv`` :208-279, ``LabeledData`` :190-205) that the labeled half of the training step
consumes - same constructor arguments, attributes (``keypoints``, ``visibility``, ``image_names``, ``keypoint_names``, ``num_keypoints``,
``num_targets``, ``height`` / ``width``, ``output_shape``, ``output_sigma``) and label-file semantics (three-row DLC header, optional
``visible`` column with values 0 / 1 / 2, an all-NaN first row is a data row, visibility synthesised from NaN labels otherwise).

What differs is where the work happens: the reference transforms one sample at a time in DataLoader workers (PIL -> imgaug -> ToTensor ->
Normalize -> ``generate_heatmaps`` on the CPU) and ships fp32 images plus fp32 targets to the GPU; here only the decoded uint8 images cross
PCIe and ``LabeledBatchProducer`` builds the whole ``HeatmapLabeledBatchDict`` on the device (resize + normalise, keypoint projection,
optional flip with the left / right swap, out-of-frame -> NaN, Gaussian targets).  The imgaug augmentation zoo and context (5-frame)
loading are outside this path.
"""

from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Iterator, Sequence

import numpy as np
import pandas as pd
import torch

from .datatypes import HeatmapLabeledBatchDict
from .producers import HostStager, LabeledBatchProducer


@dataclass
class LabeledData:
    """Result of parsing a label file (reference utils/io.py:190-205)."""

    keypoint_names: list[str]
    image_names: list[str]
    keypoints: torch.Tensor            # (N, K, 2) float32, NaN where unlabeled
    visibility: torch.Tensor | None    # (N, K) int64 in {0, 1, 2}, or None without a ``visible`` column


def parse_label_csv(csv_file: str, header_rows: list[int] | None = None) -> LabeledData:
    """Read a DLC-style label CSV once (reference utils/io.py:208-279)."""
    if header_rows is None:
        header_rows = [0, 1, 2]
    if not os.path.exists(csv_file):
        raise FileNotFoundError(f"could not find csv file at {csv_file}")
    df = pd.read_csv(csv_file, header=header_rows, index_col=0)
    if df.index.name is not None:  # pandas took an all-NaN first data row for the index name (reference :529-554)
        first = pd.DataFrame({c: np.nan for c in df.columns}, index=pd.Index([df.index.name]), columns=df.columns, dtype="float64")
        df = pd.concat([first, df])
    if header_rows in ([1, 2], [0, 1]):
        keypoint_names = [c[0] for c in df.columns if c[1] == "x"]
    else:
        keypoint_names = [c[1] for c in df.columns if c[2] == "x"]
    raw = torch.tensor(df.to_numpy(), dtype=torch.float32)
    if header_rows == [0, 1, 2] and any(c[2] == "visible" for c in df.columns):
        raw = raw.reshape(raw.shape[0], -1, 3)
        vis = raw[:, :, 2]
        invalid = set(vis[~torch.isnan(vis)].unique().tolist()) - {0.0, 1.0, 2.0}
        if invalid:
            raise ValueError(f"visibility column contains invalid values {invalid}; expected values in {{0, 1, 2}}")
        return LabeledData(keypoint_names, list(df.index), raw[:, :, :2].contiguous(), vis.long())
    return LabeledData(keypoint_names, list(df.index), raw.reshape(raw.shape[0], -1, 2), None)


def build_hflip_swap_indices(keypoint_names: Sequence[str]) -> np.ndarray:
    """Entry i = the keypoint that fills position i after a horizontal flip: ``*_left`` <-> ``*_right`` partners swap, everything else
    stays (reference data/datasets.py:203-245).  Unmatched partners raise ValueError."""
    idx = list(range(len(keypoint_names)))
    left = {n[:-5]: i for i, n in enumerate(keypoint_names) if n.endswith("_left")}
    right = {n[:-6]: i for i, n in enumerate(keypoint_names) if n.endswith("_right")}
    lonely_l = sorted(f"{b}_left" for b in set(left) - set(right))
    lonely_r = sorted(f"{b}_right" for b in set(right) - set(left))
    if lonely_l:
        raise ValueError(f"imgaug_hflip requires matching _left/_right pairs, but found _left keypoints with no _right partner: {lonely_l}")
    if lonely_r:
        raise ValueError(f"imgaug_hflip requires matching _left/_right pairs, but found _right keypoints with no _left partner: {lonely_r}")
    for base, i in left.items():
        idx[i], idx[right[base]] = right[base], i
    return np.array(idx, dtype=np.intp)


class HeatmapDataset:
    """Labels in memory, images on disk, batches built on the device."""

    def __init__(self, root_directory: str, csv_path: str, image_resize_height: int, image_resize_width: int,
                 header_rows: list[int] | None = [0, 1, 2], downsample_factor: int = 2, do_context: bool = False,
                 uniform_heatmaps: bool = False, imgaug_hflip: bool = False, device: torch.device | str | None = None) -> None:
        if do_context:
            raise NotImplementedError("context (5-frame) datasets belong to the MHCRNN models, outside the MI355X heatmap-tracker path")
        self.root_directory = str(root_directory)
        csv_file = csv_path if os.path.isfile(csv_path) else os.path.join(self.root_directory, csv_path)
        data = parse_label_csv(csv_file, header_rows=header_rows)
        self.keypoint_names, self.image_names, self.keypoints = data.keypoint_names, data.image_names, data.keypoints
        self.num_keypoints = int(self.keypoints.shape[1])
        self.num_targets = 2 * self.num_keypoints
        self.do_context = False
        self.downsample_factor = int(downsample_factor)
        self.output_sigma = 1.25
        self.uniform_heatmaps = bool(uniform_heatmaps)
        if data.visibility is None:  # synthesised from the NaN labels (reference :465-472)
            nan = torch.isnan(self.keypoints[:, :, 0])
            self.visibility = torch.where(nan, torch.full_like(nan, 1 if uniform_heatmaps else 0, dtype=torch.long),
                                          torch.full_like(nan, 2, dtype=torch.long))
        else:
            self.visibility = data.visibility
        self.imgaug_hflip = bool(imgaug_hflip)
        swap = build_hflip_swap_indices(self.keypoint_names) if imgaug_hflip else None
        self.producer = LabeledBatchProducer(image_resize_height, image_resize_width, downsample_factor=downsample_factor,
                                             uniform_heatmaps=uniform_heatmaps, hflip_swap_indices=None if swap is None else swap.tolist())
        self.device = torch.device(device) if device is not None else torch.device(f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}")
        self._stage = HostStager(self.device)
        self._rng = np.random.default_rng(0)

    @property
    def height(self) -> int:
        return self.producer.height

    @property
    def width(self) -> int:
        return self.producer.width

    @property
    def output_shape(self) -> tuple[int, int]:
        return self.producer.output_shape

    def __len__(self) -> int:
        return len(self.image_names)

    def load_images(self, indices: Sequence[int]) -> torch.Tensor:
        """(B, H, W, 3) uint8 on the host: each file decoded as RGB (single-channel images are replicated, reference :277)."""
        from PIL import Image

        frames = []
        for i in indices:
            with Image.open(os.path.join(self.root_directory, self.image_names[int(i)])) as im:
                frames.append(np.asarray(im.convert("RGB")))
        if len({f.shape for f in frames}) != 1:
            raise ValueError(f"images of one batch must share a size, got {sorted({f.shape for f in frames})}")
        return torch.from_numpy(np.stack(frames))

    def batch(self, indices: Sequence[int], hflip: torch.Tensor | None = None) -> HeatmapLabeledBatchDict:
        """The labeled batch of the step for these examples; ``hflip`` (B,) overrides the random flip decisions of ``imgaug_hflip``."""
        idx = torch.as_tensor(list(indices), dtype=torch.long)
        images = self._stage(self.load_images(idx.tolist()))   # pinned staging + copy stream: overlaps the step that is running
        if hflip is None and self.imgaug_hflip:
            hflip = torch.from_numpy(self._rng.random(len(idx)) < 0.5)  # each sample flips with probability 0.5 (reference :275)
        return self.producer(images, self.keypoints[idx].to(self.device), idxs=idx, visibility=self.visibility[idx].to(self.device),
                             hflip=hflip)

    def batches(self, batch_size: int, shuffle: bool = True, seed: int = 0, drop_last: bool = False) -> Iterator[HeatmapLabeledBatchDict]:
        order = np.random.default_rng(seed).permutation(len(self)) if shuffle else np.arange(len(self))
        for lo in range(0, len(order), batch_size):
            chunk = order[lo:lo + batch_size]
            if drop_last and len(chunk) < batch_size:
                return
            yield self.batch(chunk.tolist())
