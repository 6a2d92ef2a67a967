"""Batch producers on the device: what hands the training step its two batches.

``VideoFramePipeline``   the device half of the reference's DALI ``video_pipe`` (data/video/dali.py:70-197) plus
                         ``LitDaliWrapper._dali_output_to_tensors`` (:267-330): decoded uint8 frames already resident in HBM ->
                         ``UnlabeledBatchDict`` / ``MultiviewUnlabeledBatchDict``.  Same constructor vocabulary (``resize_dims``,
                         ``normalization_mean/std``, ``imgaug`` in {"default", "dlc", "dlc-top-down"}, a seed), same random
                         ranges (rotation U(-10, 10) deg, scale U(0.8, 1.2)^2, brightness / contrast U(0.75, 1.25), shot-noise
                         factor U(0, 10)), same outputs (the (2, 3) matrix the fused decode later undoes; the ``[-1]`` sentinel
                         when nothing geometric happened).  Video DECODE is not here: frames arrive as a uint8 tensor.
``LabeledBatchProducer`` the per-batch work of ``HeatmapDataset.__getitem__`` (data/datasets.py:262-376, :496-550) moved to the
                         device: image resize + normalise, keypoint projection, optional flip with the left / right swap,
                         out-of-frame -> NaN, visibility synthesis, Gaussian targets (``lp_heatmap_gen``) -> ``HeatmapLabeledBatchDict``
                         with nothing but the uint8 images crossing PCIe.

``FrameWindowSource``   the sequencing half of ``fn.readers.video`` for already-decoded videos (windows of ``sequence_length`` frames every
                         ``step`` frames, seeded shuffling per rank, zero-padded tails) with the host -> device copy of the next window
                         overlapping the step on the current one (pinned staging, copy stream).

The image operators are restatements of DALI's / imgaug's published definitions (neither library is available to check against):
parity of pixel values is UNPINNED; keypoints, visibility and targets are pinned against the verbatim reference dataset.
"""

from __future__ import annotations

import math
import os
from typing import Sequence

import numpy as np
import torch

from .. import ops
from .datatypes import HeatmapLabeledBatchDict, MultiviewUnlabeledBatchDict, UnlabeledBatchDict

_IMAGENET_MEAN = [0.485, 0.456, 0.406]  # reference data/__init__.py:46-47
_IMAGENET_STD = [0.229, 0.224, 0.225]

_IMGAUG_VIDEO = ("default", "none", "dlc", "dlc-lr", "dlc-top-down", "dlc-mv")


def rotation_scale_matrix(angle_deg: float, scale_xy: Sequence[float], center_xy: Sequence[float]) -> np.ndarray:
    """``fn.transforms.scale(fn.transforms.rotation(angle, center), scale, center)`` (data/video/dali.py:158-161): the (2, 3)
    matrix, source -> destination, of a rotation about ``center`` followed by an axis scale about the same point."""
    th = math.radians(angle_deg)
    cx, cy = float(center_xy[0]), float(center_xy[1])
    t0 = np.array([[1.0, 0.0, -cx], [0.0, 1.0, -cy], [0.0, 0.0, 1.0]])
    t1 = np.array([[1.0, 0.0, cx], [0.0, 1.0, cy], [0.0, 0.0, 1.0]])
    rot = np.array([[math.cos(th), math.sin(th), 0.0], [-math.sin(th), math.cos(th), 0.0], [0.0, 0.0, 1.0]])
    sc = np.diag([float(scale_xy[0]), float(scale_xy[1]), 1.0])
    return (t1 @ sc @ t0 @ t1 @ rot @ t0)[:2]


class VideoFramePipeline:
    """Decoded frames -> the unlabeled batch of the semi-supervised step."""

    def __init__(self, resize_dims: Sequence[int] | None, normalization_mean: Sequence[float] = _IMAGENET_MEAN,
                 normalization_std: Sequence[float] = _IMAGENET_STD, imgaug: str = "default", seed: int = 123456,
                 border: str = "clamp") -> None:
        if imgaug not in _IMGAUG_VIDEO:
            raise NotImplementedError(f"cfg.training.imgaug string {imgaug} must be in {list(_IMGAUG_VIDEO)}")
        self.resize_dims = None if resize_dims is None else (int(resize_dims[0]), int(resize_dims[1]))
        self.mean, self.std = list(normalization_mean), list(normalization_std)
        self.imgaug = imgaug
        self.augment = imgaug in ("dlc", "dlc-top-down")  # the reference's video pipe augments for exactly these (:153)
        if self.augment and self.resize_dims is None:
            raise AssertionError("resize_dims is required when imgaug augments the video frames")
        self.border = border
        # each rank draws its own augmentations: seed + LOCAL_RANK, as the reference seeds its DALI pipes (:558, :565-568)
        self.seed = int(seed) + int(os.environ.get("LOCAL_RANK", "0"))
        self._rng = np.random.default_rng(self.seed)
        self._calls = 0

    def _draw(self, h: int, w: int) -> dict:
        r = self._rng
        angle = r.uniform(-10.0, 10.0)
        scale = r.uniform(0.8, 1.2, size=2)
        center = (h / 2.0, w / 2.0)  # (sic) the reference passes (resize_dims[0] / 2, resize_dims[1] / 2) as the (x, y) centre
        return {"matrix": rotation_scale_matrix(angle, scale, center), "contrast": r.uniform(0.75, 1.25),
                "brightness": r.uniform(0.75, 1.25), "shot_factor": r.uniform(0.0, 10.0)}

    def _one_view(self, frames_u8: torch.Tensor, params: dict | None):
        if frames_u8.dim() != 4 or frames_u8.shape[-1] != 3:
            raise ValueError(f"frames must be (S, H, W, 3) uint8, got {tuple(frames_u8.shape)}")
        s, hs, ws, _ = frames_u8.shape
        h, w = self.resize_dims if self.resize_dims is not None else (hs, ws)
        dev = frames_u8.device
        if self.augment:
            p = params if params is not None else self._draw(h, w)
            raw = ops.frames_resize(frames_u8, h, w, self.border)
            self._calls += 1
            frames = ops.frames_augment(raw, self.mean, self.std, matrix=p["matrix"], brightness=p["brightness"], contrast=p["contrast"],
                                        shot_factor=p["shot_factor"], seed=(self.seed << 20) + self._calls)
            transform = torch.tensor(np.asarray(p["matrix"], dtype=np.float32)).to(dev)
        else:
            frames = ops.frames_resize(frames_u8, h, w, self.border, mean=self.mean, std=self.std)
            transform = torch.tensor([-1.0]).to(dev)  # "no geometric transform to undo" (:170-172)
        return frames, transform, (hs, ws)

    def __call__(self, frames_u8: torch.Tensor | Sequence[torch.Tensor], params: dict | Sequence[dict] | None = None
                 ) -> UnlabeledBatchDict | MultiviewUnlabeledBatchDict:
        """One view: (S, Hs, Ws, 3) uint8 -> UnlabeledBatchDict.  A list of views (frame-synchronised by the caller) ->
        MultiviewUnlabeledBatchDict.  ``params`` overrides the random draw (tests, replay)."""
        if torch.is_tensor(frames_u8):
            frames, transform, (hs, ws) = self._one_view(frames_u8, params)
            bbox = torch.tensor([0.0, 0.0, float(hs), float(ws)], device=frames.device).repeat(frames.shape[0], 1)
            return UnlabeledBatchDict(frames=frames, transforms=transform, bbox=bbox, is_multiview=False)
        views = [self._one_view(f, None if params is None else params[i]) for i, f in enumerate(frames_u8)]
        frames = torch.stack([v[0] for v in views], dim=1)                      # (S, V, 3, H, W)
        transforms = torch.stack([v[1] if v[1].dim() == 2 else v[1].reshape(1, 1) for v in views], dim=0)  # (V, 2, 3) or (V, 1, 1)
        bbox = torch.cat([torch.tensor([0.0, 0.0, float(v[2][0]), float(v[2][1])], device=frames.device) for v in views]
                         ).repeat(frames.shape[0], 1)                           # (S, 4V)
        return MultiviewUnlabeledBatchDict(frames=frames, transforms=transforms, bbox=bbox, is_multiview=True)


class LabeledBatchProducer:
    """uint8 labeled images + stored labels -> the labeled batch of the step, built on the device."""

    def __init__(self, image_resize_height: int, image_resize_width: int, downsample_factor: int = 2, uniform_heatmaps: bool = False,
                 hflip_swap_indices: Sequence[int] | None = None, normalization_mean: Sequence[float] = _IMAGENET_MEAN,
                 normalization_std: Sequence[float] = _IMAGENET_STD, border: str = "renorm", interpolation: str = "cubic") -> None:
        """``interpolation``: "cubic" = imgaug ``iaa.Resize``'s default (OpenCV INTER_CUBIC, rounded to uint8 levels), what the reference's
        dataset applies to every labeled image (data/datasets.py:137-143); "linear" = the antialiased filter of the video pipeline."""
        if image_resize_height % 128 != 0 or image_resize_width % 128 != 0:
            raise ValueError("image dimensions (after transformation) must be repeatably divisible by 2; "
                             f"current dimensions: height={image_resize_height}, width={image_resize_width}")
        self.height, self.width = int(image_resize_height), int(image_resize_width)
        self.downsample_factor = int(downsample_factor)
        self.output_sigma = 1.25  # reference data/datasets.py:460
        self.uniform_heatmaps = bool(uniform_heatmaps)
        self.swap = None if hflip_swap_indices is None else torch.as_tensor(list(hflip_swap_indices), dtype=torch.int32)
        self.mean, self.std = list(normalization_mean), list(normalization_std)
        self.border = border
        self.interpolation = interpolation

    @property
    def output_shape(self) -> tuple[int, int]:
        return self.height // 2 ** self.downsample_factor, self.width // 2 ** self.downsample_factor

    def __call__(self, images_u8: torch.Tensor, keypoints: torch.Tensor, idxs: torch.Tensor | None = None,
                 visibility: torch.Tensor | None = None, bbox: torch.Tensor | None = None, affine: torch.Tensor | None = None,
                 hflip: torch.Tensor | None = None) -> HeatmapLabeledBatchDict:
        """images_u8 (B, Hs, Ws, 3) uint8 on the device; keypoints (B, 2K) or (B, K, 2) in source px (NaN = unlabeled);
        optional per-sample augmentation ``affine`` (B, 2, 3) on source px (applied to image and labels alike) and ``hflip`` (B)."""
        dev = images_u8.device
        b, hs, ws, _ = images_u8.shape
        kp = keypoints.reshape(b, -1, 2).to(dev)
        k = kp.shape[1]
        if self.swap is not None and sorted(self.swap.tolist()) != list(range(k)):  # host-side list: checked before it indexes on the device
            raise ValueError(f"hflip_swap_indices must be a permutation of range({k}), got {self.swap.tolist()}")
        src_hw = torch.tensor([[float(hs), float(ws)]], device=dev).repeat(b, 1)
        kp_model, vis = ops.labeled_keypoints(kp, src_hw, self.height, self.width, affine=affine, hflip=hflip, swap=self.swap,
                                              visibility=visibility, uniform_heatmaps=self.uniform_heatmaps)
        heatmaps = ops.generate_heatmaps(kp_model, self.height, self.width, self.output_shape, self.output_sigma, vis)
        if affine is None and hflip is None:
            images = ops.frames_resize(images_u8, self.height, self.width, self.border, mean=self.mean, std=self.std,
                                       interpolation=self.interpolation)
        else:
            raw = ops.frames_resize(images_u8, self.height, self.width, self.border, interpolation=self.interpolation)
            sx, sy = self.width / ws, self.height / hs
            to_model = np.array([[sx, 0.0, 0.0], [0.0, sy, 0.0], [0.0, 0.0, 1.0]])
            flips = None if hflip is None else hflip.cpu().numpy().astype(bool)
            aff = None if affine is None else affine.detach().cpu().numpy().astype(np.float64)
            planes = []
            for i in range(b):  # one parameter set per launch (a labeled batch is tens of images)
                m = np.eye(3)
                if aff is not None:
                    a3 = np.eye(3)
                    a3[:2] = aff[i]
                    m = to_model @ a3 @ np.linalg.inv(to_model)   # the source-px affine expressed in model px
                if flips is not None and flips[i]:
                    m = np.array([[-1.0, 0.0, float(self.width)], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]]) @ m
                planes.append(ops.frames_augment(raw[i:i + 1], self.mean, self.std, matrix=m[:2]))
            images = torch.cat(planes, 0)
        if bbox is None:  # x, y, h, w of the whole source frame (reference :352-356)
            bbox = torch.tensor([[0.0, 0.0, float(hs), float(ws)]], device=dev).repeat(b, 1)
        if idxs is None:
            idxs = torch.arange(b)
        return HeatmapLabeledBatchDict(images=images, keypoints=kp_model.reshape(b, 2 * k), heatmaps=heatmaps, bbox=bbox.to(dev), idxs=idxs)


class HostStager:
    """Host tensors -> device through PINNED memory on a copy stream of its own (round 4): the transfer overlaps whatever the compute stream
    is running - with Trainer.fit's one batch of look-ahead that is the previous step - instead of sitting in front of the producers' kernels
    on the compute stream (a pageable ``.to(device)`` is a synchronous staged copy; 31 MB of labeled images per step at BASELINE's batch).
    Two reusable pinned buffers (``pin_memory()`` per call costs more than the copy itself);
    a tensor that is already pinned is copied as it is.  The returned tensor is ordered on the CURRENT stream (it waits for the copy's event)."""

    def __init__(self, device):
        self.device = torch.device(device)
        self._stream = None
        self._bufs: list[torch.Tensor | None] = [None, None]
        self._events: list = [None, None]
        self._turn = 0

    def __call__(self, host: torch.Tensor) -> torch.Tensor:
        if self.device.type != "cuda" or host.device.type != "cpu":
            return host.to(self.device)
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=self.device)
        host = host.contiguous()
        src, j = host, None
        if not host.is_pinned():
            j, self._turn = self._turn, self._turn ^ 1
            if self._events[j] is not None:
                self._events[j].synchronize()      # the copy out of this buffer two calls ago (long finished)
            nbytes = host.numel() * host.element_size()
            if self._bufs[j] is None or self._bufs[j].numel() < nbytes:
                self._bufs[j] = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
            src = self._bufs[j][:nbytes].view(host.dtype).view(host.shape)
            src.copy_(host)
        with torch.cuda.stream(self._stream):
            dev = src.to(self.device, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self._stream)
        if j is not None:
            self._events[j] = done
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(done)
        dev.record_stream(cur)
        return dev


class FrameWindowSource:
    """The sequencing half of DALI's ``fn.readers.video`` (data/video/dali.py:135-151; pipe arguments :573-606) for videos that are already
    decoded into uint8 arrays (numpy arrays / memmaps / tensors of shape (N, H, W, 3)): windows of ``sequence_length`` frames starting every
    ``step`` frames, never across two videos; ``random_shuffle`` draws a seeded permutation of the windows per epoch (training; each rank seeds
    with ``seed + LOCAL_RANK`` as the reference's pipes do, :558-568), otherwise windows come in order (prediction); with ``pad_sequences`` the
    incomplete window at the end of a video is kept and its missing frames are zero (``PredictionHandler`` trims those rows), without it the
    tail is dropped.  Windows are staged through pinned host memory and copied on a side stream one window ahead of the consumer, so the
    copy of window i+1 overlaps the step on window i.  Decoding the container format is not done here."""

    def __init__(self, videos, sequence_length: int, step: int | None = None, random_shuffle: bool = False, pad_sequences: bool = True,
                 seed: int = 123456, device: torch.device | str | None = None) -> None:
        if isinstance(videos, (np.ndarray, torch.Tensor)):
            videos = [videos]
        self.videos = list(videos)
        for v in self.videos:
            if v.ndim != 4 or v.shape[-1] != 3 or str(v.dtype).replace("torch.", "") != "uint8":
                raise ValueError(f"each video must be a uint8 array of shape (N, H, W, 3), got {v.dtype} {tuple(v.shape)}")
        if sequence_length <= 0:
            raise ValueError("sequence_length must be positive")
        self.sequence_length = int(sequence_length)
        self.step = int(step) if step is not None else self.sequence_length
        self.random_shuffle, self.pad_sequences = bool(random_shuffle), bool(pad_sequences)
        self.seed = int(seed) + int(os.environ.get("LOCAL_RANK", "0"))
        self.device = torch.device(device) if device is not None else torch.device(f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}")
        self.epoch = 0
        self.windows: list[tuple[int, int]] = []  # (video index, first frame)
        for vi, v in enumerate(self.videos):
            n, start = int(v.shape[0]), 0
            while start < n:
                if start + self.sequence_length <= n or self.pad_sequences:
                    self.windows.append((vi, start))
                start += self.step
        self._copy_stream = None

    def __len__(self) -> int:
        return len(self.windows)

    @property
    def frame_count(self) -> int:
        """frames of the first video (what ``PredictionHandler(video_file=..., frame_count=...)`` needs for one video)"""
        return int(self.videos[0].shape[0])

    def _host_window(self, vi: int, start: int) -> torch.Tensor:
        v = self.videos[vi]
        stop = min(start + self.sequence_length, int(v.shape[0]))
        chunk = v[start:stop]
        chunk = chunk if torch.is_tensor(chunk) else torch.from_numpy(np.ascontiguousarray(chunk))
        if stop - start == self.sequence_length:
            return chunk
        out = torch.zeros((self.sequence_length, *chunk.shape[1:]), dtype=torch.uint8)  # pad_sequences: redundant frames are zero
        out[: stop - start] = chunk
        return out

    def _to_device(self, host: torch.Tensor) -> tuple[torch.Tensor, object]:
        if self.device.type != "cuda":
            return host.to(self.device), None
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=self.device)
        pinned = host if host.is_pinned() else host.pin_memory()
        with torch.cuda.stream(self._copy_stream):
            dev = pinned.to(self.device, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self._copy_stream)
        return dev, (done, pinned)  # the pinned buffer stays alive until the copy has been waited for

    def __iter__(self):
        order = list(range(len(self.windows)))
        if self.random_shuffle:
            order = np.random.default_rng(self.seed + self.epoch).permutation(len(order)).tolist()
        self.epoch += 1
        nxt = self._to_device(self._host_window(*self.windows[order[0]])) if order else None
        for i in range(len(order)):
            cur = nxt
            nxt = self._to_device(self._host_window(*self.windows[order[i + 1]])) if i + 1 < len(order) else None
            dev, token = cur
            if token is not None:
                torch.cuda.current_stream(self.device).wait_event(token[0])
                dev.record_stream(torch.cuda.current_stream(self.device))
            yield dev
