"""Train / validation / test splits of a labeled dataset and the batch iterators over them (reference:
lightning_pose/data/datamodules.py:43-262 ``BaseDataModule``).

Same constructor arguments and attributes (``dataset``, ``train_dataset`` / ``val_dataset`` / ``test_dataset`` with ``.indices``, the
``*_batch_size`` and ``*_probability`` fields, ``train_frames``, ``torch_seed``) and the same split: ``split_sizes_from_probabilities`` then
``torch.utils.data.random_split`` under ``torch.Generator().manual_seed(torch_seed)``, so the same seed selects the same examples as the
reference.  The loaders yield ``HeatmapLabeledBatchDict``s built on the device by the dataset (no worker processes: nothing but the decoded
uint8 images is prepared on the host); the training loader shuffles with the reference's generator (``torch.randperm`` under
``manual_seed(torch_seed)``, continued across epochs), validation / test / full loaders run in order and never flip.
"""

from __future__ import annotations

from typing import Iterator

import torch
from torch.utils.data import Subset, random_split

from .utils import compute_num_train_frames, split_sizes_from_probabilities


class BaseDataModule:
    def __init__(self, dataset, train_batch_size: int = 16, val_batch_size: int = 16, test_batch_size: int = 1, num_workers: int | None = None,
                 train_probability: float = 0.8, val_probability: float | None = None, test_probability: float | None = None,
                 train_frames: float | int | None = None, torch_seed: int = 42) -> None:
        self.dataset = dataset
        self.train_batch_size, self.val_batch_size, self.test_batch_size = train_batch_size, val_batch_size, test_batch_size
        self.num_workers = 0  # batches are built on the device; ``num_workers`` is accepted for signature compatibility
        self.train_probability, self.val_probability, self.test_probability = train_probability, val_probability, test_probability
        self.train_frames = train_frames
        self.torch_seed = torch_seed
        self.train_dataset: Subset | None = None
        self.val_dataset: Subset | None = None
        self.test_dataset: Subset | None = None
        self._setup()
        self._train_generator = torch.Generator().manual_seed(self.torch_seed)

    def _setup(self) -> None:
        sizes = split_sizes_from_probabilities(len(self.dataset), train_probability=self.train_probability,
                                               val_probability=self.val_probability, test_probability=self.test_probability)
        self.train_dataset, self.val_dataset, self.test_dataset = random_split(
            self.dataset, sizes, generator=torch.Generator().manual_seed(self.torch_seed))
        if self.train_frames is not None:  # further subsample the training split (reference :186-193)
            n = compute_num_train_frames(len(self.train_dataset), self.train_frames)
            if n < len(self.train_dataset):
                self.train_dataset.indices = self.train_dataset.indices[:n]

    def _ordered(self, indices, batch_size: int) -> Iterator[dict]:
        for lo in range(0, len(indices), batch_size):
            yield self.dataset.batch(list(indices[lo:lo + batch_size]), hflip=torch.zeros(len(indices[lo:lo + batch_size]), dtype=torch.bool)
                                     if getattr(self.dataset, "imgaug_hflip", False) else None)

    def _rank_world(self) -> tuple[int, int]:
        import torch.distributed as dist

        return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)

    def train_dataloader(self) -> Iterator[dict]:
        """One epoch over the training split in a fresh random order (flips, if enabled, are drawn by the dataset).  Under data
        parallelism every rank draws the SAME permutation (same seed) and keeps its strided share, padded by wrapping around so all
        ranks see equally many examples - torch's DistributedSampler, which Lightning installs for the reference (its
        data/factory.py:252-255 already divides train_batch_size by the number of GPUs: pass that per-GPU size, see
        distributed.labeled_batch_per_gpu)."""
        idx = self.train_dataset.indices
        order = torch.randperm(len(idx), generator=self._train_generator).tolist()
        rank, world = self._rank_world()
        if world > 1:
            total = -(-len(order) // world) * world
            order = (order + order[:total - len(order)])[rank:total:world]
        for lo in range(0, len(order), self.train_batch_size):
            yield self.dataset.batch([idx[i] for i in order[lo:lo + self.train_batch_size]])

    def val_dataloader(self) -> Iterator[dict]:
        return self._ordered(self.val_dataset.indices, self.val_batch_size)

    def test_dataloader(self) -> Iterator[dict]:
        return self._ordered(self.test_dataset.indices, self.test_batch_size)

    def full_labeled_dataloader(self) -> Iterator[dict]:
        """Every labeled example in file order (what ``predict_dataset`` iterates, reference :248-261)."""
        return self._ordered(list(range(len(self.dataset))), self.val_batch_size)


class UnlabeledDataModule(BaseDataModule):
    """Labeled splits + an unlabeled video stream for the semi-supervised trackers (reference data/datamodules.py:252-356).

    ``video_source`` yields uint8 frame windows (``FrameWindowSource``), ``video_pipeline`` turns a window into the ``UnlabeledBatchDict``
    (``VideoFramePipeline``); together they stand where the reference has ``PrepareDALI`` / ``LitDaliWrapper``.  ``train_dataloader`` pairs
    the two streams like ``CombinedLoader(mode="max_size_cycle")``: an epoch lasts as long as the LONGER stream, the shorter one restarts."""

    def __init__(self, dataset, video_source, video_pipeline, **kwargs) -> None:
        super().__init__(dataset, **kwargs)
        self.video_source, self.video_pipeline = video_source, video_pipeline

    def unlabeled_dataloader(self) -> Iterator[dict]:
        for window in self.video_source:
            yield self.video_pipeline(window)

    def train_dataloader(self) -> Iterator[dict]:
        makers = {"labeled": super().train_dataloader, "unlabeled": self.unlabeled_dataloader}
        its = {k: iter(m()) for k, m in makers.items()}
        exhausted = {k: False for k in makers}
        while True:
            batch = {}
            for k in makers:
                try:
                    batch[k] = next(its[k])
                except StopIteration:
                    exhausted[k] = True
                    if all(exhausted.values()):
                        return
                    its[k] = iter(makers[k]())  # the shorter stream cycles
                    batch[k] = next(its[k])
            if all(exhausted.values()):
                return
            yield batch
