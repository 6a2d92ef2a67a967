"""Dataset splits and loaders (reference data/datamodules.py:43-262, data/utils.py:17-139) over the labeled dataset."""

import os

import pytest
import torch

from tests.test_widen_n2_dataset import _write_project

from tests.conftest import needs_reference  # noqa: E402

# (n, train, val, test) -> the verbatim function's answer (generated with the reference in the build container; re-checked below when present)
SPLIT_CASES = {
    (3, 0.8, None, None): [2, 1, 0], (7, 0.8, 0.1, None): [6, 1, 0], (10, 0.9, 0.05, 0.05): [9, 1, 0], (90, 0.8, None, None): [72, 9, 9],
    (93, 0.8, 0.1, None): [74, 9, 10], (101, 0.5, 0.25, 0.25): [50, 25, 26], (1000, 0.95, None, None): [950, 25, 25],
    (12345, 0.7, 0.2, None): [8641, 2469, 1235], (10, 0.8, None, None): [9, 1, 0],
}


def test_split_sizes_and_train_frames():
    from lightning_pose_amd.data.utils import compute_num_train_frames, split_sizes_from_probabilities

    for (n, tp, vp, te), want in SPLIT_CASES.items():
        assert split_sizes_from_probabilities(n, tp, vp, te) == want, (n, tp, vp, te)
    with pytest.raises(ValueError):
        split_sizes_from_probabilities(1, 0.8)
    assert [compute_num_train_frames(90, t) for t in (None, 1, 5, 0.5, 200, 2)] == [90, 90, 5, 45, 90, 2]
    with pytest.raises(ValueError):
        compute_num_train_frames(90, -1)


@needs_reference
def test_split_functions_match_the_verbatim_ones():
    from lightning_pose_amd.data.utils import compute_num_train_frames, split_sizes_from_probabilities
    from oracle import ref_loader as R

    U = R.load("data.utils")
    for n in (2, 3, 7, 10, 57, 90, 93, 101, 1000, 12345):
        for tp, vp, te in ((0.8, None, None), (0.8, 0.1, None), (0.9, 0.05, 0.05), (0.5, 0.25, 0.25), (0.95, None, None), (0.7, 0.2, None)):
            def run(f):
                try:
                    return f(n, tp, vp, te)
                except (AssertionError, ValueError) as e:
                    return type(e).__name__
            assert run(split_sizes_from_probabilities) == run(U.split_sizes_from_probabilities), (n, tp, vp, te)
        for tf in (None, 1, 5, 0.5, 200, 0.99, 2):
            assert compute_num_train_frames(n, tf) == U.compute_num_train_frames(n, tf)
    for key, want in SPLIT_CASES.items():
        assert U.split_sizes_from_probabilities(*key) == want


def test_datamodule_splits_and_loaders(stack_backend, tmp_path):
    from lightning_pose_amd.data.datamodules import BaseDataModule
    from lightning_pose_amd.data.datasets import HeatmapDataset
    from lightning_pose_amd.utils.predictions import PredictionHandler

    dev = stack_backend
    _write_project(tmp_path, with_visible=True)
    ds = HeatmapDataset(str(tmp_path), "CollectedData.csv", 128, 128, imgaug_hflip=True, device=dev)
    dm = BaseDataModule(ds, train_batch_size=2, val_batch_size=2, train_probability=0.8, torch_seed=42)
    # the reference's split: sizes from the probabilities, members from random_split under manual_seed(torch_seed)
    want = torch.utils.data.random_split(range(3), [2, 1, 0], generator=torch.Generator().manual_seed(42))
    assert [list(dm.train_dataset.indices), list(dm.val_dataset.indices), list(dm.test_dataset.indices)] == [list(w.indices) for w in want]
    train = [b["idxs"].tolist() for b in dm.train_dataloader()]
    assert sorted(sum(train, [])) == sorted(dm.train_dataset.indices)
    order = torch.randperm(2, generator=torch.Generator().manual_seed(42)).tolist()
    assert sum(train, []) == [dm.train_dataset.indices[i] for i in order]           # DataLoader(shuffle=True, generator=manual_seed(seed))
    val = list(dm.val_dataloader())
    assert [b["idxs"].tolist() for b in val] == [list(dm.val_dataset.indices)]
    full = list(dm.full_labeled_dataloader())
    assert sum([b["idxs"].tolist() for b in full], []) == [0, 1, 2]
    # validation / full batches never flip: keypoints are the plain projection
    kp0 = full[0]["keypoints"].cpu().reshape(2, 3, 2)[0, 0]
    torch.testing.assert_close(kp0, torch.tensor([10.0 * 128 / 56, 5.0 * 128 / 40]))
    # the split column of the prediction table comes from these subsets
    cfg = {"data": {"keypoint_names": ds.keypoint_names}, "model": {"model_type": "heatmap"}}
    preds = [(torch.zeros(3, 6), torch.ones(3, 3))]
    df = PredictionHandler(cfg, data_module=dm)(preds=preds)
    sets = df[("set", "", "")].tolist()
    assert [sets[i] for i in dm.train_dataset.indices] == ["train"] * 2 and sets[dm.val_dataset.indices[0]] == "validation"
    assert list(df.index) == ds.image_names
    dm2 = BaseDataModule(ds, train_probability=0.8, train_frames=1, torch_seed=42)
    assert len(dm2.train_dataset) == 2  # train_frames == 1 means "all" (a fraction of one)


def test_semi_supervised_epoch_from_files_and_frames(stack_backend, tmp_path):
    """label file + images + a decoded video -> UnlabeledDataModule -> Trainer.fit: one epoch of the semi-supervised tracker with no
    hand-made tensors in between; the streams pair up like CombinedLoader(mode="max_size_cycle")"""
    from lightning_pose_amd.data.datamodules import UnlabeledDataModule
    from lightning_pose_amd.data.datasets import HeatmapDataset
    from lightning_pose_amd.data.producers import FrameWindowSource, VideoFramePipeline
    from lightning_pose_amd.losses import LossFactory
    from lightning_pose_amd.models import SemiSupervisedHeatmapTracker
    from lightning_pose_amd.trainer import Trainer

    dev = stack_backend
    _write_project(tmp_path, with_visible=False)
    ds = HeatmapDataset(str(tmp_path), "CollectedData.csv", 128, 128, uniform_heatmaps=True, device=dev)
    video = torch.randint(0, 256, (9, 40, 56, 3), generator=torch.Generator().manual_seed(1), dtype=torch.uint8)
    dm = UnlabeledDataModule(ds, FrameWindowSource(video, 3, random_shuffle=True, pad_sequences=False, device=dev),
                             VideoFramePipeline([128, 128], imgaug="dlc", seed=2), train_batch_size=1, train_probability=0.8, torch_seed=0)
    # 2 labeled batches of 1, 3 unlabeled windows of 3 frames -> 3 steps, the labeled stream restarting once
    batches = list(dm.train_dataloader())
    assert len(batches) == 3 and all(set(b) == {"labeled", "unlabeled"} for b in batches)
    assert [tuple(b["unlabeled"]["frames"].shape) for b in batches] == [(3, 3, 128, 128)] * 3
    labeled_ids = [b["labeled"]["idxs"].tolist()[0] for b in batches]
    assert set(labeled_ids[:2]) == set(dm.train_dataset.indices) and labeled_ids[2] in dm.train_dataset.indices
    model = SemiSupervisedHeatmapTracker(num_keypoints=3, loss_factory=LossFactory({"heatmap_mse": {"log_weight": 0.0}}, None),
                                         loss_factory_unsupervised=LossFactory({"temporal": {"log_weight": 2.0, "epsilon": 0.5}}, None),
                                         backbone="resnet50", pretrained=False, torch_seed=0, device=dev)
    trainer = Trainer(max_epochs=1, data_parallel=False, limit_train_batches=1)  # (one step: the emulator is slow at 128 x 128)
    trainer.fit(model, lambda epoch: dm.train_dataloader(), val_batches=lambda epoch: dm.val_dataloader())
    assert len(trainer.logged_history) == 1 and model.global_step == 1
    # the epoch ended with a validation pass over the held-out labeled example (eval mode, folded-BatchNorm inference forward)
    assert len(trainer.validation_history) == 1 and model.training
    val = trainer.validation_history[0]
    assert {"val_supervised_loss", "val_supervised_rmse", "val_heatmap_mse_loss"} <= set(val) and val["val_supervised_loss"] > 0
    # ... and the trained module labels its own dataset: predict_dataset -> DLC-style CSV with the split column
    import pandas as pd

    from lightning_pose_amd.utils.predictions import predict_dataset

    cfg = {"data": {"keypoint_names": ds.keypoint_names}, "model": {"model_type": "heatmap"}}
    df = predict_dataset(model, dm, str(tmp_path / "predictions.csv"), cfg=cfg)
    assert df.shape == (3, 3 * 3 + 1) and list(df.index) == ds.image_names
    assert sorted(df[("set", "", "")].tolist()) == ["train", "train", "validation"]
    back = pd.read_csv(tmp_path / "predictions.csv", header=[0, 1, 2], index_col=0)
    assert back.shape == df.shape and torch.isfinite(torch.tensor(back.iloc[:, :9].to_numpy(dtype=float))).all()
    assert all(torch.isfinite(torch.tensor(h["total_loss"])) for h in trainer.logged_history)
