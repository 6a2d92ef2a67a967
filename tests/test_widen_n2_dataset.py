"""Labeled dataset: DLC-style label files + image files -> device-built labeled batches
(lightning_pose_amd/data/datasets.py; reference data/datasets.py:78-550, utils/io.py:190-279)."""

import os

import numpy as np
import pytest
import torch

from oracle import restated as O

REF_DATA = "/root/reference/data/mirror-mouse-example"
needs_reference_data = pytest.mark.skipif(not os.path.isdir(REF_DATA), reason="bundled example data only exist in the build container")


def _write_project(tmp_path, with_visible: bool, first_row_nan: bool = False):
    """3 images of 40 x 56 (one grey-scale), 3 keypoints with left / right partners, DLC three-row header"""
    from PIL import Image

    g = np.random.default_rng(0)
    (tmp_path / "labeled-data").mkdir()
    names = ["labeled-data/a.png", "labeled-data/b.png", "labeled-data/c.png"]
    for i, n in enumerate(names):
        arr = g.integers(0, 256, (40, 56) if i == 1 else (40, 56, 3), dtype=np.uint8)
        Image.fromarray(arr).save(tmp_path / n)
    kps = ["paw_left", "paw_right", "nose"]
    cols = ["x", "y", "visible"] if with_visible else ["x", "y"]
    header = [",".join(["scorer"] + ["me"] * (len(kps) * len(cols))), ",".join(["bodyparts"] + [k for k in kps for _ in cols]),
              ",".join(["coords"] + cols * len(kps))]
    xy = np.array([[[10.0, 5.0], [30.0, 20.0], [55.5, 39.0]], [[np.nan, np.nan], [1.0, 2.0], [20.0, 30.0]], [[5.0, 35.0], [50.0, 1.0], [np.nan, np.nan]]])
    if first_row_nan:
        xy[0] = np.nan
    vis = np.array([[2, 2, 1], [0, 2, 2], [2, 1, 0]])
    rows = []
    for i, n in enumerate(names):
        vals = []
        for k in range(3):
            vals += ["" if np.isnan(v) else repr(float(v)) for v in xy[i, k]]
            if with_visible:
                vals.append(str(vis[i, k]))
        rows.append(",".join([n] + vals))
    (tmp_path / "CollectedData.csv").write_text("\n".join(header + rows) + "\n")
    return names, kps, xy, vis


def test_parse_label_csv_formats(tmp_path):
    from lightning_pose_amd.data.datasets import build_hflip_swap_indices, parse_label_csv

    names, kps, xy, vis = _write_project(tmp_path, with_visible=True)
    data = parse_label_csv(str(tmp_path / "CollectedData.csv"))
    assert data.keypoint_names == kps and data.image_names == names
    np.testing.assert_array_equal(data.keypoints.numpy(), xy.astype(np.float32))
    np.testing.assert_array_equal(data.visibility.numpy(), vis)
    assert data.visibility.dtype == torch.int64
    assert build_hflip_swap_indices(kps).tolist() == [1, 0, 2]
    with pytest.raises(ValueError, match="_right partner"):
        build_hflip_swap_indices(["ear_left", "nose"])
    with pytest.raises(FileNotFoundError):
        parse_label_csv(str(tmp_path / "missing.csv"))
    bad = (tmp_path / "CollectedData.csv").read_text().replace(",2\n", ",7\n", 1)
    (tmp_path / "bad.csv").write_text(bad)
    with pytest.raises(ValueError, match="invalid values"):
        parse_label_csv(str(tmp_path / "bad.csv"))


def test_parse_label_csv_all_nan_first_row_is_data(tmp_path):
    from lightning_pose_amd.data.datasets import parse_label_csv

    sub = tmp_path / "p"
    sub.mkdir()
    names, _kps, xy, _ = _write_project(sub, with_visible=False, first_row_nan=True)
    data = parse_label_csv(str(sub / "CollectedData.csv"))
    assert data.visibility is None and data.image_names == names  # pandas alone would have swallowed row 0 as the index name
    assert torch.isnan(data.keypoints[0]).all()
    np.testing.assert_array_equal(data.keypoints[1:].numpy(), xy[1:].astype(np.float32))


@needs_reference_data
def test_parse_label_csv_matches_the_verbatim_parser_on_the_bundled_labels(golden):
    """the labels the verbatim HeatmapDataset parsed (stored in tests/golden/labeled_targets.npz) come out of this parser too"""
    from lightning_pose_amd.data.datasets import parse_label_csv

    data = parse_label_csv(os.path.join(REF_DATA, "CollectedData.csv"))
    g = golden("labeled_targets")
    idxs = [0, 3, 7, 11, 20, 33, 41, 57]
    np.testing.assert_array_equal(data.keypoints[idxs].numpy(), g["kp_src"])
    assert len(data.keypoint_names) == 17 and data.image_names[0] == "labeled-data/img01.png" and data.visibility is None


def test_dataset_batches_on_the_device(stack_backend, tmp_path):
    from lightning_pose_amd.data.datasets import HeatmapDataset

    dev = stack_backend
    names, kps, xy, vis = _write_project(tmp_path, with_visible=True)
    ds = HeatmapDataset(str(tmp_path), "CollectedData.csv", 128, 128, downsample_factor=2, imgaug_hflip=True, device=dev)
    assert len(ds) == 3 and ds.num_keypoints == 3 and ds.num_targets == 6 and ds.output_shape == (32, 32) and (ds.height, ds.width) == (128, 128)
    batch = ds.batch([2, 0, 1], hflip=torch.tensor([0, 1, 0]))
    assert tuple(batch["images"].shape) == (3, 3, 128, 128) and tuple(batch["heatmaps"].shape) == (3, 3, 32, 32)
    assert batch["idxs"].tolist() == [2, 0, 1] and batch["bbox"].cpu().tolist() == [[0.0, 0.0, 40.0, 56.0]] * 3
    order = [2, 0, 1]
    want_kp, want_vis = O.labeled_keypoints(torch.tensor(xy[order], dtype=torch.float32), torch.tensor([[40.0, 56.0]] * 3), 128, 128,
                                            hflip=torch.tensor([0, 1, 0]), swap=torch.tensor([1, 0, 2]), visibility=torch.tensor(vis[order]))
    got = batch["keypoints"].cpu().reshape(3, 3, 2)
    np.testing.assert_allclose(np.nan_to_num(got.numpy()), np.nan_to_num(want_kp.numpy()), atol=5e-5)
    want_hm = O.generate_heatmaps(want_kp, 128, 128, (32, 32), 1.25, want_vis)
    torch.testing.assert_close(batch["heatmaps"].cpu(), want_hm, atol=2e-6, rtol=0)
    # images: RGB decode (the grey-scale file replicated), imgaug's cubic resize at uint8 levels, normalise; the flipped sample mirrored
    raw = ds.load_images(order)
    assert raw.dtype == torch.uint8 and tuple(raw.shape) == (3, 40, 56, 3) and torch.equal(raw[2, ..., 0], raw[2, ..., 1])
    # (the flip happens on the resized image: the cubic taps of a mirrored pixel are the mirrored taps, up to one uint8 level at an exact .5)
    plain = O.frames_finish(O.frames_resize_cubic(raw, 128, 128))
    one_level = 1.01 / 255 / 0.224
    for got_img, want_img in ((batch["images"][0].cpu(), plain[0]), (batch["images"][1].cpu(), plain[1].flip(-1))):
        diff = (got_img - want_img).abs()
        assert float(diff.max()) <= one_level and float((diff > 3e-4).float().mean()) < 5e-3, (float(diff.max()), float((diff > 3e-4).float().mean()))
    # epoch iteration: every example once, reproducible order
    seen = [b["idxs"].tolist() for b in ds.batches(2, shuffle=True, seed=3)]
    assert sorted(sum(seen, [])) == [0, 1, 2] and seen == [b["idxs"].tolist() for b in ds.batches(2, shuffle=True, seed=3)]
    assert [b["idxs"].tolist() for b in ds.batches(2, shuffle=False, drop_last=True)] == [[0, 1]]
    with pytest.raises(NotImplementedError):
        HeatmapDataset(str(tmp_path), "CollectedData.csv", 128, 128, do_context=True, device=dev)


@needs_reference_data
def test_bundled_example_project_feeds_the_tracker(stack_backend):
    """the reference's own example project (grey-scale 406 x 396 PNGs, 17 keypoints with NaN labels) straight into a training step"""
    from lightning_pose_amd.data.datasets import HeatmapDataset
    from lightning_pose_amd.losses import LossFactory
    from lightning_pose_amd.models import HeatmapTracker

    dev = stack_backend
    ds = HeatmapDataset(REF_DATA, "CollectedData.csv", 128, 128, uniform_heatmaps=True, device=dev)
    assert len(ds) == 90 and ds.num_keypoints == 17
    batch = ds.batch([0, 5])
    assert batch["bbox"].cpu().tolist() == [[0.0, 0.0, 406.0, 396.0]] * 2
    model = HeatmapTracker(num_keypoints=17, loss_factory=LossFactory({"heatmap_mse": {"log_weight": 0.0}}, None), backbone="resnet50",
                           pretrained=False, torch_seed=0, device=dev)
    model.train()
    opt = model.configure_optimizers()["optimizer"]
    opt.zero_grad()
    loss = model.training_step(batch, 0)["loss"]
    loss.backward()
    assert torch.isfinite(loss).item() and float(model.logged["train_supervised_rmse"]) > 0
