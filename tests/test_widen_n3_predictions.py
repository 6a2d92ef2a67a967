"""Inference path: PredictionHandler / predict_batches / predict_video against the golden tables of the
reference's own PredictionHandler (tests/golden/predictions.npz, utils/predictions.py:41-330) and, end to end, against
the oracle's decode of the tracker's own heat-maps."""

import numpy as np
import pandas as pd
import pytest
import torch

from oracle import restated as O


class _Cfg(dict):
    __getattr__ = dict.get


def _cfg(names, model_type="heatmap", view_names=None):
    data = _Cfg(keypoint_names=names)
    if view_names:
        data["view_names"] = view_names
    return _Cfg(data=data, model=_Cfg(model_type=model_type))


def _batches(kp, conf, bsz):
    return [(kp[i:i + bsz], conf[i:i + bsz]) for i in range(0, kp.shape[0], bsz)]


NAMES = [f"kp{i}" for i in range(5)]


def test_single_view_video_table_matches_reference(golden):
    from lightning_pose_amd.utils.predictions import PredictionHandler

    g = golden("predictions")
    h = PredictionHandler(_cfg(NAMES), video_file="clip.mp4", frame_count=70)
    df = h(preds=_batches(g.t("v1_kp"), g.t("v1_conf"), 16))
    assert df.shape == (70, 15)  # the 10 padded rows of the last sequence are dropped
    np.testing.assert_array_equal(df.to_numpy(), g["v1_table"])
    assert ["|".join(c) for c in df.columns] == [str(c) for c in g["v1_columns"]]
    assert list(df.columns.names) == ["scorer", "bodyparts", "coords"]
    assert df.columns[0] == ("heatmap_tracker", "kp0", "x") and df.columns[5] == ("heatmap_tracker", "kp1", "likelihood")


def test_count_frames_callable_and_missing_count():
    from lightning_pose_amd.utils.predictions import PredictionHandler

    h = PredictionHandler(_cfg(NAMES), video_file="clip.mp4", count_frames=lambda f: 3)
    assert h.frame_count == 3
    with pytest.raises(RuntimeError, match="frame_count"):
        PredictionHandler(_cfg(NAMES), video_file="clip.mp4").frame_count


def test_multiview_video_tables_match_reference(golden):
    from lightning_pose_amd.utils.predictions import PredictionHandler

    g = golden("predictions")
    h = PredictionHandler(_cfg(NAMES, "heatmap_multiview_transformer", ["top", "bot"]), video_file="clip_top.mp4", frame_count=21)
    d = h(preds=_batches(g.t("v2_kp"), g.t("v2_conf"), 8), is_multiview_video=True)
    assert list(d) == ["top", "bot"]
    np.testing.assert_array_equal(d["top"].to_numpy(), g["v2_top"])
    np.testing.assert_array_equal(d["bot"].to_numpy(), g["v2_bot"])
    # one view at a time (is_multiview_video=False with a video file) yields ONE table over all K*V columns' first K: the
    # reference falls to the single-view branch, whose index has K names -> a shape error for K*V columns
    with pytest.raises(ValueError):
        h(preds=_batches(g.t("v2_kp"), g.t("v2_conf"), 8))


def test_labeled_dataset_table_with_split_column(golden):
    from lightning_pose_amd.utils.predictions import PredictionHandler

    g = golden("predictions")

    class Sub:
        def __init__(self, idx):
            self.indices = idx

    class DS:
        do_context = False
        image_names = [str(s) for s in g["v3_index"]]

        def __len__(self):
            return 12

    class DM:
        dataset = DS()
        train_dataset, val_dataset, test_dataset = Sub([0, 2, 4, 6, 8, 10]), Sub([1, 5]), Sub([3, 7])

    df = PredictionHandler(_cfg(NAMES), data_module=DM())(preds=_batches(g.t("v3_kp"), g.t("v3_conf"), 4))
    assert list(df.index) == DS.image_names
    assert df[("set", "", "")].tolist() == [str(s) for s in g["v3_set"]]
    np.testing.assert_array_equal(df.drop(columns="set", level=0).to_numpy().astype(np.float64), g["v3_table"])


def test_context_shift_matches_reference(golden):
    from lightning_pose_amd.utils.predictions import PredictionHandler

    g = golden("predictions")
    pr = _batches(g.t("v4_kp"), g.t("v4_conf"), 16)
    cfg = _cfg(NAMES, "heatmap_mhcrnn")
    np.testing.assert_array_equal(PredictionHandler(cfg, video_file="c.mp4", frame_count=30)(preds=pr).to_numpy(), g["v4_table"])
    np.testing.assert_array_equal(PredictionHandler(cfg, video_file="c.mp4", frame_count=32)(preds=pr).to_numpy(), g["v4b_table"])


def test_constructor_errors_as_reference():
    from lightning_pose_amd.utils.predictions import PredictionHandler

    with pytest.raises(ValueError, match="data_module or video_file"):
        PredictionHandler(_cfg(NAMES))
    with pytest.raises(ValueError, match="keypoint_names"):
        PredictionHandler(_Cfg(data=_Cfg(), model=_Cfg(model_type="heatmap")), video_file="x.mp4")


def test_predict_video_end_to_end(stack_backend, tmp_path):
    """predict_video over the product tracker: eval mode, no tape, bbox map inside the fused decode; keypoints equal the
    oracle's decode of the SAME heat-maps mapped to the frame, the table round-trips through CSV, training mode restored."""
    dev = stack_backend
    from lightning_pose_amd.models import HeatmapTracker
    from lightning_pose_amd.utils.predictions import predict_batches, predict_video

    K, HW, S = 3, 128, 4
    model = HeatmapTracker(num_keypoints=K, backbone="resnet50", pretrained=False, torch_seed=3, device=dev)
    model.train()
    g = torch.Generator().manual_seed(5)
    bbox = torch.tensor([[10.0, 20.0, 256.0, 192.0]]).repeat(S, 1)  # x, y, h, w of the crop in the original frame
    loader = [{"frames": torch.randn(S, 3, HW, HW, generator=g).to(dev), "bbox": bbox.to(dev)} for _ in range(2)]
    names = [f"p{i}" for i in range(K)]
    out = tmp_path / "preds" / "clip.csv"
    (tmp_path / "preds").mkdir()
    df = predict_video("clip.mp4", model, loader, str(out), cfg=_cfg(names), frame_count=6)
    assert model.training  # restored
    assert df.shape == (6, 3 * K)
    back = pd.read_csv(out, header=[0, 1, 2], index_col=0)
    np.testing.assert_allclose(back.to_numpy(), df.to_numpy(), rtol=1e-12)
    # same heat-maps -> oracle decode -> bbox map (data/bboxes.py:74-105)
    res = predict_batches(model, loader, return_heatmaps=True)
    kp = torch.vstack([r[0] for r in res]).cpu()
    conf = torch.vstack([r[1] for r in res]).cpu()
    heat = torch.vstack([r[2] for r in res]).cpu()
    np.testing.assert_array_equal(df.to_numpy()[:, 0::3], kp[:6, 0::2].double().numpy())
    want_kp, want_conf = O.soft_argmax(heat, 2, 1000.0)
    want = want_kp.reshape(-1, K, 2).clone()
    want[..., 0] = want[..., 0] / HW * 192.0 + 10.0
    want[..., 1] = want[..., 1] / HW * 256.0 + 20.0
    # softmax(T=1000) amplifies fp32 rounding of the up-sampled logits (measured 3e-5 px here; full-size bound 1e-3 px)
    torch.testing.assert_close(kp.reshape(-1, K, 2), want, atol=2e-3, rtol=0)
    torch.testing.assert_close(conf, want_conf, atol=2e-5, rtol=0)


def test_decoded_video_to_prediction_table(stack_backend):
    """uint8 video array -> FrameWindowSource (windows of 4, padded tail) -> VideoFramePipeline -> predict_video: one row per REAL frame,
    keypoints in the original frame's pixel coordinates (bbox = the whole source frame)"""
    dev = stack_backend
    from lightning_pose_amd.data.producers import FrameWindowSource, VideoFramePipeline
    from lightning_pose_amd.models import HeatmapTracker
    from lightning_pose_amd.utils.predictions import predict_video

    K = 2
    g = torch.Generator().manual_seed(0)
    video = torch.randint(0, 256, (6, 40, 56, 3), generator=g, dtype=torch.uint8)  # 6 frames of 40 x 56
    model = HeatmapTracker(num_keypoints=K, backbone="resnet50", pretrained=False, torch_seed=1, device=dev)
    src = FrameWindowSource(video, sequence_length=4, device=dev)
    pipe = VideoFramePipeline([64, 64], imgaug="default")
    df = predict_video("clip.mp4", model, (pipe(w) for w in src), cfg=_cfg(["a", "b"]), frame_count=src.frame_count)
    assert df.shape == (6, 3 * K)  # 2 windows = 8 rows, the 2 zero-padded frames dropped
    xy = df.to_numpy()
    assert np.isfinite(xy).all()
    assert (xy[:, 0::3] >= -8).all() and (xy[:, 0::3] <= 56 + 8).all() and (xy[:, 1::3] >= -8).all() and (xy[:, 1::3] <= 40 + 8).all()
