"""Host side of the batch producers over the kernel library: VideoFramePipeline mirrors the reference's DALI
video_pipe + _dali_output_to_tensors (data/video/dali.py:70-197, :267-330), LabeledBatchProducer the per-batch work of
HeatmapDataset (data/datasets.py:262-376, :496-550); both feed the tracker's training step directly."""

import numpy as np
import pytest
import torch

from oracle import restated as O


def _u8(seed, s, h, w):
    g = torch.Generator().manual_seed(seed)
    ys, xs = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing="ij")
    base = 120 + 90 * torch.sin(xs / 6.0 + torch.arange(s).view(s, 1, 1) * 0.3) * torch.cos(ys / 9.0)
    return (base.unsqueeze(-1) + 30 * torch.randn(s, h, w, 3, generator=g)).clamp(0, 255).to(torch.uint8)


def test_video_pipeline_default_is_resize_normalise_and_sentinel(stack_backend):
    from lightning_pose_amd.data.producers import VideoFramePipeline

    dev = stack_backend
    src = _u8(1, 5, 90, 120)
    pipe = VideoFramePipeline(resize_dims=[128, 128], imgaug="default")
    out = pipe(src.to(dev))
    assert out["is_multiview"] is False
    assert tuple(out["frames"].shape) == (5, 3, 128, 128)
    assert out["transforms"].cpu().tolist() == [-1.0]  # nothing geometric to undo (data/video/dali.py:170-172)
    assert out["bbox"].cpu().tolist() == [[0.0, 0.0, 90.0, 120.0]] * 5  # x, y, h, w of the ORIGINAL frame (:287-294)
    want = O.frames_finish(O.frames_resize(src, 128, 128, "clamp"))
    torch.testing.assert_close(out["frames"].cpu(), want, atol=5e-5, rtol=0)


def test_video_pipeline_dlc_matches_restated_operators(stack_backend):
    from lightning_pose_amd.data.producers import VideoFramePipeline, rotation_scale_matrix

    dev = stack_backend
    src = _u8(2, 4, 100, 80)
    pipe = VideoFramePipeline(resize_dims=[128, 128], imgaug="dlc", seed=5)
    m = rotation_scale_matrix(8.0, (1.1, 0.9), (64.0, 64.0))
    out = pipe(src.to(dev), params={"matrix": m, "brightness": 1.1, "contrast": 0.9, "shot_factor": 0.0})
    torch.testing.assert_close(out["transforms"].cpu(), torch.from_numpy(m).float())
    resized = O.frames_resize(src, 128, 128, "clamp")
    want = O.frames_finish(O.brightness_contrast(O.frames_warp_affine(resized, torch.from_numpy(m)), 1.1, 0.9))
    torch.testing.assert_close(out["frames"].cpu(), want, atol=3e-4, rtol=0)
    # the matrix handed on is the one the step's decode undoes: a point moved by it and then through undo_affine comes back
    pts = torch.tensor([[[30.0, 40.0], [100.0, 17.0]]])
    moved = torch.cat([pts, torch.ones(1, 2, 1)], -1) @ torch.from_numpy(m).float().T
    back = O.undo_affine(moved.reshape(1, 4), torch.from_numpy(m).float())
    torch.testing.assert_close(back.reshape(1, 2, 2), pts, atol=1e-3, rtol=0)
    # random draws: inside the reference's ranges, reproducible per seed, different across calls
    p1, p2 = VideoFramePipeline([128, 128], imgaug="dlc", seed=9), VideoFramePipeline([128, 128], imgaug="dlc", seed=9)
    d1, d1b, d2 = p1._draw(128, 128), p1._draw(128, 128), p2._draw(128, 128)
    np.testing.assert_array_equal(d1["matrix"], d2["matrix"])
    assert not np.array_equal(d1["matrix"], d1b["matrix"])
    for d in (d1, d1b):
        assert 0.75 <= d["brightness"] <= 1.25 and 0.75 <= d["contrast"] <= 1.25 and 0.0 <= d["shot_factor"] <= 10.0
        sx, sy = np.linalg.norm(d["matrix"][0, :2]), np.linalg.norm(d["matrix"][1, :2])
        assert 0.8 - 1e-9 <= sx <= 1.2 + 1e-9 and 0.8 - 1e-9 <= sy <= 1.2 + 1e-9
    with pytest.raises(NotImplementedError):
        VideoFramePipeline([128, 128], imgaug="unknown")


def test_video_pipeline_multiview_layout(stack_backend):
    from lightning_pose_amd.data.producers import VideoFramePipeline

    dev = stack_backend
    views = [_u8(3, 3, 60, 80).to(dev), _u8(4, 3, 70, 50).to(dev)]
    out = VideoFramePipeline([128, 128], imgaug="default")(views)
    assert out["is_multiview"] is True
    assert tuple(out["frames"].shape) == (3, 2, 3, 128, 128)
    assert tuple(out["transforms"].shape) == (2, 1, 1)  # the per-view "nothing to undo" sentinel (data/datatypes.py:227-236)
    assert out["bbox"].cpu().tolist() == [[0.0, 0.0, 60.0, 80.0, 0.0, 0.0, 70.0, 50.0]] * 3
    out_aug = VideoFramePipeline([128, 128], imgaug="dlc")(views)
    assert tuple(out_aug["transforms"].shape) == (2, 2, 3)
    # both forms go straight into a multiview tracker: (S, V, 3, H, W) frames -> (S, K*V) keypoints in each view's frame coordinates
    from lightning_pose_amd.losses import LossFactory
    from lightning_pose_amd.models import SemiSupervisedHeatmapTracker

    model = SemiSupervisedHeatmapTracker(num_keypoints=2, loss_factory=LossFactory({"heatmap_mse": {"log_weight": 0.0}}, None),
                                         loss_factory_unsupervised=LossFactory({"temporal": {"log_weight": 0.0}}, None), backbone="resnet50",
                                         pretrained=False, torch_seed=0, device=dev)
    model.eval()
    with torch.no_grad():
        for batch in (out, out_aug):
            data = model.get_loss_inputs_unlabeled(batch)
            assert tuple(data["heatmaps_pred"].shape) == (3, 4, 32, 32) and tuple(data["keypoints_pred"].shape) == (3, 8)
            assert torch.isfinite(data["keypoints_pred"]).all() and torch.isfinite(data["confidences"]).all()


def test_labeled_producer_matches_verbatim_dataset_targets(stack_backend, golden):
    from lightning_pose_amd.data.producers import LabeledBatchProducer

    dev = stack_backend
    g = golden("labeled_targets")
    prod = LabeledBatchProducer(256, 256, downsample_factor=2, uniform_heatmaps=True)
    imgs = _u8(5, 8, 406, 396)
    batch = prod(imgs.to(dev), g.t("kp_src").reshape(8, -1).to(dev), affine=g.t("affine").to(dev))
    assert tuple(batch["images"].shape) == (8, 3, 256, 256) and tuple(batch["heatmaps"].shape) == (8, 17, 64, 64)
    assert batch["bbox"].cpu().tolist() == [[0.0, 0.0, 406.0, 396.0]] * 8 and batch["idxs"].tolist() == list(range(8))
    kp = batch["keypoints"].cpu().reshape(8, 17, 2).numpy()
    np.testing.assert_allclose(np.nan_to_num(kp), np.nan_to_num(g["u1_kp_model_nan"]), atol=5e-5)
    np.testing.assert_allclose(batch["heatmaps"].cpu().numpy(), g["u1_heatmaps"], atol=2e-6)
    # sample 0 has the identity affine: its image is the plain resize
    want0 = O.frames_finish(O.frames_resize_cubic(imgs[:1], 256, 256))   # imgaug's Resize default: cubic, uint8 levels
    diff = (batch["images"][:1].cpu() - want0).abs()
    assert float(diff.max()) <= 1.01 / 255 / 0.224 and float((diff > 3e-4).float().mean()) < 2e-3   # (one level at an exact .5, rarely)
    with pytest.raises(ValueError):
        LabeledBatchProducer(250, 256)


def test_labeled_producer_flip_moves_image_and_labels_together(stack_backend):
    from lightning_pose_amd.data.producers import LabeledBatchProducer

    dev = stack_backend
    prod = LabeledBatchProducer(128, 128, hflip_swap_indices=[1, 0, 2], interpolation="linear")
    imgs = _u8(6, 2, 128, 128)
    kp = torch.tensor([[10.0, 20.0, 100.0, 30.0, 64.0, 64.0]] * 2)
    batch = prod(imgs.to(dev), kp.to(dev), hflip=torch.tensor([1, 0]))
    got = batch["keypoints"].cpu()
    torch.testing.assert_close(got[0], torch.tensor([28.0, 30.0, 118.0, 20.0, 64.0, 64.0]))  # x -> W - x, then left <-> right
    torch.testing.assert_close(got[1], kp[1])
    plain = O.frames_finish(O.frames_resize(imgs, 128, 128, "renorm"))
    torch.testing.assert_close(batch["images"][0].cpu(), plain[0].flip(-1), atol=3e-4, rtol=0)
    torch.testing.assert_close(batch["images"][1].cpu(), plain[1], atol=3e-4, rtol=0)
    with pytest.raises(ValueError, match="permutation"):
        LabeledBatchProducer(128, 128, hflip_swap_indices=[1, 0, 5])(imgs.to(dev), kp.to(dev), hflip=torch.tensor([1, 0]))


def test_producers_feed_the_training_step(stack_backend):
    """uint8 frames + stored labels -> producers -> SemiSupervisedHeatmapTracker.training_step, nothing in between"""
    from lightning_pose_amd.data.producers import LabeledBatchProducer, VideoFramePipeline
    from lightning_pose_amd.losses import LossFactory
    from lightning_pose_amd.models import SemiSupervisedHeatmapTracker

    dev = stack_backend
    K = 3
    sup = LossFactory({"heatmap_mse": {"log_weight": 0.0}}, None)
    unsup = LossFactory({"temporal": {"log_weight": 2.0, "epsilon": 1.0, "prob_threshold": 0.0}}, None)
    model = SemiSupervisedHeatmapTracker(num_keypoints=K, loss_factory=sup, loss_factory_unsupervised=unsup, backbone="resnet50",
                                         pretrained=False, torch_seed=1, device=dev)
    g = torch.Generator().manual_seed(0)
    labeled = LabeledBatchProducer(128, 128)(_u8(7, 2, 150, 170).to(dev), (torch.rand(2, 2 * K, generator=g) * 150).to(dev))
    unlabeled = VideoFramePipeline([128, 128], imgaug="dlc", seed=3)(_u8(8, 3, 150, 170).to(dev))
    model.train()
    opt = model.configure_optimizers()["optimizer"]
    opt.zero_grad()
    loss = model.training_step({"labeled": labeled, "unlabeled": unlabeled}, 0)["loss"]
    loss.backward()
    opt.step()
    assert torch.isfinite(loss).item()
    assert float(model.logged["train_heatmap_mse_loss"]) > 0


def test_temporal_heatmap_losses_train_through_the_tracker(stack_backend):
    """temporal_heatmap_mse / _kl as unsupervised losses of the semi-supervised tracker: value equals the oracle formula on the
    tracker's own heat-maps, and the gradient reaches the trunk"""
    from lightning_pose_amd.losses import LossFactory
    from lightning_pose_amd.models import SemiSupervisedHeatmapTracker

    dev = stack_backend
    K, HW = 3, 64
    sup = LossFactory({"heatmap_mse": {"log_weight": 0.0}}, None)
    unsup = LossFactory({"temporal_heatmap_mse": {"loss_name": "temporal_heatmap_mse", "log_weight": 0.0, "prob_threshold": 0.0},
                         "temporal_heatmap_kl": {"loss_name": "temporal_heatmap_kl", "log_weight": 0.0, "epsilon": 1e-6}}, None)
    model = SemiSupervisedHeatmapTracker(num_keypoints=K, loss_factory=sup, loss_factory_unsupervised=unsup, backbone="resnet50",
                                         pretrained=False, torch_seed=2, device=dev)
    g = torch.Generator().manual_seed(1)
    batch = {"frames": torch.randn(4, 3, HW, HW, generator=g).to(dev), "transforms": torch.tensor([-1.0]).to(dev),
             "bbox": torch.tensor([[0.0, 0.0, HW, HW]]).repeat(4, 1).to(dev), "is_multiview": False}
    model.train()
    opt = model.configure_optimizers()["optimizer"]
    opt.zero_grad()
    data = model.get_loss_inputs_unlabeled(batch)
    loss, logs = model.loss_factory_unsup(stage="train", anneal_weight=1.0, **data)
    loss.backward()
    hm = data["heatmaps_pred"].detach().cpu()
    want_mse = ((hm[1:] - hm[:-1]) ** 2).mean((-1, -2)).mean()
    kl = ((hm[1:] + 1e-10) * (torch.log(hm[1:] + 1e-10) - torch.log(hm[:-1] + 1e-10))).sum((-1, -2))
    want_kl = torch.relu(kl - 1e-6).mean()
    got = {d["name"]: float(torch.as_tensor(d["value"]).detach()) for d in logs}
    assert got["train_temporal_heatmap_mse_loss"] == pytest.approx(float(want_mse), rel=1e-4)
    assert got["train_temporal_heatmap_kl_loss"] == pytest.approx(float(want_kl), rel=1e-4, abs=1e-9)
    assert float(model.net.G.abs().sum()) > 0


def test_frame_window_source_sequences_like_the_video_reader(stack_backend):
    from lightning_pose_amd.data.producers import FrameWindowSource, VideoFramePipeline

    dev = stack_backend
    vid_a = (torch.arange(10).view(10, 1, 1, 1) + torch.zeros(10, 4, 6, 3)).to(torch.uint8)        # frame index in every pixel
    vid_b = (100 + torch.arange(7).view(7, 1, 1, 1) + torch.zeros(7, 4, 6, 3)).to(torch.uint8)
    # prediction: in order, the tail window padded with zero frames, windows never span two videos
    src = FrameWindowSource([vid_a.numpy(), vid_b], sequence_length=4, random_shuffle=False, pad_sequences=True, device=dev)
    firsts = [w[:, 0, 0, 0].cpu().tolist() for w in src]
    assert firsts == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 0, 0], [100, 101, 102, 103], [104, 105, 106, 0]]
    assert len(src) == 5 and src.frame_count == 10
    # no padding: incomplete tails are dropped; a step smaller than the window overlaps windows (context loaders: step = length - 4)
    src = FrameWindowSource(vid_a, sequence_length=6, step=2, pad_sequences=False, device=dev)
    assert [w[0, 0, 0, 0].item() for w in src] == [0, 2, 4]
    # training: a seeded permutation per epoch, reproducible, all windows exactly once
    s1 = FrameWindowSource([vid_a, vid_b], 4, random_shuffle=True, seed=7, device=dev)
    s2 = FrameWindowSource([vid_a, vid_b], 4, random_shuffle=True, seed=7, device=dev)
    e1, e1b, e2 = [w[0, 0, 0, 0].item() for w in s1], [w[0, 0, 0, 0].item() for w in s1], [w[0, 0, 0, 0].item() for w in s2]
    assert e1 == e2 and sorted(e1) == [0, 4, 8, 100, 104] and sorted(e1b) == sorted(e1)
    # windows go straight into the frame pipeline
    out = VideoFramePipeline([32, 32], imgaug="default")(next(iter(FrameWindowSource(vid_a, 4, device=dev))))
    assert tuple(out["frames"].shape) == (4, 3, 32, 32) and out["bbox"].cpu().tolist() == [[0.0, 0.0, 4.0, 6.0]] * 4
    with pytest.raises(ValueError):
        FrameWindowSource(torch.zeros(3, 4, 6, 3), 2, device=dev)  # not uint8


def test_host_stager_copies_through_reused_pinned_buffers(stack_backend):
    """HostStager (round 4): pageable host tensors of changing sizes go through two reusable pinned buffers on a copy stream and arrive
    intact, in order, on the consumer's stream; pinned inputs and device tensors pass through"""
    from lightning_pose_amd.data.producers import HostStager

    dev = stack_backend
    stage = HostStager(dev)
    hosts = [_u8(20 + i, 2 + (i % 3), 40 + 8 * i, 56) for i in range(6)]
    outs = [stage(h) for h in hosts]                 # six copies in flight over two buffers: reuse waits for the older copy
    for h, o in zip(hosts, outs):
        assert o.device.type == torch.device(dev).type and torch.equal(o.cpu(), h)
    f = torch.arange(12, dtype=torch.float32).reshape(3, 4)
    assert torch.equal(stage(f).cpu(), f)            # any dtype
    if torch.device(dev).type == "cuda":
        p = hosts[0].pin_memory()
        assert torch.equal(stage(p).cpu(), hosts[0])
        d = hosts[1].to(dev)
        assert stage(d) is not None and torch.equal(stage(d).cpu(), hosts[1])
