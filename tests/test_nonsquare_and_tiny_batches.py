"""Shapes the BASELINE configs do not exercise, through the product stack in fp32 (emulated kernels on CPU / the device under -m gpu) against
the oracle's restated tracker: non-square frames (the reference's image_resize_dims are (height, width), e.g. 256 x 384 rigs), a labeled
batch of ONE frame (BatchNorm statistics over a single image), an unlabeled window whose rows do not end on a tile boundary (the joint pass
falls back to two passes)."""

import pytest
import torch

from oracle import restated as O


@pytest.mark.parametrize("H,W,Bl,S", [(64, 96, 1, 3), (96, 64, 2, 2), (32, 128, 3, 2)])
def test_nonsquare_semisupervised_step_fp32(stack_backend, H, W, Bl, S):
    from lightning_pose_amd.losses import LossFactory
    from lightning_pose_amd.models import SemiSupervisedHeatmapTracker

    dev = stack_backend
    K = 3
    g = torch.Generator().manual_seed(H * 7 + W)
    sup = LossFactory({"heatmap_mse": {"log_weight": 0.0}}, None)
    # (epsilon 5 px rectifies the temporal term of this random-init net to zero on both sides: frame-to-frame differences of ~1e-4 px between
    # nearly flat maps would otherwise contribute a gradient of unit vectors whose DIRECTION is rounding noise)
    unsup = LossFactory({"temporal": {"log_weight": 2.0, "epsilon": 5.0, "prob_threshold": 0.0}}, None)
    model = SemiSupervisedHeatmapTracker(num_keypoints=K, loss_factory=sup, loss_factory_unsupervised=unsup, backbone="resnet50",
                                         pretrained=False, torch_seed=13, device=dev, precision="fp32")
    ref = O.OracleTracker(K, 2, torch_seed=13)
    kp = torch.rand(Bl, K, 2, generator=g) * torch.tensor([W, H], dtype=torch.float32)
    eye = torch.tensor([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])
    batch = {
        "labeled": {"images": torch.randn(Bl, 3, H, W, generator=g), "keypoints": kp.reshape(Bl, 2 * K),
                    "heatmaps": O.generate_heatmaps(kp, H, W, (H // 4, W // 4)),
                    "bbox": torch.tensor([[3.0, 4.0, 2.0 * H, 2.0 * W]]).repeat(Bl, 1), "idxs": torch.arange(Bl)},
        "unlabeled": {"frames": torch.randn(S, 3, H, W, generator=g), "transforms": eye.unsqueeze(0),
                      "bbox": torch.tensor([[3.0, 4.0, 2.0 * H, 2.0 * W]]).repeat(S, 1), "is_multiview": False},
    }
    dbatch = {k: {kk: (vv.to(dev) if torch.is_tensor(vv) else vv) for kk, vv in v.items()} for k, v in batch.items()}
    ref.train()
    want_loss, want_logs = O.training_step(ref, batch, {"temporal": {"log_weight": 2.0, "epsilon": 5.0, "prob_threshold": 0.0}}, 1.0)
    want_loss.backward()
    model.train()
    model.total_unsupervised_importance = torch.tensor(1.0)
    model.configure_optimizers()["optimizer"].zero_grad()
    out = model.training_step(dbatch, 0)
    out["loss"].backward()
    got = {k: float(v) for k, v in model.logged.items()}
    assert set(got) == set(want_logs)
    # the heat-map loss is well conditioned at any weights: north_star's fp32 bar
    for k in ("train_heatmap_mse_loss", "train_supervised_loss"):
        assert got[k] == pytest.approx(float(want_logs[k].detach()), rel=1e-4), k
    assert got["train_temporal_loss"] == 0.0 == float(want_logs["train_temporal_loss"].detach())
    assert got["total_loss"] == pytest.approx(float(want_logs["total_loss"].detach()), rel=1e-4)
    # what the unsupervised losses saw: keypoints of nearly flat maps (soft-argmax at T = 1000 amplifies 1e-7 to ~1e-2 px), in frame pixels
    # through the (1, 2, 3) transform and the non-square bbox map
    seen = model.get_loss_inputs_unlabeled(dbatch["unlabeled"])
    with torch.no_grad():
        aug, conf = O.soft_argmax(ref(batch["unlabeled"]["frames"]), 2, 1000.0)
        kp_frame = O.model_to_frame(O.undo_affine(aug, eye, False), H, W, batch["unlabeled"]["bbox"], 1)
    torch.testing.assert_close(seen["keypoints_pred"].detach().cpu(), kp_frame, atol=0.1, rtol=0)
    torch.testing.assert_close(seen["confidences"].detach().cpu(), conf, atol=1e-4, rtol=1e-3)
    # the heat-maps themselves, train mode, both batches - shapes (h, w) = (H / 4, W / 4), not transposed
    with torch.no_grad():
        ref_heat = ref(batch["unlabeled"]["frames"])
        heat = model.forward(dbatch["unlabeled"]["frames"]).cpu()
    assert heat.shape == (S, K, H // 4, W // 4)
    torch.testing.assert_close(heat, ref_heat, atol=1e-4 * float(ref_heat.max()), rtol=1e-3)
    # parameter gradients of the supervised part reach the stem through the non-square geometry
    gw = getattr(model.head.upsampling_layers, "2").weight.grad.cpu()
    rw = getattr(ref.head.upsampling_layers, "2").weight.grad
    torch.testing.assert_close(gw, rw, atol=2e-3 * float(rw.abs().max()), rtol=2e-2)
    g1 = getattr(model.backbone, "0").weight.grad
    # (one labeled 64 x 96 frame: layer4's BatchNorm layers normalise over 2 x 3 = 6 values per channel, invstd is large and the backward
    # pass amplifies fp32 rounding differences on the way to the stem - 3 % here, 2e-3 with the 4-frame batch of test_fp32_parity.py)
    assert float(g1.norm()) == pytest.approx(float(getattr(ref.backbone, "0").weight.grad.norm()), rel=0.1 if Bl == 1 else 2e-2)
    assert bool(torch.isfinite(model.net.G).all())


def test_single_transform_shapes_and_table_lengths(stack_backend):
    """undo_affine_transform_batch: one matrix for the whole batch may arrive as (2, 3) or as (1, 2, 3) - the reference replicates a single
    inverse over the batch (data/utils.py:176-180) - and a per-frame table shorter than the batch is an error, not an out-of-bounds read
    (the (1, 2, 3) form used to be read as a per-frame table: NaN keypoints and NaN gradients; found by the test above)"""
    from lightning_pose_amd.data.bboxes import model_to_frame_batch  # noqa: F401  (imported for its side: registers nothing, shape check)
    from lightning_pose_amd.data.utils import undo_affine_transform_batch

    dev = stack_backend
    g = torch.Generator().manual_seed(3)
    kp = (torch.rand(5, 8, generator=g) * 50).to(dev)
    tf = torch.tensor([[0.9, 0.1, 3.0], [-0.2, 1.1, -4.0]])
    want = O.undo_affine(kp.cpu(), tf, False)
    for form in (tf, tf.unsqueeze(0), tf.unsqueeze(0).repeat(5, 1, 1)):
        got = undo_affine_transform_batch(kp, form.to(dev), False).cpu()
        torch.testing.assert_close(got, want, atol=1e-4, rtol=1e-5)
    with pytest.raises(ValueError, match="affine transforms for a batch"):
        undo_affine_transform_batch(kp, tf.unsqueeze(0).repeat(3, 1, 1).to(dev), False)


def test_multiview_transforms_as_the_dali_wrapper_stacks_them(stack_backend):
    """LitDaliWrapper stacks one (1, 2, 3) matrix per view (reference data/video/dali.py:311-314): transforms arrive as (V, 1, 2, 3), and
    undo_affine_transform_batch indexes transforms[v] - a (1, 2, 3) single transform - per view (data/utils.py:219-228)"""
    from lightning_pose_amd.data.utils import undo_affine_transform_batch

    dev = stack_backend
    g = torch.Generator().manual_seed(5)
    V, Kv, S = 2, 3, 4
    kp = (torch.rand(S, 2 * Kv * V, generator=g) * 60).to(dev)
    tf = torch.tensor([[[1.0, 0.05, 1.0], [-0.05, 1.0, 0.5]], [[0.9, 0.0, 2.0], [0.0, 1.1, -1.0]]])
    want = O.undo_affine(kp.cpu(), tf, True)
    for form in (tf, tf.unsqueeze(1)):
        got = undo_affine_transform_batch(kp, form.to(dev), True).cpu()
        torch.testing.assert_close(got, want, atol=1e-4, rtol=1e-5)


def test_predict_step_nonsquare_eval_mode(stack_backend):
    """eval-mode prediction (folded BatchNorm, no tape) on non-square frames: keypoints land in the frame's coordinate system through the
    non-square bbox map; compared with the oracle's eval forward + decode on the same weights"""
    from lightning_pose_amd.losses import LossFactory
    from lightning_pose_amd.models import HeatmapTracker

    dev = stack_backend
    K, H, W, B = 4, 64, 96, 3
    model = HeatmapTracker(num_keypoints=K, loss_factory=LossFactory({"heatmap_mse": {"log_weight": 0.0}}, None), backbone="resnet50",
                           pretrained=False, torch_seed=17, device=dev, precision="fp32")
    ref = O.OracleTracker(K, 2, torch_seed=17)
    g = torch.Generator().manual_seed(9)
    images = torch.randn(B, 3, H, W, generator=g)
    bbox = torch.tensor([[10.0, 20.0, 128.0, 288.0]]).repeat(B, 1)
    batch = {"images": images.to(dev), "keypoints": torch.zeros(B, 2 * K, device=dev), "bbox": bbox.to(dev), "idxs": torch.arange(B)}
    model.eval()
    ref.eval()
    with torch.no_grad():
        kp, conf = model.predict_step(batch, 0)
        heat = ref(images)
        aug, want_conf = O.soft_argmax(heat, 2, 1000.0)
        want = O.model_to_frame(aug, H, W, bbox, 1)
    assert kp.shape == (B, 2 * K) and conf.shape == (B, K)
    torch.testing.assert_close(conf.cpu(), want_conf, atol=1e-4, rtol=1e-3)
    torch.testing.assert_close(kp.cpu(), want, atol=0.3, rtol=0)   # (flat random-init maps: see the test above; 0.3 frame px = 0.1 model px)


def test_loss_wrappers_refuse_mismatched_shapes(stack_backend):
    """the C ABI takes raw pointers: every wrapper checks, on the host, the shapes the reference's own torch expressions would refuse -
    a mismatch is an error, never an out-of-bounds read"""
    from lightning_pose_amd import ops

    dev = stack_backend
    z = lambda *s: torch.zeros(*s, device=dev)  # noqa: E731
    with pytest.raises(ValueError, match="heat-map targets"):
        ops.heatmap_mse(z(2, 3, 8, 8), z(2, 3, 16, 16))
    with pytest.raises(ValueError, match="temporal loss"):
        ops.temporal_loss(z(4, 6), z(4, 2), torch.tensor(0.0), 0.0)
    with pytest.raises(ValueError, match="temporal epsilon"):
        ops.temporal_loss(z(4, 6), z(4, 3), torch.tensor([0.1, 0.2]), 0.0)
    with pytest.raises(ValueError, match="unimodal_mse"):
        ops.unimodal_mse(z(4, 6), z(4, 3, 8, 8), z(4, 2), 32, 32)
    with pytest.raises(ValueError, match="pca loss"):
        ops.pca_loss(z(4, 6), torch.zeros(1, 3, dtype=torch.int32, device=dev), z(4), z(2, 6), 0.0)
    with pytest.raises(ValueError, match="rmse"):
        ops.rmse(z(4, 6), z(4, 8))
    from lightning_pose_amd.engine import Engine

    eng = Engine(3, 2, dev)
    for bad in (z(2, 1, 64, 64), z(2, 3, 64)):
        with pytest.raises(ValueError, match="images must be"):
            eng.forward(bad, training=True)
        with pytest.raises(ValueError, match="images must be"):
            eng.forward_infer(bad)
    with pytest.raises(ValueError, match="bounding boxes"):
        ops.decode(torch.full((4, 3, 8, 8), 1 / 64, device=dev), 2, 1000.0, ops.DecodeFrameMap(None, False, z(3, 4) + 1, 1, 32, 32, 3))


@pytest.mark.gpu
def test_frame_map_tables_left_on_the_host_are_moved():
    """transforms / bbox handed over as CPU tensors with device heat-maps (the reference moves its inverse matrices to the keypoints' device,
    data/utils.py:170-172): same keypoints as with device tables - not a host pointer dereferenced by a kernel"""
    from lightning_pose_amd import ops

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(2)
    heat = torch.softmax(torch.randn(3, 4, 16 * 16, generator=g) * 4, -1).reshape(3, 4, 16, 16).to(dev)
    tf = torch.tensor([[0.9, 0.1, 3.0], [-0.2, 1.1, -4.0]])
    bbox = torch.tensor([[5.0, 6.0, 100.0, 120.0]]).repeat(3, 1)
    want = ops.decode(heat, 2, 1000.0, ops.DecodeFrameMap(tf.to(dev), False, bbox.to(dev), 1, 64, 64, 4))
    got = ops.decode(heat, 2, 1000.0, ops.DecodeFrameMap(tf, False, bbox, 1, 64, 64, 4))
    for a, b in zip(got, want):
        assert torch.equal(a, b)
