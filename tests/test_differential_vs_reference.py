"""Differential tests: the product's drop-in functions and loss classes against the VERBATIM reference modules (executed from
/root/reference in the build container, from the verbatim copy oracle/_ref that oracle/make_ref.py ships on the GPU box) on randomly drawn
inputs - the golden fixtures pin single draws, this pins the interface behaviour (argument shapes, masks, broadcasting rules, logged names)
over many.  Every test runs twice: on the CPU-emulated kernels (default suite) and, under `-m gpu`, with the product's tensors on cuda:0 -
the HIP kernels through liblp_hip.so - next to the reference executed on the host; the comparison happens on the host."""

import numpy as np
import pytest
import torch

from tests.conftest import needs_reference

pytestmark = [needs_reference, pytest.mark.reference]


@pytest.fixture()
def dev(stack_backend):
    """device of the PRODUCT's tensors (cpu + emulated kernels, or cuda:0 + liblp_hip.so); the reference always runs on the host"""
    return stack_backend


def _d(x, dev):
    """a fresh copy of a (possibly nested) input on the product's device"""
    if torch.is_tensor(x):
        return x.detach().clone().to(dev)
    if isinstance(x, dict):
        return {k: _d(v, dev) for k, v in x.items()}
    return x


def _ref(name):
    from oracle import ref_loader as R

    R.install_stubs()
    return R.load(name)


@pytest.mark.parametrize("seed", range(6))
def test_undo_affine_transform_batch_every_shape(dev, seed):
    from lightning_pose_amd.data.utils import undo_affine_transform_batch

    U = _ref("data.utils")
    r = np.random.default_rng(seed)
    S, K = int(r.integers(1, 7)), int(r.integers(1, 6))
    kp = torch.from_numpy(r.uniform(0, 80, (S, 2 * K)).astype(np.float32))

    def mat():
        a = np.eye(2) + r.normal(0, 0.2, (2, 2))
        return torch.from_numpy(np.concatenate([a, r.normal(0, 5, (2, 1))], 1).astype(np.float32))

    forms = [mat(), mat().unsqueeze(0), torch.stack([mat() for _ in range(S)]), torch.tensor([1.0]), torch.ones(S, 1)]
    for tf in forms:
        want = U.undo_affine_transform_batch(kp.clone(), tf.clone(), False)
        got = undo_affine_transform_batch(_d(kp, dev), _d(tf, dev), False).cpu()
        torch.testing.assert_close(got, want, atol=2e-4, rtol=1e-5)
    V = int(r.integers(2, 4))
    kpv = torch.from_numpy(r.uniform(0, 80, (S, 2 * K * V)).astype(np.float32))
    tfv = torch.stack([mat() for _ in range(V)])
    for tf in (tfv, tfv.unsqueeze(1)):      # (V, 2, 3) and (V, 1, 2, 3): what the DALI wrapper stacks
        want = U.undo_affine_transform_batch(kpv.clone(), tf.clone(), True)
        got = undo_affine_transform_batch(_d(kpv, dev), _d(tf, dev), True).cpu()
        torch.testing.assert_close(got, want, atol=2e-4, rtol=1e-5)


@pytest.mark.parametrize("seed", range(6))
def test_model_to_frame_batch(dev, seed):
    from lightning_pose_amd.data.bboxes import model_to_frame_batch

    Bx = _ref("data.bboxes")
    r = np.random.default_rng(100 + seed)
    B, K, V = int(r.integers(1, 6)), int(r.integers(1, 5)), int(r.integers(1, 4))
    H, W = 32 * int(r.integers(1, 5)), 32 * int(r.integers(1, 5))
    kp = torch.from_numpy(r.uniform(0, min(H, W), (B, 2 * K * V)).astype(np.float32))
    bbox = torch.from_numpy(np.concatenate([r.uniform(0, 50, (B, V, 2)), r.uniform(60, 400, (B, V, 2))], 2).reshape(B, 4 * V).astype(np.float32))
    if V == 1:
        batches = [{"images": torch.zeros(B, 3, H, W), "bbox": bbox}, {"frames": torch.zeros(B, 3, H, W), "bbox": bbox, "is_multiview": False}]
    else:
        batches = [{"images": torch.zeros(B, V, 3, H, W), "bbox": bbox, "num_views": torch.full((B,), V)},
                   {"frames": torch.zeros(B, V, 3, H, W), "bbox": bbox, "is_multiview": True}]
    for bd in batches:
        want = Bx.model_to_frame_batch(bd, kp.clone(), in_place=False)
        got = model_to_frame_batch(_d(bd, dev), _d(kp, dev)).cpu()
        torch.testing.assert_close(got, want, atol=1e-3, rtol=1e-5)


@pytest.mark.parametrize("seed", range(6))
def test_heatmap_functions(dev, seed):
    from lightning_pose_amd.data.heatmaps import evaluate_heatmaps_at_location, generate_heatmaps

    Hm = _ref("data.heatmaps")
    r = np.random.default_rng(200 + seed)
    B, K = int(r.integers(1, 5)), int(r.integers(1, 6))
    h, w = int(r.integers(8, 40)), int(r.integers(8, 40))
    H, W = 4 * h, 4 * w
    kp = torch.from_numpy(r.uniform(-20, 1.1 * max(H, W), (B, K, 2)).astype(np.float32))
    kp[0, 0] = float("nan")
    vis = torch.from_numpy(r.integers(0, 3, (B, K)).astype(np.int64))
    sigma = float(r.uniform(1.0, 3.0))   # (floor(sigma * num_stds) == 0 is an error in the reference itself: an empty padded slice)
    for v in (None, vis):
        want = Hm.generate_heatmaps(kp.clone(), H, W, (h, w), sigma=sigma, visibility=v)
        got = generate_heatmaps(_d(kp, dev), H, W, (h, w), sigma=sigma, visibility=_d(v, dev)).cpu()
        torch.testing.assert_close(got, want, atol=3e-7, rtol=1e-4)
    heat = torch.softmax(torch.from_numpy(r.normal(0, 2, (B, K, h * w)).astype(np.float32)), -1).reshape(B, K, h, w)
    # (locations inside the map: the reference indexes its padded copy directly and raises IndexError beyond it; the kernel zero-pads)
    locs = torch.from_numpy((r.uniform(0, 1, (B, K, 2)) * np.array([w - 1e-3, h - 1e-3])).astype(np.float32))
    for ns in (1, 2):
        want = Hm.evaluate_heatmaps_at_location(heat, locs.clone(), sigma=sigma, num_stds=ns)
        got = evaluate_heatmaps_at_location(_d(heat, dev), _d(locs, dev), sigma=sigma, num_stds=ns).cpu()
        torch.testing.assert_close(got, want, atol=2e-6, rtol=1e-5)


def _logs(pairs):
    return {d["name"]: float(d["value"].detach() if torch.is_tensor(d["value"]) else d["value"]) for d in pairs}


@pytest.mark.parametrize("seed", range(6))
def test_loss_classes_values_gradients_and_logs(dev, seed):
    from lightning_pose_amd.losses import losses as P

    L = _ref("losses.losses")
    r = np.random.default_rng(300 + seed)
    S, K = int(r.integers(2, 9)), int(r.integers(1, 7))
    h, w = int(r.integers(6, 20)), int(r.integers(6, 20))

    def compare(ref_loss, prod_loss, kwargs, grad_key, stage="train"):
        a = {k: (v.clone().requires_grad_(k == grad_key) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in kwargs.items()}
        b = {k: (_d(v, dev).requires_grad_(k == grad_key) if torch.is_tensor(v) and v.is_floating_point() else _d(v, dev)) for k, v in kwargs.items()}
        want, want_logs = ref_loss(stage=stage, **a)
        got, got_logs = prod_loss(stage=stage, **b)
        assert float(got.detach()) == pytest.approx(float(want.detach()), rel=3e-5, abs=1e-8)
        assert _logs(got_logs).keys() == _logs(want_logs).keys()
        for k_, v_ in _logs(want_logs).items():
            assert _logs(got_logs)[k_] == pytest.approx(v_, rel=3e-5, abs=1e-8), k_
        if float(want) != 0.0 and torch.isfinite(want):
            want.backward()
            got.backward()
            torch.testing.assert_close(b[grad_key].grad.cpu(), a[grad_key].grad, atol=3e-6 * float(a[grad_key].grad.abs().max()) + 1e-12, rtol=3e-4)

    # temporal: scalar / per-keypoint epsilon, with and without a confidence threshold
    kp = torch.from_numpy(r.uniform(0, 60, (S, 2 * K)).astype(np.float32))
    conf = torch.from_numpy(r.uniform(0, 1, (S, K)).astype(np.float32))
    for eps in (0.0, float(r.uniform(0, 5)), [float(v) for v in r.uniform(0, 8, K)]):
        for thr in (0.0, 0.4):
            kw = dict(epsilon=eps, prob_threshold=thr, log_weight=float(r.uniform(-1, 3)))
            compare(L.TemporalLoss(**kw), P.TemporalLoss(**kw), {"keypoints_pred": kp, "confidences": conf}, "keypoints_pred")
    compare(L.TemporalLoss(epsilon=1.0), P.TemporalLoss(epsilon=1.0), {"keypoints_pred": kp}, "keypoints_pred", stage="val")
    # heat-map losses: some maps unlabeled (all-zero targets)
    targ = torch.softmax(torch.from_numpy(r.normal(0, 3, (S, K, h * w)).astype(np.float32)), -1).reshape(S, K, h, w)
    drop = torch.from_numpy(r.uniform(0, 1, (S, K)) < 0.3)
    drop[0, 0] = False
    targ[drop] = 0.0
    pred = torch.softmax(torch.from_numpy(r.normal(0, 1, (S, K, h * w)).astype(np.float32)), -1).reshape(S, K, h, w)
    for name in ("HeatmapMSELoss", "HeatmapKLLoss", "HeatmapJSLoss"):
        lw = float(r.uniform(-1, 2))
        compare(getattr(L, name)(log_weight=lw), getattr(P, name)(log_weight=lw), {"heatmaps_targ": targ, "heatmaps_pred": pred}, "heatmaps_pred")
    # RMSE metric: NaN pairs are unlabeled keypoints
    kt = torch.from_numpy(r.uniform(0, 60, (S, 2 * K)).astype(np.float32))
    miss = torch.from_numpy(r.uniform(0, 1, (S, K)) < 0.3)
    miss[0, 0] = False
    kt.reshape(S, K, 2)[miss] = float("nan")
    want, _ = L.RegressionRMSELoss()(keypoints_targ=kt, keypoints_pred=kp, stage=None)
    got, _ = P.RegressionRMSELoss()(keypoints_targ=_d(kt, dev), keypoints_pred=_d(kp, dev), stage=None)
    assert float(got) == pytest.approx(float(want), rel=1e-5)


@pytest.mark.parametrize("seed", range(6))
def test_run_subpixelmaxima_random_maps(dev, seed):
    """the fused decode (bicubic x2^ds upsample + 5 x 5 blur + soft-argmax at T = 1000 + confidence window) against the verbatim
    ``run_subpixelmaxima`` on random, moderately peaked maps of random (non-square) sizes"""
    from lightning_pose_amd import ops

    hm = _ref("models.heads.heatmap")
    r = np.random.default_rng(400 + seed)
    ds = int(r.choice([1, 2, 3]))
    lo = {1: 8, 2: 9, 3: 11}[ds]
    B, K, h, w = int(r.integers(1, 4)), int(r.integers(1, 5)), int(r.integers(lo, 40)), int(r.integers(lo, 56))
    heat = torch.softmax(torch.from_numpy(r.normal(0, float(r.uniform(2, 5)), (B, K, h * w)).astype(np.float32)), -1).reshape(B, K, h, w)
    want_kp, want_conf = hm.run_subpixelmaxima(heat.clone(), ds, torch.tensor(1000.0))
    fm = ops.DecodeFrameMap(None, False, None, 1, h << ds, w << ds, K)
    kp, _, conf = (t if t is None else t.cpu() for t in ops.decode(_d(heat, dev), ds, 1000.0, fm))
    torch.testing.assert_close(conf, want_conf, atol=3e-5, rtol=1e-4)
    # soft-argmax at T = 1000 is as well conditioned as the map is peaked: 1e-3 px on these random maps, 1e-4 px on the golden (fitted) ones
    torch.testing.assert_close(kp, want_kp, atol=2e-3, rtol=0)


@pytest.mark.parametrize("seed", range(4))
def test_pca_losses_with_the_reference_fit(dev, seed):
    """PCALoss (single- and multi-view) with parameters fitted by the verbatim KeypointPCA: the product's own fit (utils/pca.py) gives the
    same parameters, and the loss class the same value, logs and gradient"""
    from lightning_pose_amd.losses import losses as P
    from oracle import ref_loader as R

    R.install_stubs()
    L = R.load("losses.losses")
    r = np.random.default_rng(500 + seed)
    K, N, S = int(r.integers(4, 9)), 60, int(r.integers(2, 7))
    basis = r.normal(0, 1, (3, 2 * K))
    data = torch.from_numpy((r.normal(0, 6, (N, 3)) @ basis + 40 + r.normal(0, 0.3, (N, 2 * K))).astype(np.float32))
    cols = sorted(r.choice(K, size=int(r.integers(2, K + 1)), replace=False).tolist())
    kp = torch.from_numpy((r.normal(0, 6, (S, 3)) @ basis + 40 + r.normal(0, 2.0, (S, 2 * K))).astype(np.float32))
    ref_pca = R.fit_keypoint_pca("pca_singleview", data, components_to_keep=0.99, columns_for_singleview_pca=cols)
    prod = P.PCALoss(loss_name="pca_singleview", components_to_keep=0.99, columns_for_singleview_pca=cols, data_arr=data, device=str(dev),
                     log_weight=1.0)
    for key in ("mean", "kept_eigenvectors"):
        a, b = prod.pca.parameters[key].float().cpu(), ref_pca.parameters[key].float().cpu()
        if key == "kept_eigenvectors":     # principal axes are defined up to sign
            sgn = torch.sign((a * b).sum(1, keepdim=True))
            a = a * sgn
        # both fits run in float64 numpy and hand over float32: the mean (values ~40) to a few ulp, the axes to the last bit or two
        torch.testing.assert_close(a, b, atol=5e-5 if key == "mean" else 2e-6, rtol=0)
    assert float(prod.pca.parameters["epsilon"]) == pytest.approx(float(ref_pca.parameters["epsilon"]), rel=5e-5)
    ref_loss = L.PCALoss.__new__(L.PCALoss)      # the reference class around the reference fit, without a data module
    L.Loss.__init__(ref_loss, epsilon=float(ref_pca.parameters["epsilon"]), log_weight=1.0)
    ref_loss.loss_name, ref_loss.pca, ref_loss.device = "pca_singleview", ref_pca, "cpu"
    a, b = kp.clone().requires_grad_(True), _d(kp, dev).requires_grad_(True)
    want, want_logs = ref_loss(keypoints_pred=a, stage="train")
    got, got_logs = prod(keypoints_pred=b, stage="train")
    assert float(got.detach()) == pytest.approx(float(want.detach()), rel=1e-4, abs=1e-6)   # (north_star's fp32 bar; measured 2e-6)
    assert _logs(got_logs).keys() == _logs(want_logs).keys()
    if float(want.detach()) > 0:
        want.backward()
        got.backward()
        torch.testing.assert_close(b.grad.cpu(), a.grad, atol=1e-4 * float(a.grad.abs().max()), rtol=1e-3)   # (measured 5e-6)


@pytest.mark.parametrize("K,H,W,ds,B", [(2, 96, 96, 3, 2), (5, 64, 96, 2, 3), (1, 96, 64, 2, 2), (3, 64, 64, 1, 2)])
def test_supervised_tracker_variants_fp32(dev, K, H, W, ds, B):
    """The reference's own HeatmapTracker (verbatim models/heatmap_tracker.py + heads + losses) and the product's, same seed (so the same
    weights by construction order), one supervised step in fp32 on configurations the golden steps do not cover: downsample_factor 3 (one
    upsampling layer less, heat-maps H / 8), non-square frames, a single keypoint (17 keypoints: the golden steps).  Compared: state_dict names and values
    before the step, heat-maps, every logged scalar that is well conditioned on a random-init net, head gradients, predict_step."""
    from lightning_pose_amd.losses import LossFactory
    from lightning_pose_amd.models import HeatmapTracker
    from oracle import ref_loader as R

    R.install_stubs()
    T, Fa, Hm = R.load("models.heatmap_tracker"), R.load("losses.factory"), R.load("data.heatmaps")
    ref = T.HeatmapTracker(num_keypoints=K, loss_factory=Fa.LossFactory({"heatmap_mse": {"log_weight": 0.0}}, None), backbone="resnet50",
                           pretrained=False, torch_seed=21, downsample_factor=ds, image_size=max(H, W))
    model = HeatmapTracker(num_keypoints=K, loss_factory=LossFactory({"heatmap_mse": {"log_weight": 0.0}}, None), backbone="resnet50",
                           pretrained=False, torch_seed=21, downsample_factor=ds, device=dev, precision="fp32")
    sd_ref, sd = ref.state_dict(), model.state_dict()
    assert sorted(k for k in sd_ref if "num_batches_tracked" not in k) == sorted(k for k in sd if "num_batches_tracked" not in k)
    for k_, v_ in sd_ref.items():
        if "num_batches_tracked" not in k_:
            torch.testing.assert_close(sd[k_].cpu().reshape(v_.shape).float(), v_.float(), atol=0, rtol=0, msg=lambda m, k_=k_: f"{k_}: {m}")
    g = torch.Generator().manual_seed(K * 100 + H + W)
    kp = torch.rand(B, K, 2, generator=g) * torch.tensor([W, H], dtype=torch.float32)
    hs, ws = H // (2 ** ds), W // (2 ** ds)
    batch = {"images": torch.randn(B, 3, H, W, generator=g), "keypoints": kp.reshape(B, 2 * K),
             "heatmaps": Hm.generate_heatmaps(kp, H, W, (hs, ws)), "bbox": torch.tensor([[2.0, 3.0, 3.0 * H, 2.0 * W]]).repeat(B, 1),
             "idxs": torch.arange(B)}
    ref.train()
    model.train()
    want = ref.training_step({k_: (v_.clone() if torch.is_tensor(v_) else v_) for k_, v_ in batch.items()}, 0)
    want["loss"].backward()
    model.configure_optimizers()["optimizer"].zero_grad()
    got = model.training_step({k_: (v_.to(dev) if torch.is_tensor(v_) else v_) for k_, v_ in batch.items()}, 0)
    got["loss"].backward()
    assert float(got["loss"].detach()) == pytest.approx(float(want["loss"].detach()), rel=1e-4)
    assert set(model.logged) == set(ref.logged)
    for name in ("train_heatmap_mse_loss", "train_supervised_loss", "train_heatmap_mse_loss_weighted", "heatmap_mse_weight"):
        assert float(model.logged[name]) == pytest.approx(float(ref.logged[name]), rel=1e-4), name
    # frame-pixel RMSE between labels and the soft-argmax of nearly flat maps: tens of pixels, conditioned like the keypoints (0.1 px)
    assert float(model.logged["train_supervised_rmse"]) == pytest.approx(float(ref.logged["train_supervised_rmse"]), abs=0.2)
    layers = [n_ for n_, _ in ref.head.named_parameters()]
    for n_ in layers:
        a = dict(model.head.named_parameters())[n_].grad.cpu()
        b = dict(ref.head.named_parameters())[n_].grad
        if n_ == [m_ for m_ in layers if m_.endswith(".bias")][-1]:
            continue  # (the last layer's bias: analytically zero under the soft-max - both sides hold rounding noise, 1e-9)
        assert float((a - b).norm()) <= 2e-3 * float(b.norm()), n_
    with torch.no_grad():
        heat_ref = ref(batch["images"])
        heat = model.forward(batch["images"].to(dev)).cpu()
    assert heat.shape == heat_ref.shape == (B, K, hs, ws)
    torch.testing.assert_close(heat, heat_ref, atol=1e-4 * float(heat_ref.max()), rtol=1e-3)
    ref.eval()
    model.eval()
    with torch.no_grad():
        kp_ref, conf_ref = ref.predict_step({k_: (v_.clone() if torch.is_tensor(v_) else v_) for k_, v_ in batch.items()}, 0)
        kp_got, conf_got = model.predict_step({k_: (v_.to(dev) if torch.is_tensor(v_) else v_) for k_, v_ in batch.items()}, 0)
    torch.testing.assert_close(conf_got.cpu(), conf_ref, atol=1e-4, rtol=1e-3)
    torch.testing.assert_close(kp_got.cpu(), kp_ref, atol=0.3, rtol=0)
    # validation_step / test_step (Lightning runs them in eval mode under no_grad): logged names and the well-conditioned values
    for step_name, stage in (("validation_step", "val"), ("test_step", "test")):
        ref.logged, model.logged = {}, {}
        with torch.no_grad():
            getattr(ref, step_name)({k_: (v_.clone() if torch.is_tensor(v_) else v_) for k_, v_ in batch.items()}, 0)
            getattr(model, step_name)({k_: (v_.to(dev) if torch.is_tensor(v_) else v_) for k_, v_ in batch.items()}, 0)
        assert set(model.logged) == set(ref.logged) and f"{stage}_supervised_loss" in ref.logged, (step_name, sorted(ref.logged))
        assert float(model.logged[f"{stage}_supervised_loss"]) == pytest.approx(float(ref.logged[f"{stage}_supervised_loss"]), rel=1e-4)


def test_multiview_supervised_tracker_fp32(dev):
    """5-D (B, V, 3, H, W) labeled batches through the verbatim HeatmapTracker and the product's: heat-maps regrouped to (B, K, h, w) with
    K = keypoints over all views, per-view bbox map in predict_step, return_heatmaps=True"""
    from lightning_pose_amd.losses import LossFactory
    from lightning_pose_amd.models import HeatmapTracker
    from oracle import ref_loader as R

    R.install_stubs()
    T, Fa, Hm = R.load("models.heatmap_tracker"), R.load("losses.factory"), R.load("data.heatmaps")
    Kv, V, HW, B = 3, 2, 64, 2      # num_keypoints is PER VIEW: the network sees B * V images and emits Kv maps for each
    K = Kv * V
    ref = T.HeatmapTracker(num_keypoints=Kv, loss_factory=Fa.LossFactory({"heatmap_mse": {"log_weight": 0.0}}, None), backbone="resnet50",
                           pretrained=False, torch_seed=8, image_size=HW)
    model = HeatmapTracker(num_keypoints=Kv, loss_factory=LossFactory({"heatmap_mse": {"log_weight": 0.0}}, None), backbone="resnet50",
                           pretrained=False, torch_seed=8, device=dev, precision="fp32")
    g = torch.Generator().manual_seed(12)
    kp = torch.rand(B, K, 2, generator=g) * HW
    bbox = torch.tensor([[0.0, 0.0, 64.0, 64.0, 10.0, 20.0, 128.0, 96.0]]).repeat(B, 1)
    batch = {"images": torch.randn(B, V, 3, HW, HW, generator=g), "keypoints": kp.reshape(B, 2 * K),
             "heatmaps": Hm.generate_heatmaps(kp, HW, HW, (HW // 4, HW // 4)), "bbox": bbox, "num_views": torch.full((B,), V),
             "idxs": torch.arange(B)}
    clone = lambda: {k_: (v_.clone() if torch.is_tensor(v_) else v_) for k_, v_ in batch.items()}  # noqa: E731
    ref.train()
    model.train()
    want = ref.training_step(clone(), 0)
    model.configure_optimizers()["optimizer"].zero_grad()
    got = model.training_step({k_: (v_.to(dev) if torch.is_tensor(v_) else v_) for k_, v_ in batch.items()}, 0)
    assert float(got["loss"].detach()) == pytest.approx(float(want["loss"].detach()), rel=1e-4)
    assert set(model.logged) == set(ref.logged)
    ref.eval()
    model.eval()
    with torch.no_grad():
        kp_ref, conf_ref, heat_ref = ref.predict_step(clone(), 0, return_heatmaps=True)
        kp_got, conf_got, heat_got = model.predict_step({k_: (v_.to(dev) if torch.is_tensor(v_) else v_) for k_, v_ in batch.items()}, 0,
                                                        return_heatmaps=True)
    assert heat_got.shape == heat_ref.shape == (B, K, HW // 4, HW // 4)
    torch.testing.assert_close(heat_got.cpu(), heat_ref, atol=1e-4 * float(heat_ref.max()), rtol=1e-3)
    torch.testing.assert_close(conf_got.cpu(), conf_ref, atol=1e-4, rtol=1e-3)
    torch.testing.assert_close(kp_got.cpu(), kp_ref, atol=0.3, rtol=0)   # (frame pixels through the second view's 128 x 96 box at [10, 20])


def test_semisupervised_tracker_fp32(dev):
    """The verbatim SemiSupervisedHeatmapTracker (models/heatmap_tracker.py:208-333, models/base.py:627-701) and the product's: one step with
    temporal + pca_singleview (the verbatim KeypointPCA fit on both sides), a transform handed over as DALI does for a single view
    (2, 3), labeled + unlabeled batches.  Logged names identical; well-conditioned scalars at 1e-4; keypoint-space scalars of this
    random-init net at their conditioning."""
    from lightning_pose_amd.losses import LossFactory
    from lightning_pose_amd.losses import losses as P
    from lightning_pose_amd.models import SemiSupervisedHeatmapTracker
    from oracle import ref_loader as R

    R.install_stubs()
    T, Fa, Hm, L = R.load("models.heatmap_tracker"), R.load("losses.factory"), R.load("data.heatmaps"), R.load("losses.losses")
    K, HW, Bl, S = 4, 64, 2, 3
    g = torch.Generator().manual_seed(77)
    fit = torch.randn(80, 3, generator=g) @ torch.randn(3, 2 * K, generator=g) * 6 + 30
    kpca = R.fit_keypoint_pca("pca_singleview", fit, components_to_keep=0.99)
    temporal = {"log_weight": 1.0, "epsilon": 5.0, "prob_threshold": 0.0}
    # reference side: the factory builds TemporalLoss itself; PCALoss is placed around the in-memory fit (it would need a data module)
    r_unsup = Fa.LossFactory({"temporal": dict(temporal)}, None)
    r_pca = L.PCALoss.__new__(L.PCALoss)
    L.Loss.__init__(r_pca, log_weight=2.0)
    r_pca.device, r_pca.loss_name, r_pca.pca, r_pca.epsilon = "cpu", "pca_singleview", kpca, kpca.parameters["epsilon"]
    r_unsup.loss_instance_dict["pca_singleview"] = r_pca
    ref = T.SemiSupervisedHeatmapTracker(num_keypoints=K, loss_factory=Fa.LossFactory({"heatmap_mse": {"log_weight": 0.0}}, None),
                                         loss_factory_unsupervised=r_unsup, backbone="resnet50", pretrained=False, torch_seed=9, image_size=HW)
    p_unsup = LossFactory({"temporal": dict(temporal),
                           "pca_singleview": {"loss_name": "pca_singleview", "log_weight": 2.0, "components_to_keep": 0.99, "data_arr": fit,
                                              "device": str(dev)}}, None)
    model = SemiSupervisedHeatmapTracker(num_keypoints=K, loss_factory=LossFactory({"heatmap_mse": {"log_weight": 0.0}}, None),
                                         loss_factory_unsupervised=p_unsup, backbone="resnet50", pretrained=False, torch_seed=9, device=dev,
                                         precision="fp32")
    assert isinstance(p_unsup.loss_instance_dict["pca_singleview"], P.PCALoss)
    ref.total_unsupervised_importance = torch.tensor(1.0)
    model.total_unsupervised_importance = torch.tensor(1.0)
    kp = torch.rand(Bl, K, 2, generator=g) * HW
    tf = torch.tensor([[1.0, 0.02, 0.5], [-0.02, 1.0, -0.5]])
    batch = {"labeled": {"images": torch.randn(Bl, 3, HW, HW, generator=g), "keypoints": kp.reshape(Bl, 2 * K),
                         "heatmaps": Hm.generate_heatmaps(kp, HW, HW, (HW // 4, HW // 4)),
                         "bbox": torch.tensor([[0.0, 0.0, 64.0, 64.0]]).repeat(Bl, 1), "idxs": torch.arange(Bl)},
             "unlabeled": {"frames": torch.randn(S, 3, HW, HW, generator=g), "transforms": tf,
                           "bbox": torch.tensor([[0.0, 0.0, 64.0, 64.0]]).repeat(S, 1), "is_multiview": False}}
    clone = lambda d: {k_: ({kk: (vv.clone() if torch.is_tensor(vv) else vv) for kk, vv in v_.items()}) for k_, v_ in d.items()}  # noqa: E731
    ref.train()
    model.train()
    want = ref.training_step(clone(batch), 0)
    model.configure_optimizers()["optimizer"].zero_grad()
    got = model.training_step(_d(batch, dev), 0)
    assert set(model.logged) == set(ref.logged), sorted(set(model.logged) ^ set(ref.logged))
    for name in ("train_heatmap_mse_loss", "train_supervised_loss", "temporal_weight", "pca_singleview_weight", "total_unsupervised_importance"):
        assert float(model.logged[name]) == pytest.approx(float(ref.logged[name]), rel=1e-4), name
    assert float(model.logged["train_temporal_loss"]) == 0.0 == float(ref.logged["train_temporal_loss"])
    # reprojection error of keypoints that sit near the map centre (flat maps) against a PCA of data around 30 px: tens of pixels
    assert float(model.logged["train_pca_singleview_loss"]) == pytest.approx(float(ref.logged["train_pca_singleview_loss"]), rel=1e-2)
    assert float(got["loss"].detach()) == pytest.approx(float(want["loss"].detach()), rel=1e-2)
