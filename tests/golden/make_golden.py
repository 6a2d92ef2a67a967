"""Generate tests/golden/*.npz by running the reference's OWN modules (verbatim, under stubs).

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden.py
Every array stored here is an input fed to, or an output produced by, unmodified reference
code loaded by oracle/ref_loader.py.  The fixtures travel to the GPU box, where
/root/reference does not exist.
"""

from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

try:  # before any stub is installed: transformers probes torchvision with importlib and chokes on the oracle's stand-in module
    import transformers  # noqa: E402
    _VIT_CLASSES = (transformers.ViTModel, transformers.ViTConfig)
except Exception:  # noqa: BLE001
    transformers = None

from oracle import ref_loader as R  # noqa: E402


def _np(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


def save(name, **arrs):
    path = os.path.join(os.environ.get("LP_GOLDEN_OUT", HERE), name + ".npz")  # LP_GOLDEN_OUT: regenerate elsewhere (reproducibility test)
    np.savez_compressed(path, **{k: _np(v) for k, v in arrs.items()})
    print(f"wrote {name}.npz  ({os.path.getsize(path) / 1024:.1f} KiB)")


def peaked_heatmaps(g, b, k, h, w, sharp=1.0):
    """Head-like heatmaps: spatial softmax of smooth blobs + noise (sum to one per map)."""
    ys = torch.arange(h).view(1, 1, h, 1).float()
    xs = torch.arange(w).view(1, 1, 1, w).float()
    cx = torch.rand(b, k, 1, 1, generator=g) * (w - 1)
    cy = torch.rand(b, k, 1, 1, generator=g) * (h - 1)
    logit = -((xs - cx) ** 2 + (ys - cy) ** 2) / (2 * 1.5 ** 2) * sharp * 8 + 0.3 * torch.randn(b, k, h, w, generator=g)
    return torch.softmax(logit.reshape(b, k, -1), -1).reshape(b, k, h, w)


def gen_decode():
    hm = R.load("models.heads.heatmap")
    g = torch.Generator().manual_seed(0)
    out = {}
    for tag, (b, k, h, w) in {"a": (3, 5, 16, 16), "b": (2, 17, 24, 32), "c": (2, 3, 64, 64)}.items():
        x = peaked_heatmaps(g, b, k, h, w)
        out[f"{tag}_in"] = x
        for ds in (1, 2, 3):
            if ds == 3 and h > 32:
                continue
            kp, conf = hm.run_subpixelmaxima(x.clone(), ds, torch.tensor(1000.0))
            out[f"{tag}_kp_ds{ds}"] = kp
            out[f"{tag}_conf_ds{ds}"] = conf
    # flat / low-contrast maps (random-init network regime) and a border peak
    x = torch.softmax(0.01 * torch.randn(2, 4, 16 * 16, generator=g), -1).reshape(2, 4, 16, 16)
    out["flat_in"] = x
    kp, conf = hm.run_subpixelmaxima(x.clone(), 2, torch.tensor(1000.0))
    out["flat_kp_ds2"], out["flat_conf_ds2"] = kp, conf
    x = torch.zeros(1, 4, 12, 12)
    x[0, 0, 0, 0] = 1.0
    x[0, 1, 11, 11] = 1.0
    x[0, 2, 0, 7] = 1.0
    x[0, 3, 5, 11] = 1.0
    out["edge_in"] = x
    kp, conf = hm.run_subpixelmaxima(x.clone(), 2, torch.tensor(1000.0))
    out["edge_kp_ds2"], out["edge_conf_ds2"] = kp, conf
    # upsample alone
    u_in = torch.randn(2, 3, 10, 14, generator=g)
    out["up_in"] = u_in
    out["up_out"] = hm.upsample(u_in.clone())
    save("decode", **out)


def gen_heatmaps():
    H = R.load("data.heatmaps")
    g = torch.Generator().manual_seed(1)
    kp = torch.rand(4, 6, 2, generator=g) * 128
    kp[0, 0] = float("nan")
    kp[1, 2, 0] = -20.0           # far out of bounds
    kp[2, 3] = torch.tensor([129.5, 64.0])  # inside the +-1 heatmap-pixel margin (x*32/128 = 32.4)
    kp[3, 4] = torch.tensor([140.0, 64.0])  # outside margin
    kp[3, 5] = torch.tensor([-3.0, -3.9])   # -0.75,-0.97 on the grid: inside margin
    vis = torch.tensor([[0, 2, 2, 1, 2, 2], [2, 2, 2, 2, 1, 0], [2, 1, 0, 2, 2, 2], [2, 2, 2, 2, 2, 2]])
    out = {"kp": kp, "vis": vis}
    out["hm_novis"] = H.generate_heatmaps(kp.clone(), 128, 128, (32, 32))
    out["hm_vis"] = H.generate_heatmaps(kp.clone(), 128, 128, (32, 32), visibility=vis)
    out["hm_rect"] = H.generate_heatmaps(kp.clone(), 128, 160, (32, 40), sigma=2.0)
    # confidence window
    p = torch.rand(2, 3, 20, 24, generator=g)
    p = p / p.sum(dim=(2, 3), keepdim=True)
    locs = torch.rand(2, 3, 2, generator=g) * torch.tensor([23.0, 19.0])
    locs[0, 0] = torch.tensor([0.2, 0.7])
    locs[1, 2] = torch.tensor([23.0, 19.0])
    out["cw_p"], out["cw_locs"] = p, locs
    out["cw_out"] = H.evaluate_heatmaps_at_location(p, locs)
    save("heatmaps", **out)


def gen_geometry():
    U = R.load("data.utils")
    B = R.load("data.bboxes")
    g = torch.Generator().manual_seed(2)
    kp = torch.rand(6, 8, generator=g) * 100
    th = 0.3
    A = torch.tensor([[1.1 * np.cos(th), -1.1 * np.sin(th), 5.0], [0.9 * np.sin(th), 0.9 * np.cos(th), -3.0]], dtype=torch.float32)
    out = {"kp": kp, "A": A}
    out["undo_single"] = U.undo_affine_transform_batch(kp.clone(), A, False)
    As = A.unsqueeze(0).repeat(6, 1, 1) + 0.05 * torch.randn(6, 2, 3, generator=g)
    out["As"] = As
    out["undo_perframe"] = U.undo_affine_transform_batch(kp.clone(), As, False)
    Av = As[:2]
    out["undo_multiview"] = U.undo_affine_transform_batch(kp.clone(), Av, True)
    out["undo_sentinel"] = U.undo_affine_transform_batch(kp.clone(), torch.tensor([-1.0]), False)
    bbox = torch.tensor([[10.0, 20.0, 200.0, 300.0]]).repeat(6, 1) + torch.rand(6, 4, generator=g)
    out["bbox"] = bbox
    batch = {"frames": torch.zeros(6, 3, 128, 160), "bbox": bbox, "is_multiview": False}
    out["m2f_single"] = B.model_to_frame_batch(batch, kp.clone())
    bbox2 = torch.cat([bbox, bbox * 0.5 + 1.0], dim=1)
    out["bbox2"] = bbox2
    batch = {"frames": torch.zeros(6, 2, 3, 128, 160), "bbox": bbox2, "is_multiview": True}
    out["m2f_multiview"] = B.model_to_frame_batch(batch, kp.clone())
    batch = {"images": torch.zeros(6, 3, 128, 160), "bbox": bbox}
    out["m2f_labeled"] = B.model_to_frame_batch(batch, kp.clone())
    save("geometry", **out)


def gen_losses():
    L = R.load("losses.losses")
    Fa = R.load("losses.factory")
    H = R.load("data.heatmaps")
    g = torch.Generator().manual_seed(3)
    out = {}
    # heatmap losses
    kp = torch.rand(5, 4, 2, generator=g) * 64
    kp[0, 1] = float("nan")
    kp[3, 3] = float("nan")
    targ = H.generate_heatmaps(kp, 64, 64, (16, 16))
    pred = peaked_heatmaps(g, 5, 4, 16, 16, sharp=0.5)
    out["hm_targ"], out["hm_pred"] = targ, pred
    out["heatmap_mse"] = L.HeatmapMSELoss()(targ, pred, stage="train")[0]
    out["heatmap_kl"] = L.HeatmapKLLoss()(targ, pred, stage="train")[0]
    out["heatmap_js"] = L.HeatmapJSLoss()(targ, pred, stage="train")[0]
    # temporal
    kps = torch.cumsum(torch.randn(9, 10, generator=g) * 6, dim=0) + 50
    conf = torch.rand(9, 5, generator=g)
    out["t_kp"], out["t_conf"] = kps, conf
    out["temporal_plain"] = L.TemporalLoss()(kps)[0]
    out["temporal_eps"] = L.TemporalLoss(epsilon=5.0)(kps)[0]
    out["temporal_conf"] = L.TemporalLoss(epsilon=3.0, prob_threshold=0.3)(kps, conf)[0]
    eps_list = [1.0, 2.0, 3.0, 4.0, 5.0]
    out["t_eps_list"] = torch.tensor(eps_list)
    out["temporal_epslist"] = L.TemporalLoss(epsilon=eps_list, prob_threshold=0.3)(kps, conf)[0]
    # rmse
    kt = torch.rand(6, 8, generator=g) * 100
    kt[1, 2:4] = float("nan")
    kt[4, 0:2] = float("nan")
    kpred = kt + torch.randn(6, 8, generator=g)
    kpred = torch.nan_to_num(kpred, nan=7.0)
    out["r_targ"], out["r_pred"] = kt, kpred
    out["rmse"] = L.RegressionRMSELoss()(kt, kpred)[0]
    # pca singleview: low-rank data + noise, NaNs in the fit data
    n, kk = 200, 7
    basis = torch.randn(3, 2 * kk, generator=g)
    data = torch.randn(n, 3, generator=g) @ basis * 10 + 100 + 0.5 * torch.randn(n, 2 * kk, generator=g)
    data_nan = data.clone()
    data_nan[5, 0:2] = float("nan")
    data_nan[17, 6:8] = float("nan")
    cols = [0, 1, 2, 4, 5, 6]
    for tag, ctk in (("sv99", 0.99), ("sv3", 3)):
        kpca = R.fit_keypoint_pca("pca_singleview", data_nan, components_to_keep=ctk,
                                  columns_for_singleview_pca=cols)
        out[f"pca_{tag}_mean"] = kpca.parameters["mean"]
        out[f"pca_{tag}_kept"] = kpca.parameters["kept_eigenvectors"]
        out[f"pca_{tag}_eps"] = kpca.parameters["epsilon"]
        test = data[:16] + 3.0 * torch.randn(16, 2 * kk, generator=g)
        out[f"pca_{tag}_test"] = test
        loss = L.PCALoss.__new__(L.PCALoss)
        L.Loss.__init__(loss, log_weight=0.0)
        loss.device, loss.loss_name, loss.pca = "cpu", "pca_singleview", kpca
        loss.epsilon = kpca.parameters["epsilon"]
        out[f"pca_{tag}_loss"] = loss(test, stage="train")[0]
        out[f"pca_{tag}_err"] = kpca.compute_reprojection_error(kpca._format_data(test))
    out["pca_fit_data"] = data_nan
    out["pca_cols"] = np.array(cols)
    # pca multiview: 2 views x 4 keypoints (8 keypoints total), mirrored matches per view
    mcm = [[0, 1, 2], [4, 5, 6]]
    xyz = torch.randn(150, 3, 3, generator=g) * 20
    P = torch.randn(2, 2, 3, generator=g)
    mv = torch.full((150, 8, 2), 50.0) + 3 * torch.randn(150, 8, 2, generator=g)
    for v in range(2):
        for j in range(3):
            mv[:, mcm[v][j]] = xyz[:, j] @ P[v].T + 0.3 * torch.randn(150, 2, generator=g)
    mv = mv.reshape(150, 16)
    kpca = R.fit_keypoint_pca("pca_multiview", mv, components_to_keep=3, mirrored_column_matches=mcm)
    out["pca_mv_fit_data"] = mv
    out["pca_mv_mcm"] = np.array(mcm)
    out["pca_mv_mean"] = kpca.parameters["mean"]
    out["pca_mv_kept"] = kpca.parameters["kept_eigenvectors"]
    out["pca_mv_eps"] = kpca.parameters["epsilon"]
    test = mv[:10] + 2.0 * torch.randn(10, 16, generator=g)
    out["pca_mv_test"] = test
    loss = L.PCALoss.__new__(L.PCALoss)
    L.Loss.__init__(loss, log_weight=0.0)
    loss.device, loss.loss_name, loss.pca = "cpu", "pca_multiview", kpca
    loss.epsilon = kpca.parameters["epsilon"]
    out["pca_mv_loss"] = loss(test, stage="train")[0]
    # loss factory: weights + anneal
    fac = Fa.LossFactory({"heatmap_mse": {"log_weight": 0.0}}, None)
    for aw in (None, 0.0, 0.5):
        tot, logs = fac(stage="train", anneal_weight=aw, heatmaps_targ=targ, heatmaps_pred=pred)
        out[f"fac_sup_aw{aw}"] = tot
    fac = Fa.LossFactory({"temporal": {"log_weight": 5.0, "epsilon": 3.0, "prob_threshold": 0.3}}, None)
    for aw in (None, 0.0, 0.5, 1.0):
        tot, logs = fac(stage="train", anneal_weight=aw, keypoints_pred=kps, confidences=conf)
        out[f"fac_unsup_aw{aw}"] = tot
    out["fac_log_names"] = np.array([d["name"] for d in logs])
    save("losses", **out)


def gen_callbacks():
    C = R.load("callbacks")

    class M:
        current_epoch = 0
        global_step = 0

    m = M()
    cb = C.AnnealWeight(attr_name="total_unsupervised_importance", init_val=0.0, increase_factor=0.01,
                        final_val=1.0, freeze_until_epoch=3)
    cb.on_train_start(None, m)
    vals = []
    for e in range(120):
        m.current_epoch = e
        cb.on_train_epoch_start(None, m)
        vals.append(float(m.total_unsupervised_importance))
    ub = C.UnfreezeBackbone(unfreeze_epoch=5, initial_ratio=0.1, warm_up_ratio=1.5)
    lrs = []
    head_lr = 1e-3
    for e in range(20):
        if e == 12:
            head_lr *= 0.5  # scheduler milestone during warm-up
        lrs.append(0.0 if ub._warmed_up and False else (ub._get_backbone_lr(None, e, head_lr) if not ub._warmed_up else -1.0))
    save("callbacks", anneal=np.array(vals), unfreeze_lr=np.array(lrs))


def gen_tracker_step():
    """One semi-supervised training step of the reference's own SemiSupervisedHeatmapTracker (ResNet-50,
    random init, torch_seed=7) at 64x64, K=3: logged scalars + a few gradient checksums."""
    T = R.load("models.heatmap_tracker")
    Fa = R.load("losses.factory")
    H = R.load("data.heatmaps")
    g = torch.Generator().manual_seed(4)
    K, S, Bl, HW = 3, 5, 4, 64
    sup = Fa.LossFactory({"heatmap_mse": {"log_weight": 0.0}}, None)
    unsup = Fa.LossFactory({"temporal": {"log_weight": 2.0, "epsilon": 1.0, "prob_threshold": 0.0}}, None)
    model = T.SemiSupervisedHeatmapTracker(num_keypoints=K, loss_factory=sup, loss_factory_unsupervised=unsup,
                                           backbone="resnet50", pretrained=False, torch_seed=7, image_size=HW)
    model.total_unsupervised_importance = torch.tensor(0.5)
    images = torch.randn(Bl, 3, HW, HW, generator=g)
    kp = torch.rand(Bl, 2 * K, generator=g) * HW
    kp[1, 2:4] = float("nan")
    hm = H.generate_heatmaps(kp.reshape(Bl, K, 2), HW, HW, (HW // 4, HW // 4))
    bbox_l = torch.tensor([[3.0, 5.0, 128.0, 96.0]]).repeat(Bl, 1)
    frames = torch.randn(S, 3, HW, HW, generator=g)
    th = 0.1
    A = torch.tensor([[np.cos(th), -np.sin(th), 2.0], [np.sin(th), np.cos(th), -1.0]], dtype=torch.float32)
    bbox_u = torch.tensor([[0.0, 0.0, 64.0, 64.0]]).repeat(S, 1)
    batch = {
        "labeled": {"images": images, "keypoints": kp.clone(), "heatmaps": hm, "bbox": bbox_l,
                    "idxs": torch.arange(Bl)},
        "unlabeled": {"frames": frames, "transforms": A, "bbox": bbox_u, "is_multiview": False},
    }
    model.train()
    out = model.training_step(batch, 0)
    out["loss"].backward()
    logged = {k: float(v) for k, v in model.logged.items()}
    with torch.no_grad():
        heat = model.forward(images)
    arrs = dict(images=images, keypoints=kp, heatmaps=hm, bbox_l=bbox_l, frames=frames, A=A, bbox_u=bbox_u,
                log_names=np.array(list(logged.keys())), log_values=np.array(list(logged.values())),
                loss=out["loss"],
                g_head_last_w=model.head.upsampling_layers[2].weight.grad,
                g_head_first_b=model.head.upsampling_layers[1].bias.grad,
                g_conv1_norm=model.backbone[0].weight.grad.norm(),
                g_l4_last_norm=model.backbone[7][2].conv3.weight.grad.norm(),
                w_conv1_sum=model.backbone[0].weight.detach().double().sum(),
                w_head1_sum=model.head.upsampling_layers[1].weight.detach().double().sum(),
                heat_after_step_train_mode=heat)
    save("tracker_step", **arrs)


def _stub_prediction_imports():
    """utils/predictions.py also imports cv2 / moviepy (video rendering, not on the path): empty stand-ins."""
    import types
    for n in ("cv2", "moviepy"):
        if n not in sys.modules:
            try:
                __import__(n)
            except Exception:  # noqa: BLE001
                sys.modules[n] = types.ModuleType(n)
    if not hasattr(sys.modules["moviepy"], "VideoFileClip"):
        sys.modules["moviepy"].VideoFileClip = object


def gen_predictions():
    """PredictionHandler (utils/predictions.py:41-330) on seeded per-batch outputs: single-view video with a padded last
    sequence, multiview video, labeled dataset with split column, and the context-model shift."""
    _stub_prediction_imports()
    P = R.load("utils.predictions")
    g = torch.Generator().manual_seed(11)
    K = 5
    names = [f"kp{i}" for i in range(K)]

    class H(P.PredictionHandler):  # the reference counts frames by opening the video with OpenCV; pin the count instead
        n_frames = 0

        @property
        def frame_count(self):
            return self.n_frames

    def batches(n_batches, bsz, cols):
        return [(torch.rand(bsz, 2 * cols, generator=g) * 300, torch.rand(bsz, cols, generator=g)) for _ in range(n_batches)]

    arrs = {}
    # (1) single-view video: 5 sequences of 16, 70 real frames
    cfg = R._wrap({"data": {"keypoint_names": names}, "model": {"model_type": "heatmap"}})
    pr = batches(5, 16, K)
    h = H(cfg, video_file="clip.mp4"); h.n_frames = 70
    df = h(preds=pr)
    arrs.update(v1_kp=torch.vstack([p[0] for p in pr]), v1_conf=torch.vstack([p[1] for p in pr]), v1_table=df.to_numpy(),
                v1_columns=np.array(["|".join(c) for c in df.columns]))
    # (2) multiview video, 2 views
    cfg_mv = R._wrap({"data": {"keypoint_names": names, "view_names": ["top", "bot"]}, "model": {"model_type": "heatmap_multiview_transformer"}})
    pr = batches(3, 8, 2 * K)
    h = H(cfg_mv, video_file="clip_top.mp4"); h.n_frames = 21
    d = h(preds=pr, is_multiview_video=True)
    arrs.update(v2_kp=torch.vstack([p[0] for p in pr]), v2_conf=torch.vstack([p[1] for p in pr]), v2_top=d["top"].to_numpy(),
                v2_bot=d["bot"].to_numpy())
    # (3) labeled dataset with split indices
    class Sub:
        def __init__(self, idx):
            self.indices = idx

    class DS:
        do_context = False
        image_names = [f"labeled-data/vid/img{i:03d}.png" for i in range(12)]

        def __len__(self):
            return 12

    class DM:
        dataset = DS()
        train_dataset, val_dataset, test_dataset = Sub([0, 2, 4, 6, 8, 10]), Sub([1, 5]), Sub([3, 7])

    pr = batches(3, 4, K)
    df = P.PredictionHandler(cfg, data_module=DM())(preds=pr)
    arrs.update(v3_kp=torch.vstack([p[0] for p in pr]), v3_conf=torch.vstack([p[1] for p in pr]),
                v3_table=df.drop(columns="set", level=0).to_numpy().astype(np.float64),
                v3_set=np.array(df[("set", "", "")].tolist()), v3_index=np.array(list(df.index)))
    # (4) context shift (pure tensor logic; the context models themselves are out of scope)
    cfg_ctx = R._wrap({"data": {"keypoint_names": names}, "model": {"model_type": "heatmap_mhcrnn"}})
    pr = batches(2, 16, K)
    h = H(cfg_ctx, video_file="clip.mp4"); h.n_frames = 30
    df = h(preds=pr)
    arrs.update(v4_kp=torch.vstack([p[0] for p in pr]), v4_conf=torch.vstack([p[1] for p in pr]), v4_table=df.to_numpy())
    h.n_frames = 32  # exactly as many frames as rows
    arrs.update(v4b_table=h(preds=pr).to_numpy())
    save("predictions", **arrs)


def _load_verbatim_datasets():
    """data/datasets.py, executed unchanged under stand-ins for the image libraries it imports at module level (imgaug, cv2,
    torchvision.transforms, the camera-group helper): only its keypoint / visibility / heat-map code is exercised."""
    import importlib.util
    import types

    R.install_stubs()

    class _Aug:
        def add(self, *a, **k):
            pass

    def mod(name, **attrs):
        m = sys.modules.get(name) or types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    for n in ("cv2",):
        try:
            __import__(n)
        except Exception:  # noqa: BLE001
            mod(n)
    mod("imgaug")
    mod("imgaug.augmenters", Sequential=_Aug, Resize=lambda *a, **k: None)
    mod("imgaug.augmenters.size")
    sys.modules["imgaug"].augmenters = sys.modules["imgaug.augmenters"]
    sys.modules["imgaug.augmenters"].size = sys.modules["imgaug.augmenters.size"]
    tv = sys.modules["torchvision"]
    tv.transforms = mod("torchvision.transforms", ToTensor=lambda: None, Normalize=lambda **k: None, Compose=lambda l: None)
    mod("lightning_pose.data.cameras", CameraGroup=object)
    data_pkg = sys.modules["lightning_pose.data"]
    data_pkg._IMAGENET_MEAN, data_pkg._IMAGENET_STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    spec = importlib.util.spec_from_file_location("lightning_pose.data._datasets_verbatim",
                                                  os.path.join(R.REFERENCE_ROOT, "lightning_pose", "data", "datasets.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m, _Aug


def gen_labeled_targets():
    """HeatmapDataset (data/datasets.py:380-550) on the bundled mirror-mouse labels: the verbatim __init__ parses the CSV and
    synthesises visibility (:465-472); the verbatim compute_heatmap (:496-523) turns model-space keypoints into targets, with
    out-of-frame points set to NaN.  Keypoints reach model space by imgaug's Resize projection x / from_w * to_w (imgaug is
    not installed: that one line is applied here), after an augmentation affine that pushes some points out of the frame."""
    D, Aug = _load_verbatim_datasets()
    root = os.path.join(R.REFERENCE_ROOT, "data", "mirror-mouse-example")
    H = W = 256
    SRC_H, SRC_W = 406, 396  # size of the bundled frames (labeled-data/*.png)
    out = {}
    for tag, uniform in (("u0", False), ("u1", True)):
        ds = D.HeatmapDataset(root_directory=root, csv_path="CollectedData.csv", image_resize_height=H, image_resize_width=W,
                              imgaug_transform=Aug(), downsample_factor=2, uniform_heatmaps=uniform)
        idxs = [0, 3, 7, 11, 20, 33, 41, 57]
        kp_src = ds.keypoints[idxs].clone()                       # (8, 17, 2) source px, NaN where unlabeled
        vis = ds.visibility[idxs].clone()
        g = torch.Generator().manual_seed(3)
        th = (torch.rand(len(idxs), generator=g) - 0.5) * 0.8
        sc = 0.8 + 0.6 * torch.rand(len(idxs), generator=g)
        A = torch.zeros(len(idxs), 2, 3)
        A[:, 0, 0], A[:, 0, 1], A[:, 1, 0], A[:, 1, 1] = sc * torch.cos(th), -sc * torch.sin(th), sc * torch.sin(th), sc * torch.cos(th)
        A[:, :, 2] = (torch.rand(len(idxs), 2, generator=g) - 0.5) * 160
        A[0] = torch.tensor([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])
        x = A[:, None, 0, 0] * kp_src[..., 0] + A[:, None, 0, 1] * kp_src[..., 1] + A[:, None, 0, 2]
        y = A[:, None, 1, 0] * kp_src[..., 0] + A[:, None, 1, 1] * kp_src[..., 1] + A[:, None, 1, 2]
        kp_model = torch.stack([x / SRC_W * W, y / SRC_H * H], -1)
        hms, kps = [], []
        for i in range(len(idxs)):
            ex = {"keypoints": kp_model[i].reshape(-1).clone(), "visibility": vis[i]}
            hms.append(ds.compute_heatmap(ex))
            kps.append(ex["keypoints"].reshape(-1, 2))  # compute_heatmap wrote the NaNs through the view
        out.update({f"{tag}_vis": vis, f"{tag}_heatmaps": torch.stack(hms), f"{tag}_kp_model_nan": torch.stack(kps)})
        if tag == "u0":
            out.update(kp_src=kp_src, affine=A, src_hw=torch.tensor([[SRC_H, SRC_W]] * len(idxs), dtype=torch.float32))
    # hflip with the left/right swap (:288-293, :364-366): restated sequence applied to the verbatim-parsed labels
    save("labeled_targets", **out)


def gen_temporal_heatmap():
    """TemporalHeatmapLoss (losses/losses.py:706-869), verbatim: mse / kl, scalar and per-keypoint epsilon, confidence threshold."""
    L = R.load("losses.losses")
    g = torch.Generator().manual_seed(21)
    S, K, h, w = 6, 4, 16, 16
    hm = peaked_heatmaps(g, S, K, h, w, sharp=0.6)
    hm[3, 1] = hm[2, 1]                      # an identical pair: distance exactly 0
    conf = torch.rand(S, K, generator=g)
    out = dict(hm=hm, conf=conf)
    cases = {"mse_plain": ("temporal_heatmap_mse", 0.0, 0.0), "kl_plain": ("temporal_heatmap_kl", 0.0, 0.0),
             "mse_thr": ("temporal_heatmap_mse", 0.0, 0.4), "kl_thr_eps": ("temporal_heatmap_kl", 0.5, 0.4),
             "mse_eps_list": ("temporal_heatmap_mse", [0.0, 2e-5, 1e-4, 1.0], 0.2)}
    for tag, (name, eps, thr) in cases.items():
        loss = L.TemporalHeatmapLoss(loss_name=name, epsilon=eps, prob_threshold=thr, log_weight=1.5)
        p = hm.clone().requires_grad_(True)
        val, logs = loss(heatmaps_pred=p, confidences=conf, stage="train")
        (0.7 * val).backward()
        out[f"{tag}_loss"], out[f"{tag}_grad"] = val, p.grad
        out[f"{tag}_elementwise"] = loss.compute_loss(predictions=hm)
    out["log_names"] = np.array([d["name"] for d in logs])
    out["weight"] = logs[1]["value"]
    save("temporal_heatmap", **out)


def _train_head(model, inp, gen_hm, steps, lr):
    """Adam on the reference head alone, over the cached trunk features of the step's own frames (the trunk stays at its seeded
    initialisation, BatchNorm in training mode per batch, as in the measured step): after it the heat-maps are smooth peaks."""
    cfg, batch = inp["cfg"], inp["batch"]
    K, HW = cfg["K"], cfg["HW"]
    h = HW // 4
    lab = batch["labeled"] if "labeled" in batch else batch
    sets = [(lab["images"].reshape(-1, 3, HW, HW), lab["heatmaps"].reshape(-1, K, h, h))]
    if "unlabeled" in batch:
        sets.append((batch["unlabeled"]["frames"].reshape(-1, 3, HW, HW), gen_hm(inp["unl_centres"].clone(), HW, HW, (h, h))))
    model.train()
    with torch.no_grad():
        feats = [(model.backbone(x), t) for x, t in sets]
    opt = torch.optim.Adam(model.head.parameters(), lr=lr)
    for _ in range(steps):
        opt.zero_grad()
        loss = 0.0
        for f, t in feats:
            p = model.head(f)
            keep = t.flatten(2).sum(-1) > 0
            loss = loss + ((p - t) ** 2)[keep].mean() * h * h
        loss.backward()
        opt.step()
    model.zero_grad()
    return float(loss)


def gen_step_parity(names=None):
    """End-to-end parity steps on PEAKED heat-maps at BASELINE.json's configs (tests/golden/step_inputs.py): one training step of the
    reference's own (Semi)SupervisedHeatmapTracker - verbatim modules, fp32, torch CPU - after its head was trained on the step's frames
    (_train_head; the trained head weights are stored so the product starts from the same model).  Stored: every logged scalar, the loss
    inputs the losses saw (predicted keypoints in frame and model coordinates, confidences), per-map peak statistics of the heat-maps,
    and parameter gradients (head + stem in full, a norm per convolution / BatchNorm)."""
    from oracle import restated as O
    from tests.golden.step_inputs import (HEAD_TRAIN_LR, HEAD_TRAIN_STEPS, PCA_LOG_WEIGHT, RESIDUAL_GAIN, STEP_CONFIGS, TEMPORAL, TORCH_SEED,
                                          make_step_inputs, seeded_backbone_weights)

    T = R.load("models.heatmap_tracker")
    Fa = R.load("losses.factory")
    L = R.load("losses.losses")
    H = R.load("data.heatmaps")
    for name in (names or list(STEP_CONFIGS)):
        inp = make_step_inputs(name, H.generate_heatmaps)
        cfg, batch = inp["cfg"], inp["batch"]
        K, V, HW = cfg["K"], cfg["V"], cfg["HW"]
        backbone = cfg.get("backbone", "resnet50")
        if backbone != "resnet50":
            # models/backbones/vit.py:26-27 calls ViTModel.from_pretrained("facebook/dino-vits16"): no network here, so the same architecture
            # is constructed from its config (facebook/dino-vits16: hidden 384, 12 layers, 6 heads, MLP 1536, patch 16, 224-px position
            # table) and its weights replaced by the seeded draw both sides share (step_inputs.seeded_backbone_weights)
            def _from_config(model_name, add_pooling_layer=False, **kw):
                assert model_name == "facebook/dino-vits16", model_name
                c = transformers.ViTConfig(hidden_size=384, num_hidden_layers=12, num_attention_heads=6, intermediate_size=1536, patch_size=16,
                                           image_size=224, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
                return transformers.ViTModel(c, add_pooling_layer=add_pooling_layer)
            transformers.ViTModel.from_pretrained = staticmethod(_from_config)
        sup = Fa.LossFactory({"heatmap_mse": {"log_weight": 0.0}}, None)
        semi = cfg["S"] > 0
        if semi:
            unsup = Fa.LossFactory({"temporal": dict(TEMPORAL)}, None)
            ptype = "pca_multiview" if V > 1 else "pca_singleview"
            kpca = R.fit_keypoint_pca(ptype, inp["pca_fit"], components_to_keep=3 if V > 1 else 0.99, mirrored_column_matches=inp["mcm"],
                                      columns_for_singleview_pca=inp["cols"])
            loss = L.PCALoss.__new__(L.PCALoss)
            L.Loss.__init__(loss, log_weight=PCA_LOG_WEIGHT)
            loss.device, loss.loss_name, loss.pca = "cpu", ptype, kpca
            loss.epsilon = kpca.parameters["epsilon"]
            unsup.loss_instance_dict[ptype] = loss
            model = T.SemiSupervisedHeatmapTracker(num_keypoints=K, loss_factory=sup, loss_factory_unsupervised=unsup, backbone=backbone,
                                                   pretrained=False, torch_seed=TORCH_SEED, image_size=HW)
            model.total_unsupervised_importance = torch.tensor(1.0)
        else:
            model = T.HeatmapTracker(num_keypoints=K, loss_factory=sup, backbone="resnet50", pretrained=False, torch_seed=TORCH_SEED,
                                     image_size=HW)
        with torch.no_grad():
            for n_, p_ in model.named_parameters():
                if n_.endswith("bn3.weight"):
                    p_.fill_(RESIDUAL_GAIN)
            if backbone != "resnet50":
                sd_ = model.state_dict()
                new = seeded_backbone_weights(sd_)
                assert new, "no backbone tensors found"
                sd_.update(new)
                model.load_state_dict(sd_)
                arrs_names = np.array(sorted(new))
        fit_loss = _train_head(model, inp, H.generate_heatmaps, cfg.get("head_steps", HEAD_TRAIN_STEPS), HEAD_TRAIN_LR)
        seen = {}
        for meth in ("get_loss_inputs_labeled", "get_loss_inputs_unlabeled"):
            if hasattr(model, meth):
                orig = getattr(model, meth)

                def wrapped(batch_dict, _orig=orig, _m=meth):
                    d = _orig(batch_dict)
                    seen[_m] = {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in d.items()}
                    return d
                setattr(model, meth, wrapped)
        model.train()
        arrs = {"head/" + n_: p_.detach().clone() for n_, p_ in model.head.named_parameters()}
        # what the bf16-mixed POLICY itself costs on this model: the same weights through oracle.restated.forward_bf16_policy (torch
        # CPU, rounding to bf16 where the product does) -> keypoints / confidences next to the fp32 ones below
        lab_b = batch["labeled"] if semi else batch
        if backbone == "resnet50":
            orc = O.OracleTracker(K, 2, torch_seed=0)
            orc.load_state_dict({k_: v_ for k_, v_ in model.state_dict().items()}, strict=True)
            orc.train()
        else:
            arrs["backbone_names"] = arrs_names  # the tensors both sides overwrote (name check in the test)
        with torch.no_grad():
            for tag, bd, key in ((("lab", lab_b, "images"),) + ((("unl", batch["unlabeled"], "frames"),) if semi else ())) if backbone == "resnet50" else ():
                x_ = bd[key]
                h_ = O.forward_bf16_policy(orc, x_.reshape(-1, 3, HW, HW))
                h_ = h_.reshape(x_.shape[0], -1, h_.shape[-2], h_.shape[-1])
                kp_, conf_ = O.soft_argmax(h_, 2, 1000.0)
                if tag == "unl":
                    arrs["bf16ref_unl_keypoints_pred_augmented"] = kp_.clone()
                    kp_ = O.undo_affine(kp_, bd["transforms"], bool(bd.get("is_multiview", False)))
                arrs[f"bf16ref_{tag}_keypoints_pred"] = O.model_to_frame(kp_, HW, HW, bd["bbox"], V)
                arrs[f"bf16ref_{tag}_confidences"] = conf_
                arrs[f"bf16ref_{tag}_heat_max"] = h_.flatten(2).max(-1).values
        outputs_only = bool(cfg.get("outputs_only"))
        if outputs_only:   # (c4full: the forward quantities only - see step_inputs.STEP_CONFIGS)
            with torch.no_grad():
                out = model.training_step(batch, 0)
        else:
            out = model.training_step(batch, 0)
            out["loss"].backward()
        logged = {k: float(v) for k, v in model.logged.items()}
        arrs.update(log_names=np.array(list(logged)), log_values=np.array(list(logged.values())), loss=out["loss"].detach(),
                    head_fit_loss=np.float32(fit_loss))
        for meth, tag in (("get_loss_inputs_labeled", "lab"), ("get_loss_inputs_unlabeled", "unl")):
            if meth in seen:
                d = seen[meth]
                for k in ("keypoints_pred", "keypoints_pred_augmented", "confidences", "keypoints_targ"):
                    if k in d:
                        arrs[f"{tag}_{k}"] = d[k]
                hm = d["heatmaps_pred"]
                flat = hm.reshape(hm.shape[0], hm.shape[1], -1)
                arrs[f"{tag}_heat_max"], arrs[f"{tag}_heat_argmax"] = flat.max(-1).values, flat.argmax(-1)
                if name == "s64":
                    arrs[f"{tag}_heat"] = hm
        if semi:
            arrs.update(pca_mean=kpca.parameters["mean"], pca_kept=kpca.parameters["kept_eigenvectors"], pca_eps=kpca.parameters["epsilon"])
        if not outputs_only:
            sd_grads = {n_: p_.grad for n_, p_ in model.named_parameters() if p_.grad is not None}
            for n_, gr in sd_grads.items():
                if n_.startswith("head.") or n_ == "backbone.0.weight":
                    arrs["grad/" + n_] = gr
            names_sorted = sorted(sd_grads)
            arrs["grad_names"] = np.array(names_sorted)
            arrs["grad_norms"] = np.array([float(sd_grads[n_].norm()) for n_ in names_sorted])
        save(f"step_{name}", **arrs)
        print("   ", {k: round(v, 6) for k, v in logged.items()}, "head fit", round(fit_loss, 6))


if __name__ == "__main__":
    torch.set_num_threads(8)
    GENS = {"decode": gen_decode, "heatmaps": gen_heatmaps, "geometry": gen_geometry, "losses": gen_losses,
            "callbacks": gen_callbacks, "tracker_step": gen_tracker_step, "predictions": gen_predictions, "labeled_targets": gen_labeled_targets, "temporal_heatmap": gen_temporal_heatmap, "step_parity": gen_step_parity}
    for name in (sys.argv[1:] or list(GENS)):  # `make_golden.py predictions` regenerates one fixture only; `step_parity:c4full` one step config
        if ":" in name:
            gname, sub = name.split(":", 1)
            GENS[gname](sub.split(","))
        else:
            GENS[name]()
