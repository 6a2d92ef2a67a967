#!/usr/bin/env python
"""bench.py - training frames/s of the MI355X-native heatmap-tracker step (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python bench.py --gpus N --steps K --warmup W          (no WORLD_SIZE in the environment: spawns its own N ranks, see self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one full optimisation step of ResNet-50 SemiSupervisedHeatmapTracker at 384x384, K=17, on synthetic
data: forward + backward over 64 labeled and 128 unlabeled frames per GPU (BASELINE config C2/C3), heatmap_mse +
temporal + pca_singleview + unimodal_mse losses, gradient all-reduce over RCCL when N > 1, fused Adam.  Inputs are
resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline      MFMA roofline of the convolution kernels (the dominant cost): algorithmic FLOPs / HIP-event time per
                launch, measured on the launch stream inside the timed steps; by_kernel gives per-symbol averages that
                the committed rocprofv3 summary (profiles/) must reproduce.
  cpu_baseline  the CPU oracle (oracle/restated.py: the reference's arithmetic restated in torch fp32, pinned against
                the verbatim reference modules) timed on this host's cores for a bounded sample of the same workload.
"""

from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import _lp_bootstrap  # noqa: E402,F401

MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0
HBM_SUSTAINED_GBS = 6300.0      # what a streaming kernel reaches on this part (same guide): the bandwidth leg of a launch's own roof
VALU_FP32_PEAK_TFLOPS = 157.0   # fp32 vector peak, same guide
TRAIN_GFLOP_PER_FRAME = {384: 72.4, 256: 32.2}  # 3 x 2 x (trunk + head) MACs
VIT_S_TRAIN_GFLOP_PER_FRAME = {384: 93.1, 256: 36.9}  # SURVEY.md section 8(d), ViT-S/16


def _sync(dev) -> None:
    if dev.type == "cuda":
        torch.cuda.synchronize()


def synth_batch(dev, rank: int, size: int, n_lab: int, n_unlab: int, K: int):
    """Seeded synthetic labeled + unlabeled batch of SURVEY.md section 8(d), generated on the device."""
    from lightning_pose_amd import ops

    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    images = torch.randn(n_lab, 3, size, size, generator=g).to(dev)
    kp = torch.rand(n_lab, K, 2, generator=g) * size
    nan = torch.rand(n_lab, K, generator=g) < 0.088          # NaN rate of the bundled mirror-mouse labels
    kp[nan] = float("nan")
    vis = torch.where(nan, torch.ones_like(nan, dtype=torch.int32), torch.full_like(nan, 2, dtype=torch.int32))
    kp_d = kp.to(dev)
    heat = ops.generate_heatmaps(kp_d, size, size, (size // 4, size // 4), 1.25, vis.to(dev))
    bbox = torch.tensor([[0.0, 0.0, float(size), float(size)]])
    # temporally coherent unlabeled window: K blobs following a random walk over low-amplitude noise
    centres = torch.cumsum(torch.randn(n_unlab, K, 2, generator=g) * 4.0, dim=0) + torch.rand(1, K, 2, generator=g) * size
    centres = centres.clamp(8, size - 8).to(dev)
    ys = torch.arange(size, device=dev).view(1, 1, size, 1).float()
    xs = torch.arange(size, device=dev).view(1, 1, 1, size).float()
    frames = torch.randn(n_unlab, 3, size, size, generator=g).to(dev) * 0.5
    for k0 in range(K):  # accumulate blobs without materialising (S, K, H, W)
        blob = 3.0 * torch.exp(-((xs - centres[:, k0, 0].view(-1, 1, 1, 1)) ** 2 + (ys - centres[:, k0, 1].view(-1, 1, 1, 1)) ** 2) / 72.0)
        frames += blob
    th = math.radians(float(torch.rand(1, generator=g)) * 20 - 10)
    sc = 0.8 + 0.4 * float(torch.rand(1, generator=g))
    c = size / 2
    a = torch.tensor([[sc * math.cos(th), -sc * math.sin(th), 0.0], [sc * math.sin(th), sc * math.cos(th), 0.0]])
    a[:, 2] = torch.tensor([c, c]) - a[:, :2] @ torch.tensor([c, c])
    return {
        "labeled": {"images": images, "keypoints": kp_d.reshape(n_lab, 2 * K), "heatmaps": heat,
                    "bbox": bbox.repeat(n_lab, 1).to(dev), "idxs": torch.arange(n_lab)},
        "unlabeled": {"frames": frames, "transforms": a.to(dev), "bbox": bbox.repeat(n_unlab, 1).to(dev), "is_multiview": False},
    }


def synth_multiview_batch(dev, rank: int, size: int, n_lab: int, n_unlab: int, K: int, V: int):
    """Config C5: V views per frame, keypoint axis K*V (views of one frame stay on one GPU).  Every view sees
    the same 3-D random walk through its own affine "camera", so the multiview PCA loss has a low-dimensional structure to find."""
    from lightning_pose_amd import ops

    g = torch.Generator(device="cpu").manual_seed(4321 + rank)
    cams = torch.eye(2).repeat(V, 1, 1) + 0.25 * torch.randn(V, 2, 2, generator=g)        # per-view linear map of the common track
    shift = torch.rand(V, 1, 2, generator=g) * size * 0.2

    def tracks(n):
        base = torch.cumsum(torch.randn(n, K, 2, generator=g) * 3.0, dim=0) + size * (0.3 + 0.4 * torch.rand(1, K, 2, generator=g))
        per_view = torch.einsum("vij,nkj->nvki", cams, base - size / 2) + size / 2 + shift.unsqueeze(0)
        return per_view.clamp(6, size - 6)                                                 # (n, V, K, 2)

    def render(centres):                                                                   # (n, V, K, 2) -> (n, V, 3, size, size)
        n = centres.shape[0]
        c = centres.reshape(n * V, K, 2).to(dev)
        ys = torch.arange(size, device=dev).view(1, 1, size, 1).float()
        xs = torch.arange(size, device=dev).view(1, 1, 1, size).float()
        img = torch.randn(n * V, 3, size, size, generator=g).to(dev) * 0.5
        for k0 in range(K):
            img += 3.0 * torch.exp(-((xs - c[:, k0, 0].view(-1, 1, 1, 1)) ** 2 + (ys - c[:, k0, 1].view(-1, 1, 1, 1)) ** 2) / 72.0)
        return img.reshape(n, V, 3, size, size)

    lab = tracks(n_lab)
    kp = lab.reshape(n_lab, V * K, 2).clone()
    nan = torch.rand(n_lab, V * K, generator=g) < 0.088
    kp[nan] = float("nan")
    vis = torch.where(nan, torch.ones_like(nan, dtype=torch.int32), torch.full_like(nan, 2, dtype=torch.int32))
    heat = ops.generate_heatmaps(kp.to(dev), size, size, (size // 4, size // 4), 1.25, vis.to(dev))
    unl = tracks(n_unlab)
    bbox = torch.tensor([[0.0, 0.0, float(size), float(size)] * V])
    tfs = []
    for _ in range(V):  # one augmentation matrix per view (data/video/dali.py:158-164)
        th = math.radians(float(torch.rand(1, generator=g)) * 20 - 10)
        sc = 0.8 + 0.4 * float(torch.rand(1, generator=g))
        c = size / 2
        a = torch.tensor([[sc * math.cos(th), -sc * math.sin(th), 0.0], [sc * math.sin(th), sc * math.cos(th), 0.0]])
        a[:, 2] = torch.tensor([c, c]) - a[:, :2] @ torch.tensor([c, c])
        tfs.append(a)
    return {
        "labeled": {"images": render(lab), "keypoints": kp.reshape(n_lab, 2 * K * V).to(dev), "heatmaps": heat,
                    "bbox": bbox.repeat(n_lab, 1).to(dev), "num_views": torch.full((n_lab,), V), "idxs": torch.arange(n_lab)},
        "unlabeled": {"frames": render(unl), "transforms": torch.stack(tfs).to(dev), "bbox": bbox.repeat(n_unlab, 1).to(dev),
                      "is_multiview": True},
    }


def multiview_pca_training_array(K: int, V: int, size: int) -> torch.Tensor:
    """Stand-in for the labelled multiview keypoints the PCA is fitted on: (N, 2*K*V), rows = frames, view-major keypoint axis; the views
    are affine images of a common 3-D point cloud, so 3 components explain them (the reference keeps 3, losses/losses.py:505-508)."""
    g = torch.Generator().manual_seed(98)
    pts = torch.randn(400, K, 3, generator=g) * (size / 8)
    proj = torch.randn(V, 2, 3, generator=g)
    obs = torch.einsum("vij,nkj->nvki", proj, pts) + size / 2 + torch.randn(400, V, K, 2, generator=g)
    return obs.reshape(400, V * K * 2)


def pca_training_array(K: int, size: int) -> torch.Tensor:
    """Synthetic stand-in for the labelled keypoints the PCA is fitted on (no dataset files on the GPU box)."""
    g = torch.Generator().manual_seed(99)
    basis = torch.randn(6, 2 * K, generator=g)
    data = torch.randn(300, 6, generator=g) @ basis * (size / 16) + size / 2 + torch.randn(300, 2 * K, generator=g)
    return data


def build_model(dev, K: int, size: int, torch_seed: int = 0, backbone: str = "resnet50", views: int = 1, precision: str = "bf16-mixed"):
    from lightning_pose_amd.losses import LossFactory
    from lightning_pose_amd.models import SemiSupervisedHeatmapTracker

    cols = [k for k in range(K) if k not in (7, 15, 16)] if K == 17 else list(range(K))
    sup = LossFactory({"heatmap_mse": {"log_weight": 0.0}}, None)
    if views > 1:  # config C5: pca_multiview + temporal (BASELINE.json configs[4]); keypoint k of view v is column v*K + k
        mcm = [[v * K + k for k in range(K)] for v in range(views)]
        unsup = LossFactory({
            "temporal": {"log_weight": 5.0, "epsilon": 20.0, "prob_threshold": 0.05},
            "pca_multiview": {"loss_name": "pca_multiview", "log_weight": 5.0, "components_to_keep": 3, "mirrored_column_matches": mcm,
                              "data_arr": multiview_pca_training_array(K, views, size), "device": str(dev)},
        }, None)
        return SemiSupervisedHeatmapTracker(num_keypoints=K, loss_factory=sup, loss_factory_unsupervised=unsup, backbone=backbone,
                                            downsample_factor=2, pretrained=False, torch_seed=torch_seed, device=dev, precision=precision)
    unsup = LossFactory({
        "temporal": {"log_weight": 5.0, "epsilon": 20.0, "prob_threshold": 0.05},
        "pca_singleview": {"loss_name": "pca_singleview", "log_weight": 5.0, "components_to_keep": 0.99,
                           "columns_for_singleview_pca": cols, "data_arr": pca_training_array(K, size), "device": str(dev)},
        "unimodal_mse": {"log_weight": 5.0, "prob_threshold": 0.05, "original_image_height": size, "original_image_width": size},
    }, None)
    return SemiSupervisedHeatmapTracker(num_keypoints=K, loss_factory=sup, loss_factory_unsupervised=unsup, backbone=backbone,
                                        downsample_factor=2, pretrained=False, torch_seed=torch_seed, device=dev, precision=precision)


def pmc_traffic():
    """Average HBM bytes per convolution launch from the committed rocprofv3 PMC passes of this same command (FETCH_SIZE and WRITE_SIZE in
    separate runs, corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes; profiles/r06_final.sh -> profiles/r06_pmc_traffic.json
    via profiles/summarize_pmc.py).  The counters need rocprofv3 around the process, so this is the committed measurement of the same
    workload, not a live one (``algorithmic_bytes_per_launch`` next to it IS computed live) - and it is only reported when it was taken
    on THESE kernels: the file records a digest of the convolution sources, and a file whose digest differs from the tree's is refused
    (``traffic`` = null, ``traffic_source`` says why).  -> (bytes | None, source note)"""
    import hashlib
    h = hashlib.sha256()
    try:
        for name in ("conv.hip", "conv_pipe.h", "conv_res2d.h", "conv_stem_wgrad.h", "lp_common.h"):   # (profiles/summarize_pmc.py: KERNEL_SOURCES)
            with open(os.path.join(ROOT, "lightning-pose_amd", "csrc", name), "rb") as fh:
                h.update(fh.read())
    except OSError:
        return None, "kernel sources not found"
    name = "r06_pmc_traffic.json"
    try:
        with open(os.path.join(ROOT, "profiles", name)) as fh:
            rec = json.load(fh)
        if rec.get("kernels_sha256") != h.hexdigest():
            return None, f"{name} refused: taken on other kernel sources (digest {str(rec.get('kernels_sha256'))[:12]} != {h.hexdigest()[:12]})"
        return round(rec["conv_hbm_bytes_per_launch"]), name
    except (OSError, KeyError, ValueError):
        return None, "no PMC traffic file committed for these kernels"


def hbm_rooflines(dev, size: int, K: int, frames: int, reps: int = 10) -> dict:
    """HBM roofline of the heat-map kernels: algorithmic bytes per launch / time per launch, HIP events on the launch
    stream around the C-ABI CALLS themselves (buffers allocated beforehand - no Python wrapper, autograd or allocation inside the timed
    region).  Bytes per frame: one K-stack of fp32 heat-maps = K*h*w*4 (626 688 B at 96x96, K = 17); decode fwd reads it once, decode bwd
    reads it and writes the gradient stack, generation writes it once, heat-map MSE fwd reads two stacks, its bwd reads two and writes one.
    The fused decode is fp32-VALU work by construction (one pass over the map, ~9 kFLOP per heat-map pixel for the x16 up-sampled
    soft-argmax): its VALU fraction is reported next to the HBM fraction."""
    import ctypes as C

    from lightning_pose_amd import _lib, ops
    from lightning_pose_amd.ops import _p

    lib = _lib.lib()
    h = size // 4
    stack = float(frames * K * h * h * 4)
    heat = torch.softmax(torch.randn(frames, K, h * h, device=dev) * 4.0, -1).reshape(frames, K, h, h).contiguous()
    kp = (torch.rand(frames, K, 2, device=dev) * size).contiguous()
    fm = ops.DecodeFrameMap(None, False, None, 1, size, size, K)
    targ = ops.generate_heatmaps(kp, size, size, (h, h)).contiguous()
    tables, _keep = ops._device_tables(h, h, 2, heat.device)
    kp_aug, kp_frame = torch.empty(frames, K, 2, device=dev), torch.empty(frames, K, 2, device=dev)
    conf, stats = torch.empty(frames, K, device=dev), torch.empty(frames, K, 4, device=dev)
    g_frame, g_heat = torch.ones(frames, K, 2, device=dev), torch.empty_like(heat)
    out_hm = torch.empty_like(heat)
    ws = torch.empty(int(lib.lp_heatmap_mse_workspace_bytes(frames, K)), device=dev, dtype=torch.uint8)
    loss, go = torch.empty(1, device=dev), torch.ones(1, device=dev)
    st = ops._stream

    cases = {
        "decode_fwd": (stack, lambda: lib.lp_decode_fwd(_p(heat), frames, K, h, h, 2, 1000.0, C.byref(tables), C.byref(fm.struct), _p(kp_aug),
                                                        _p(kp_frame), _p(conf), _p(stats), 0, st())),
        "decode_bwd": (2 * stack, lambda: lib.lp_decode_bwd(_p(heat), frames, K, h, h, 2, 1000.0, C.byref(tables), C.byref(fm.struct), _p(stats),
                                                            None, _p(g_frame), _p(g_heat), 0, 0, st())),
        "heatmap_gen": (stack, lambda: lib.lp_heatmap_gen(_p(kp), None, frames, K, size, size, h, h, 1.25, _p(out_hm), st())),
        "heatmap_mse_fwd": (2 * stack, lambda: lib.lp_heatmap_loss_fwd(_lib.HM_MSE, _p(targ), _p(heat), frames, K, h, h, _p(loss), _p(ws), st())),
        "heatmap_mse_bwd": (3 * stack, lambda: lib.lp_heatmap_loss_bwd(_lib.HM_MSE, _p(targ), _p(heat), frames, K, h, h, _p(ws), _p(go), _p(g_heat),
                                                                       0, st())),
    }
    # fp32 VALU work of the fused decode: per output pixel of the up-sampled (4h x 4w) map ~12 + 12 FMAs for the two separable tap passes
    # + ~10 for the online soft-max / expectation = ~70 FLOP; x 16 pixels per heat-map pixel.  The backward recomputes it and adds the scatter.
    valu_flop = {"decode_fwd": frames * K * (4 * h) * (4 * h) * 70.0, "decode_bwd": frames * K * (4 * h) * (4 * h) * 130.0}
    cuda = dev.type == "cuda"
    out = {}
    for name, (nbytes, fn) in cases.items():
        for _ in range(2):
            rc = fn()
            assert rc == 0, (name, rc)
        if cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = 1000.0 * e0.elapsed_time(e1) / reps
        else:  # (tests drive this on the emulated kernels)
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            us = 1e6 * (time.perf_counter() - t0) / reps
        gbs = nbytes / us / 1e3
        out[name] = {"us": round(us, 1), "achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}
        if name in valu_flop:
            tf = valu_flop[name] / us / 1e6
            out[name].update(valu_tflops=round(tf, 1), valu_frac=round(tf / VALU_FP32_PEAK_TFLOPS, 3))
    return {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "frames": frames, "algorithmic_bytes_per_frame": int(stack / frames),
            "timed": "HIP events around the C-ABI calls (lp_decode_fwd / lp_decode_bwd / lp_heatmap_gen / lp_heatmap_loss_fwd / _bwd)",
            "note": "the fused decode is fp32-VALU-bound by construction (valu_frac: of the 157 TFLOP/s fp32 vector peak); HBM bytes are its floor",
            "kernels": out}


def cpu_baseline_reference(size: int, K: int, n_lab: int = 4, n_unlab: int = 8, steps: int = 2) -> dict:
    """The reference's OWN SemiSupervisedHeatmapTracker (verbatim modules, executed by oracle/ref_loader.py from /root/reference or from the
    copy oracle/make_ref.py shipped to oracle/_ref/; torch fp32 on the host cores - the reference trains fp32 only, train.py:411-428) timed
    over full optimisation steps: training_step -> backward -> Adam.step on a bounded sample of the bench's workload.  Its losses are the
    three of the bench's four that exist in the reference snapshot (heatmap_mse + temporal + pca_singleview; unimodal_mse does not)."""
    from oracle import ref_loader as R

    T, Fa, L, H = R.load("models.heatmap_tracker"), R.load("losses.factory"), R.load("losses.losses"), R.load("data.heatmaps")
    torch.manual_seed(0)
    cols = [k for k in range(K) if k not in (7, 15, 16)] if K == 17 else list(range(K))
    sup = Fa.LossFactory({"heatmap_mse": {"log_weight": 0.0}}, None)
    unsup = Fa.LossFactory({"temporal": {"log_weight": 5.0, "epsilon": 20.0, "prob_threshold": 0.05}}, None)
    kpca = R.fit_keypoint_pca("pca_singleview", pca_training_array(K, size), components_to_keep=0.99, columns_for_singleview_pca=cols)
    pl = L.PCALoss.__new__(L.PCALoss)
    L.Loss.__init__(pl, log_weight=5.0)
    pl.device, pl.loss_name, pl.pca, pl.epsilon = "cpu", "pca_singleview", kpca, kpca.parameters["epsilon"]
    unsup.loss_instance_dict["pca_singleview"] = pl
    model = T.SemiSupervisedHeatmapTracker(num_keypoints=K, loss_factory=sup, loss_factory_unsupervised=unsup, backbone="resnet50",
                                           pretrained=False, torch_seed=0, image_size=size)
    model.total_unsupervised_importance = torch.tensor(1.0)
    opt = model.configure_optimizers()["optimizer"]
    g = torch.Generator().manual_seed(5)
    kp = torch.rand(n_lab, 2 * K, generator=g) * size
    batch = {
        "labeled": {"images": torch.randn(n_lab, 3, size, size, generator=g), "keypoints": kp,
                    "heatmaps": H.generate_heatmaps(kp.reshape(n_lab, K, 2), size, size, (size // 4, size // 4)),
                    "bbox": torch.tensor([[0.0, 0.0, size, size]]).repeat(n_lab, 1), "idxs": torch.arange(n_lab)},
        "unlabeled": {"frames": torch.randn(n_unlab, 3, size, size, generator=g),
                      "transforms": torch.tensor([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]),
                      "bbox": torch.tensor([[0.0, 0.0, size, size]]).repeat(n_unlab, 1), "is_multiview": False},
    }
    model.train()
    times = []
    for i in range(steps + 1):   # 1 warm-up + `steps` timed steps
        t0 = time.perf_counter()
        opt.zero_grad()
        loss = model.training_step(batch, i)["loss"]
        loss.backward()
        opt.step()
        if i > 0:
            times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    return {"value": round((n_lab + n_unlab) / med, 3), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "reference",
            "sample": f"frames/s of a {n_lab + n_unlab}-frame step, NOT of the {64 + 128}-frame batch the GPU line times (per-frame extrapolation from batch "
                      f"{n_lab + n_unlab}: the reference's 192-frame fp32 step needs more host memory and minutes per step): "
                      f"{len(times)} timed full steps (training_step + backward + Adam) of {n_lab} labeled + {n_unlab} unlabeled {size}x{size} frames, "
                      f"fp32, median {med:.2f} s/step; the reference's own SemiSupervisedHeatmapTracker / LossFactory / losses (verbatim modules from "
                      f"{'/root/reference' if R.REFERENCE_ROOT.startswith('/root/reference') else 'oracle/_ref (oracle/make_ref.py)'}; kornia / "
                      "torchvision restated by oracle/thirdparty.py) with heatmap_mse + temporal + pca_singleview (unimodal_mse is not in the "
                      "reference snapshot), final loss %.5f" % float(loss.detach())}


def cpu_baseline(size: int, K: int, n_lab: int = 4, n_unlab: int = 8, steps: int = 2) -> dict:
    """Oracle (fp32 torch CPU restatement of the reference path) timed on the host cores for a bounded sample."""
    from oracle import restated as O

    torch.manual_seed(0)
    model = O.OracleTracker(K, 2, torch_seed=0)
    opt = torch.optim.Adam([{"params": model.backbone.parameters(), "lr": 0.0}, {"params": model.head.parameters()}], lr=1e-3)
    g = torch.Generator().manual_seed(5)
    kp = torch.rand(n_lab, 2 * K, generator=g) * size
    batch = {
        "labeled": {"images": torch.randn(n_lab, 3, size, size, generator=g), "keypoints": kp,
                    "heatmaps": O.generate_heatmaps(kp.reshape(n_lab, K, 2), size, size, (size // 4, size // 4)),
                    "bbox": torch.tensor([[0.0, 0.0, size, size]]).repeat(n_lab, 1)},
        "unlabeled": {"frames": torch.randn(n_unlab, 3, size, size, generator=g),
                      "transforms": torch.tensor([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]),
                      "bbox": torch.tensor([[0.0, 0.0, size, size]]).repeat(n_unlab, 1), "is_multiview": False},
    }
    cols = [k for k in range(K) if k not in (7, 15, 16)] if K == 17 else list(range(K))
    fit = O.fit_pca(pca_training_array(K, size)[:, sorted(c for k in cols for c in (2 * k, 2 * k + 1))].numpy(), 0.99)
    cfg = {"temporal": {"log_weight": 5.0, "epsilon": 20.0, "prob_threshold": 0.05},
           "pca_singleview": {"log_weight": 5.0, "mean": fit["mean"], "kept_eigenvectors": fit["kept_eigenvectors"], "epsilon": float(fit["epsilon"]),
                              "columns": cols},
           "unimodal_mse": {"log_weight": 5.0, "prob_threshold": 0.05}}
    model.train()
    times = []
    for i in range(steps + 1):  # 1 warm-up + `steps` timed steps (2 by default: the driver's lease should not be mostly CPU work)
        t0 = time.perf_counter()
        opt.zero_grad()
        loss, _ = O.training_step(model, batch, cfg, 1.0)
        loss.backward()
        opt.step()
        if i > 0:
            times.append(time.perf_counter() - t0)
        if len(times) >= 2 and sum(times) >= 25.0:
            break
    steps = len(times)
    med = sorted(times)[len(times) // 2]
    return {"value": round((n_lab + n_unlab) / med, 3), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"frames/s of a {n_lab + n_unlab}-frame step (per-frame extrapolation from batch {n_lab + n_unlab}, not the 192-frame batch of the GPU line): "
                      f"{steps} timed full steps (fwd+bwd+Adam) of {n_lab} labeled + {n_unlab} unlabeled {size}x{size} frames, fp32, "
                      f"median {med:.2f} s/step; oracle/restated.py OracleTracker + training_step (kind 'port': the torch fp32 restatement "
                      "that tests/ pin against the verbatim reference, with the bench's four losses: heatmap_mse + temporal + pca_singleview + unimodal_mse)"}


def cpu_baseline_in_subprocess(size: int, K: int, steps: int) -> dict:
    """The baseline leg in its OWN process (`bench.py --cpu-baseline-only`): executing the reference's modules installs stand-ins for
    torchvision / kornia / lightning in sys.modules (oracle/ref_loader.py) - test infrastructure that must not leak into the process that
    measures the product (the ViT secondary lines import transformers, which probes for the real torchvision)."""
    import subprocess

    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--size", str(size), "--keypoints", str(K),
                        "--cpu-baseline-steps", str(steps)], capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        raise RuntimeError(f"cpu baseline process failed (rc {r.returncode}): {r.stderr[-300:]}")
    return json.loads(lines[-1])


def predict_bench(args, model, batch, dev, rank: int, world: int) -> dict:
    """Inference over the resident frames: predict_step (trunk with folded BatchNorm -> head -> fused decode incl. the bbox map)."""
    import torch.distributed as dist

    from lightning_pose_amd.utils.predictions import predict_batches

    frames = torch.cat([batch["labeled"]["images"], batch["unlabeled"]["frames"]], 0)
    bbox = torch.cat([batch["labeled"]["bbox"], batch["unlabeled"]["bbox"]], 0)
    loader = [{"frames": frames, "bbox": bbox}]
    for _ in range(args.warmup):
        predict_batches(model, loader)
    _sync(dev)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = predict_batches(model, loader)
    _sync(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    n = frames.shape[0] * (frames.shape[1] if frames.dim() == 5 else 1) * world
    value = n * args.steps / elapsed
    gf = TRAIN_GFLOP_PER_FRAME.get(args.size) if args.backbone == "resnet50" else VIT_S_TRAIN_GFLOP_PER_FRAME.get(args.size)
    line = {"metric": f"inference frames/sec (whole node), {args.backbone} {args.size}x{args.size} {args.keypoints}-kp", "value": round(value, 2),
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"predict_step over {n // world} resident frames per GPU (eval mode, no tape), keypoints + confidences out",
                       "global_batch": n, "parallelism": f"dp{world}", "finite": bool(torch.isfinite(out[0][0]).all())}}
    if gf:  # forward only = a third of the training FLOPs per frame
        line["model_tflops_per_gpu"] = round(value / world * gf / 3 / 1e3, 2)
        line["mfma_frac_end_to_end"] = round(value / world * gf / 3 / 1e3 / MFMA_BF16_PEAK_TFLOPS, 4)
    return line


def _decode_prune_label(model) -> str:
    """which decode kernels THIS model's chooser (ops._DecodePruneAuto, one per tracker) ended on"""
    if os.environ.get("LP_DECODE_PRUNE", "auto").strip().lower() not in ("", "auto"):
        return "environment (LP_DECODE_PRUNE=%s)" % os.environ["LP_DECODE_PRUNE"]
    st = getattr(model, "_decode_prune", None)
    if st is None:
        return "unset"
    return "pruned kernels (chosen from the maps)" if st.want == 1 else "plain kernels (chosen from the maps)" if st.calls >= st.FIRST else "plain kernels"


def fit_line(args, dev, rank: int, world: int) -> dict:
    """What a user's training loop gets: Trainer.fit over the device-side producers - decoded uint8 frames on the HOST (a video of
    `unlabeled`-frame windows, a labeled set of uint8 images + stored keypoints, both at the bundled example's 406 x 396 source size) ->
    FrameWindowSource (pinned copy on a side stream, one window ahead) -> VideoFramePipeline (antialiased resize + normalise) /
    LabeledBatchProducer (bicubic resize, keypoint projection, heat-map targets, all on the device) -> the same step as the headline line.
    PCIe-inclusive by construction (56 MB of video frames + 31 MB of labeled images per step); the logged scalars reach the host every
    `log_every_n_steps` only.  Timed: one fit() of `steps` batches after a warm-up fit() of `warmup`.  A fit() call itself costs ~10 ms
    (loader set-up, the epoch-end record's synchronisation): 44.5 / 43.8 / 43.3 / 43.0 ms per step at 6 / 12 / 24 / 48 timed steps next to
    41.6 for the resident batch on the same box (profiles/r06s_fit_steps.txt), i.e. +2.9 % per step in a long run; the default run's
    secondary line times 24 steps (rounds 4 - 5 timed 6 and read -6 %)."""
    from lightning_pose_amd.data.producers import FrameWindowSource, HostStager, LabeledBatchProducer, VideoFramePipeline
    from lightning_pose_amd.trainer import Trainer

    Hs, Ws, K, size = 406, 396, args.keypoints, args.size
    model = build_model(dev, K, size, backbone=args.backbone)
    g = torch.Generator().manual_seed(77 + rank)
    n_win = args.warmup + args.steps
    video = torch.randint(0, 256, (n_win * args.unlabeled, Hs, Ws, 3), generator=g, dtype=torch.uint8)
    if dev.type == "cuda":
        video = video.pin_memory()   # (a decoder writes into pinned buffers; FrameWindowSource copies windows of a pinned video as they are)
    lab_u8 = torch.randint(0, 256, (args.labeled, Hs, Ws, 3), generator=g, dtype=torch.uint8).pin_memory() if dev.type == "cuda" else \
        torch.randint(0, 256, (args.labeled, Hs, Ws, 3), generator=g, dtype=torch.uint8)
    kp = torch.rand(args.labeled, K, 2, generator=g) * torch.tensor([Ws, Hs])
    kp[torch.rand(args.labeled, K, generator=g) < 0.088] = float("nan")
    src = FrameWindowSource(video, args.unlabeled, random_shuffle=False, pad_sequences=False, device=dev)
    pipe = VideoFramePipeline([size, size], imgaug="default")
    prod = LabeledBatchProducer(size, size, uniform_heatmaps=True)
    stage, kp_dev = HostStager(dev), kp.to(dev)

    def batches(first: int, n: int):
        it = iter(src)
        for i, frames_u8 in enumerate(it):
            if i < first:
                continue
            if i >= first + n:
                break
            labeled = prod(stage(lab_u8), kp_dev)   # (HostStager: what HeatmapDataset.batch does with the images it loaded)
            yield {"labeled": labeled, "unlabeled": pipe(frames_u8)}

    trainer = Trainer(max_epochs=1, data_parallel=False, log_every_n_steps=50)
    model.total_unsupervised_importance = torch.tensor(1.0)
    if args.warmup > 0:
        trainer.fit(model, lambda epoch: batches(0, args.warmup))
    _sync(dev)
    t0 = time.perf_counter()
    trainer.fit(model, lambda epoch: batches(args.warmup, args.steps))
    _sync(dev)
    elapsed = time.perf_counter() - t0
    frames = (args.labeled + args.unlabeled) * args.steps
    rec = trainer.logged_history[-1] if trainer.logged_history else {}
    return {"metric": f"training frames/sec through Trainer.fit + device-side producers, ResNet-50 {size}x{size} {K}-kp semi-sup",
            "value": round(frames / elapsed, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"Trainer.fit: {args.labeled} labeled uint8 images + a window of {args.unlabeled} uint8 video frames per step, "
                                   f"{Hs}x{Ws} on the host -> pinned copy -> LabeledBatchProducer / VideoFramePipeline on the device -> the headline step "
                                   "(PCIe-inclusive; scalars logged every 50 steps + at the end of the epoch)",
                       "host_records": len(trainer.logged_history), "final_loss": round(float(rec.get("total_loss", float("nan"))), 6)}}


def train_line(args, dev, rank: int, world: int) -> dict:
    """Build the model + batch of ``args`` and measure it: the JSON line (without printing it)."""
    import torch.distributed as dist

    from lightning_pose_amd.trainer import Trainer

    if getattr(args, "fit", False):
        return fit_line(args, dev, rank, world)
    model = build_model(dev, args.keypoints, args.size, backbone=args.backbone, views=args.views, precision=getattr(args, "precision", "bf16-mixed"))
    if args.views > 1:
        batch = synth_multiview_batch(dev, rank, args.size, args.labeled, args.unlabeled, args.keypoints, args.views)
    else:
        batch = synth_batch(dev, rank, args.size, args.labeled, args.unlabeled, args.keypoints)
    if args.predict:
        return predict_bench(args, model, batch, dev, rank, world)

    if getattr(args, "peaked", False):
        # the maps of a TRAINED head are single peaks; a random-init head (xavier gain 0.01) gives numerically flat ones.  Scaling the head's
        # weights makes the soft-max outputs peaked (at arbitrary places - enough for the decode kernels, whose cost depends on how many
        # pixels carry weight, not on where): what the decode costs in a real run, and whether the pruned kernels get chosen
        sd = model.state_dict()
        for k_ in sd:
            if k_.startswith("head") and k_.endswith("weight"):
                sd[k_] = sd[k_] * 200
        model.load_state_dict(sd)
    trainer = Trainer(max_epochs=1, data_parallel=dist.is_initialized(), sync_batchnorm=not args.no_sync_bn)
    trainer.setup(model)
    if getattr(args, "unfrozen", False):   # the regime after UnfreezeBackbone fired (reference callbacks.py:126-148): every group trains,
        for g_ in model.optimizers().param_groups:   # so the transposed weight copies of the data gradients are refreshed every step
            g_["lr"] = g_["lr"] if g_["lr"] > 0 else 1e-4
    model.train()
    model.total_unsupervised_importance = torch.tensor(1.0)

    def barrier():
        if dist.is_initialized():
            dist.barrier()

    for i in range(args.warmup):
        trainer.training_batch(model, batch, i)
    _sync(dev)
    barrier()
    # Per-launch HIP events (the roofline entry) bracket every convolution launch of a SAMPLE of the timed steps - every 5th,
    # starting with the 3rd: an event record is a barrier packet on the launch stream, and 632 of them per step cost ~4 % of the
    # step (measured: 3190 vs 3330 frames/s), which would make the probe part of the result it measures.
    prof_sink: list = []
    prof_steps = 0
    mem0 = torch.cuda.memory_stats(dev) if dev.type == "cuda" else {}
    t0 = time.perf_counter()
    for i in range(args.steps):
        sampled = not args.no_profile and i % 5 == min(2, args.steps - 1)
        model.net.profile = prof_sink if sampled else None
        prof_steps += int(sampled)
        loss = trainer.training_batch(model, batch, args.warmup + i)
    host_enqueue = time.perf_counter() - t0  # the host is done enqueueing here; the GPU still drains (no sync inside a step)
    _sync(dev)
    barrier()
    elapsed = time.perf_counter() - t0
    prof = prof_sink
    model.net.profile = None
    # host cost of one step when nothing throttles it: enqueue a step onto the idle GPU and stop the clock BEFORE synchronising
    # (in the timed loop above the host runs ahead until the runtime's queue back-pressure paces it to the GPU)
    t1 = time.perf_counter()
    trainer.training_batch(model, batch, args.warmup + args.steps)
    host_idle_queue = time.perf_counter() - t1
    _sync(dev)
    if dist.is_initialized():
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    solo = trainer.dp is None or not trainer.dp.active

    # what crosses GPUs per step: SyncBatchNorm all-reduces (one per BatchNorm layer and direction, both segments of the joint pass in one
    # message), the gradient buckets, one packed message of logged scalars
    n_msgs = getattr(model.net, "sync_bn_messages", 0) // max(1, args.warmup + args.steps + 1)
    # (the messages are lp_fxsum values - two int64 words = 16 B per sum - [segments = 2][2 sums][C] per BatchNorm layer, forward and backward)
    bn_bytes = sum(2 * 2 * b.C * 16 for b in getattr(model.net.plan, "bns", [])) * 2 if getattr(model.net, "sync_bn", False) else 0
    comm = {"sync_bn_messages": n_msgs, "sync_bn_bytes": bn_bytes, "grad_buckets": (0 if solo else -(-model.net.G.numel() * 4 // (64 << 20))),
            "grad_bytes": 0 if solo else model.net.G.numel() * 4, "logged_scalar_messages": 0 if solo else 1}
    if not solo:
        comm["backend"] = dist.get_backend()
        comm["buckets_sent_during_backward"] = trainer.dp.buckets_during_backward
        comm["sync_bn_transport"] = ("all_gather + local add in rank order (LP_SYNCBN_GATHER=1)" if getattr(model.net, "sync_bn_gather", False)
                                     else "all_reduce")
    mem1 = torch.cuda.memory_stats(dev) if dev.type == "cuda" else {}
    memory = {"max_allocated_gb": round(mem1.get("allocated_bytes.all.peak", 0) / 2 ** 30, 2), "max_reserved_gb": round(mem1.get("reserved_bytes.all.peak", 0) / 2 ** 30, 2),
              "device_allocs_in_timed_steps": mem1.get("num_device_alloc", 0) - mem0.get("num_device_alloc", 0),
              "device_frees_in_timed_steps": mem1.get("num_device_free", 0) - mem0.get("num_device_free", 0),
              "alloc_retries": mem1.get("num_alloc_retries", 0), "ooms": mem1.get("num_ooms", 0)}
    frames_per_step = (args.labeled + args.unlabeled) * args.views * world
    value = frames_per_step * args.steps / elapsed
    prec = getattr(args, "precision", "bf16-mixed")
    is_vit = args.backbone != "resnet50"
    arch = {"resnet50": "ResNet-50", "vits_dino": "ViT-S/16", "vitb_dino": "ViT-B/16"}[args.backbone]
    out = {
        "metric": (f"training view-images/sec (whole node), multiview {arch} {args.views} views x {args.size}x{args.size} {args.keypoints}-kp semi-sup"
                   if args.views > 1 else f"training frames/sec (whole node), {arch} {args.size}x{args.size} {args.keypoints}-kp semi-sup"),
        "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1000 * elapsed / args.steps, 3), "host_enqueue_ms_per_step": round(1000 * host_enqueue / args.steps, 3),
        "host_enqueue_idle_queue_ms": round(1000 * host_idle_queue, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if prec == "fp32" else "bf16", "data": "synthetic",
        "config": {"workload": (f"C5: multiview {arch} SemiSupervisedHeatmapTracker, {args.views} views x {args.size}x{args.size}, K={args.keypoints} per "
                                f"view, {args.labeled} labeled + {args.unlabeled} unlabeled frames (x {args.views} views) per GPU, heatmap_mse + "
                                "temporal + pca_multiview, Adam (backbone lr=0 as at step 0), bf16-mixed; value counts view-images") if args.views > 1 else
                               f"{'C4' if is_vit else 'C2/C3'}: {arch} SemiSupervisedHeatmapTracker {args.size}x{args.size}, K={args.keypoints}, "
                               f"{args.labeled} labeled + {args.unlabeled} unlabeled frames per GPU, heatmap_mse + temporal + "
                               "pca_singleview + unimodal_mse, Adam (" + ("every group trains: backbone unfrozen" if getattr(args, "unfrozen", False)
                                                                          else "backbone lr=0 as at step 0") + "), " +
                               ("fp32 (the reference's own precision, train.py:411-428; the validation executor Fp32Engine on v_mfma_f32_32x32x2_f32: untuned, "
                                "nothing fused)" if prec == "fp32" else "bf16-mixed")
                               + (", head weights x200 (peaked heat-maps: ~4 of 147 456 up-sampled pixels carry weight, as with a trained head)" if getattr(args, "peaked", False) else ""),
                   "global_batch": frames_per_step, "parallelism": f"dp{world}", "sync_batchnorm": bool(getattr(model.net, "sync_bn", False)),
                   "comm_per_step": comm, "memory": memory,
                   "decode_prune": _decode_prune_label(model),
                   "final_loss": round(float(loss), 6)},
    }
    if rank == 0:
        if prof:
            by: dict[str, list[float]] = {}
            tot_ms, tot_flops = 0.0, 0.0
            tot_bytes = 0.0
            tot_roof_ms = 0.0
            for tag, flops, e0, e1, nbytes, *_layer in prof:
                ms = e0.elapsed_time(e1)
                # the launch's OWN roof: whichever of its algorithmic FLOPs at the dense bf16 MFMA peak and its algorithmic bytes at the
                # HBM rate this part sustains (6.3 TB/s, MI355X_MICROARCH.md) takes longer
                roof_ms = 1e3 * max(flops / (MFMA_BF16_PEAK_TFLOPS * 1e12), nbytes / (HBM_SUSTAINED_GBS * 1e9))
                rec = by.setdefault(tag, [0, 0.0, 0.0, 0.0, 0.0])
                rec[0] += 1
                rec[1] += ms
                rec[2] += flops
                rec[3] += nbytes
                rec[4] += roof_ms
                tot_ms += ms
                tot_flops += flops
                tot_bytes += nbytes
                tot_roof_ms += roof_ms
            dump = os.environ.get("LP_DUMP_LAUNCHES")
            if dump:  # per-launch (tag, GFLOP, us) of the LAST timed step, for kernel tuning
                per_step = len(prof) // prof_steps
                with open(dump, "w") as fh:
                    json.dump([[t, round(f / 1e9, 3), round(1000 * a.elapsed_time(b), 1), round(nb / 1e6, 2), (ly[0] if ly else "")] for t, f, a, b, nb, *ly in prof[-per_step:]], fh)
            ach = tot_flops / (tot_ms * 1e-3) / 1e12
            traffic, traffic_src = pmc_traffic() if (not is_vit and args.size == 384 and args.views == 1) else (None, None)
            out["roofline"] = {
                "bound": "mfma", "kernel": ("lp_gemm_nt / conv_wgrad_kernel / attn_fwd_kernel / attn_bwd_kv_kernel (all MFMA launches of the ViT)" if is_vit
                                            else "conv_pipe_kernel / conv_wgrad_pipe_kernel / conv_igemm_kernel / conv_wgrad_kernel (all MFMA convolution launches; by_kernel splits them)"),
                "achieved": round(ach, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4),
                # each launch against its OWN roof: sum over launches of max(FLOPs / 2.5 PFLOP/s, algorithmic bytes / 6.3 TB/s), over the
                # measured time - 1.0 would mean every launch is either MFMA- or HBM-bound at the hardware rate
                "attainable_frac": round(tot_roof_ms / tot_ms, 4),
                "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": round(tot_bytes / len(prof)),
                "traffic_over_algorithmic": round(traffic / (tot_bytes / len(prof)), 3) if traffic else None,
                "hbm_gbs_algorithmic": round(tot_bytes / (tot_ms * 1e-3) / 1e9, 1),
                "launches_per_step": len(prof) // prof_steps, "mfma_ms_per_step": round(tot_ms / prof_steps, 3),
                "sampled_steps": prof_steps,
                "by_kernel": {k: {"launches_per_step": v[0] // prof_steps, "avg_us": round(1000 * v[1] / v[0], 2),
                                  "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 2), "algorithmic_gbs": round(v[3] / (v[1] * 1e-3) / 1e9, 1),
                                  "attainable_frac": round(v[4] / v[1], 4)}
                              for k, v in sorted(by.items())},
            }
        gf = (VIT_S_TRAIN_GFLOP_PER_FRAME if args.backbone == "vits_dino" else {} if is_vit else TRAIN_GFLOP_PER_FRAME).get(args.size)
        if gf and prec != "fp32":   # (the fp32 line runs on the fp32 MFMA pipe: not priced against the bf16 peak)
            out["model_tflops_per_gpu"] = round(value / world * gf / 1e3, 2)
            out["mfma_frac_end_to_end"] = round(value / world * gf / 1e3 / MFMA_BF16_PEAK_TFLOPS, 4)
        if world == 1 and not dist.is_initialized() and not args.no_profile and args.views == 1 and not getattr(args, "_secondary", False):
            try:  # secondary rooflines; never allowed to cost the measured line
                out["roofline_hbm"] = hbm_rooflines(dev, args.size, args.keypoints, args.labeled + args.unlabeled)
            except Exception as e:  # noqa: BLE001
                out["roofline_hbm"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not dist.is_initialized() and not args.no_cpu_baseline and not is_vit and args.views == 1 and not getattr(args, "_secondary", False):
            try:
                out["cpu_baseline"] = cpu_baseline_in_subprocess(args.size, args.keypoints, args.cpu_baseline_steps)
            except Exception as e:  # noqa: BLE001 - the baseline must never cost the measured line
                out["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
                                       "sample": f"failed: {type(e).__name__}: {e}"}
    return out


def self_launch(n: int, argv: list[str], entry: str | None = None, env: dict | None = None) -> int:
    """``python bench.py --gpus N`` typed without a launcher: spawn the N ranks ourselves, exactly as the documented command does
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free port> bench.py ...`, one
    process per GPU, LOCAL_RANK -> device), the way the reference's trainer spawns its own DDP ranks from `devices=cfg.training.num_gpus`
    (reference train.py:411-428).  The children inherit stdout, so rank 0's ONE JSON line is this process's output; returns the launcher's
    exit code.  ``entry`` / ``env`` (tests): the script each rank runs and its environment."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ if env is None else env)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), entry or os.path.abspath(__file__), *argv]
    if os.environ.get("LP_BENCH_DRY_RUN") == "1":   # (tests: the launcher command and the environment its ranks would get, nothing started)
        print(json.dumps({"cmd": cmd, "env": {k: env[k] for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "OMP_NUM_THREADS", "LP_SYNCBN_GATHER") if k in env}}))
        return 0
    return subprocess.run(cmd, env=env).returncode


def main(argv: list[str] | None = None, device: torch.device | None = None) -> None:
    """``device`` (tests only): run the whole flow on that device - the CPU with the emulated kernel library - instead of cuda:LOCAL_RANK."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=384)
    ap.add_argument("--labeled", type=int, default=64)
    ap.add_argument("--unlabeled", type=int, default=128)
    ap.add_argument("--keypoints", type=int, default=17)
    ap.add_argument("--backbone", default="resnet50", choices=["resnet50", "vits_dino", "vitb_dino"],
                    help="resnet50 = BASELINE configs C2/C3 (the headline metric); vits_dino = config C4")
    ap.add_argument("--views", type=int, default=1, help="4 = BASELINE config C5 (multiview: --size 256 --labeled 16 --unlabeled 32 "
                    "gives the same 192 images per GPU); frames/s then counts view-images")
    ap.add_argument("--predict", action="store_true", help="secondary line: inference frames/s (eval mode, BatchNorm folded into the "
                    "convolutions, fused decode) over the same frames; the headline metric stays the training step")
    ap.add_argument("--no-secondary", dest="secondary", action="store_false", help="skip the short secondary lines (256 px, ViT-S, multiview, "
                    "inference) the default single-GPU run appends under \"secondary\"")
    ap.add_argument("--unfrozen", action="store_true", help="secondary line: the backbone group trains too (lr > 0), as after UnfreezeBackbone")
    ap.add_argument("--peaked", action="store_true", help="secondary line: head weights x200 -> peaked heat-maps as a trained head gives them "
                    "(the decode then picks its pruned kernels by itself, ops._DecodePruneAuto)")
    ap.add_argument("--precision", default="bf16-mixed", choices=["bf16-mixed", "fp32"], help="fp32 = the reference's own precision on the "
                    "validation executor (secondary line resnet50_384_fp32; the headline is BASELINE.json's bf16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-steps", type=int, default=2, help="timed CPU steps of the baseline leg (after 1 warm-up step)")
    ap.add_argument("--fit", action="store_true", help="secondary line: Trainer.fit over the device-side producers (uint8 host frames -> "
                    "FrameWindowSource / LabeledBatchProducer / VideoFramePipeline -> step), i.e. what a user's training loop gets")
    ap.add_argument("--no-sync-bn", action="store_true", help="N > 1: per-rank BatchNorm statistics (the reference sets sync_batchnorm=True, train.py:427; A/B only)")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-launch HIP events")
    ap.add_argument("--syncbn-gather", action="store_true", help="N > 1: SyncBatchNorm messages as ONE all-gather + an ordered add of the ranks' rows "
                    "(int64 fixed-point sums) instead of an all-reduce (= LP_SYNCBN_GATHER=1; the A/B the first 8-GPU call decides)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="(internal) print the cpu_baseline object and exit: the reference's own "
                    "step when its modules are present (/root/reference, or oracle/_ref on the GPU box), the restated port otherwise")
    args = ap.parse_args(argv)
    if args.syncbn_gather:
        os.environ["LP_SYNCBN_GATHER"] = "1"   # (read by Engine.__init__; self_launch's children inherit it)
    if args.cpu_baseline_only:
        from oracle import ref_loader as _R
        print(json.dumps((cpu_baseline_reference if _R.available() else cpu_baseline)(args.size, args.keypoints, steps=args.cpu_baseline_steps)),
              flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and device is None:   # no launcher around us: be the launcher
        raise SystemExit(self_launch(args.gpus, list(sys.argv[1:] if argv is None else argv)))

    import torch.distributed as dist

    from lightning_pose_amd.distributed import init_process_group_from_env
    from lightning_pose_amd.trainer import Trainer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:   # (checked before the rendezvous: a mismatched launcher must fail, not hang waiting for ranks that never come)
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    rank, local_rank, world = init_process_group_from_env()
    if os.environ.get("LP_FORCE_DEVICE") is not None:  # functional multi-rank test on a 1-GPU box (with LP_DIST_BACKEND=gloo)
        local_rank = int(os.environ["LP_FORCE_DEVICE"])
    dev = torch.device(f"cuda:{local_rank}") if device is None else device
    if dev.type == "cuda":
        torch.cuda.set_device(dev)

    out = train_line(args, dev, rank, world)
    if rank == 0:
        headline = (args.size, args.labeled, args.unlabeled, args.keypoints, args.views, args.backbone) == (384, 64, 128, 17, 1, "resnet50")
        if args.secondary and world == 1 and not dist.is_initialized() and not args.predict and headline and dev.type == "cuda":
            # the other BASELINE.json configs / north_star sizes in the same run (short: 3 timed steps each), so the driver's BENCH file
            # holds them too; the headline stays `value`
            import copy
            sec = {}
            for tag, over in (("resnet50_256", dict(size=256)), ("resnet50_384_unfrozen_backbone", dict(unfrozen=True)),
                              ("resnet50_384_peaked_maps", dict(peaked=True, warmup=4, steps=5)),
                              ("resnet50_384_trainer_fit", dict(fit=True, warmup=2, steps=24)), ("c4_vits_dino_384", dict(backbone="vits_dino")),
                              ("c5_multiview_4x256", dict(views=4, size=256, labeled=16, unlabeled=32)),
                              ("predict_resnet50_384", dict(predict=True)), ("predict_vits_dino_384", dict(predict=True, backbone="vits_dino")),
                              # the reference trains fp32 only (train.py:411-428): the same step at its precision, on the fp32 validation executor
                              ("resnet50_384_fp32", dict(precision="fp32", warmup=1, steps=2, no_profile=True))):
                a2 = copy.copy(args)
                a2.steps, a2.warmup, a2.no_cpu_baseline, a2._secondary = 3, 2, True, True
                for k_, v_ in over.items():
                    setattr(a2, k_, v_)
                try:
                    torch.cuda.empty_cache() if dev.type == "cuda" else None
                    r = train_line(a2, dev, rank, world)
                    sec[tag] = {k_: r[k_] for k_ in ("metric", "value", "unit", "dtype", "ms_per_step", "steps", "model_tflops_per_gpu", "mfma_frac_end_to_end") if k_ in r}
                    if "roofline" in r:
                        sec[tag]["roofline"] = {k_: r["roofline"][k_] for k_ in ("achieved", "frac", "unit", "launches_per_step", "mfma_ms_per_step",
                                                                                 "hbm_gbs_algorithmic") if k_ in r["roofline"]}
                    sec[tag]["workload"] = r["config"]["workload"]
                    if "decode_prune" in r["config"]:
                        sec[tag]["decode_prune"] = r["config"]["decode_prune"]
                except Exception as e:  # noqa: BLE001 - never allowed to cost the headline line
                    sec[tag] = {"error": f"{type(e).__name__}: {e}"}
            out["secondary"] = sec
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
