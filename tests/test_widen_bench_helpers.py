"""bench.py's synthetic workloads drive the product step (tiny sizes, emulated kernels on CPU / the device under -m gpu): the
single-view C2 batch and the multiview C5 batch (--views)."""

import numpy as np
import torch

import bench


def _step(model, batch):
    model.train()
    model.total_unsupervised_importance = torch.tensor(1.0)
    opt = model.configure_optimizers()["optimizer"]
    opt.zero_grad()
    loss = model.training_step(batch, 0)["loss"]
    loss.backward()
    opt.step()
    return {k: float(v) for k, v in model.logged.items()}


def test_singleview_workload_steps(stack_backend):
    dev = stack_backend
    K, size = 17, 64
    model = bench.build_model(dev, K, size)
    batch = bench.synth_batch(dev, 0, size, 2, 3, K)
    assert tuple(batch["labeled"]["heatmaps"].shape) == (2, K, 16, 16) and tuple(batch["unlabeled"]["transforms"].shape) == (2, 3)
    got = _step(model, batch)
    for name in ("train_heatmap_mse_loss", "train_temporal_loss", "train_pca_singleview_loss", "train_unimodal_mse_loss", "total_loss"):
        assert np.isfinite(got[name]), name


def test_multiview_workload_steps(stack_backend):
    dev = stack_backend
    K, V, size = 3, 2, 64
    model = bench.build_model(dev, K, size, views=V)
    batch = bench.synth_multiview_batch(dev, 0, size, 1, 3, K, V)
    assert tuple(batch["labeled"]["images"].shape) == (1, V, 3, size, size)
    assert tuple(batch["labeled"]["heatmaps"].shape) == (1, K * V, 16, 16) and tuple(batch["labeled"]["keypoints"].shape) == (1, 2 * K * V)
    assert tuple(batch["unlabeled"]["frames"].shape) == (3, V, 3, size, size) and tuple(batch["unlabeled"]["transforms"].shape) == (V, 2, 3)
    assert tuple(batch["unlabeled"]["bbox"].shape) == (3, 4 * V) and batch["unlabeled"]["is_multiview"] is True
    got = _step(model, batch)
    for name in ("train_heatmap_mse_loss", "train_temporal_loss", "train_pca_multiview_loss", "total_loss"):
        assert np.isfinite(got[name]), name
    assert float(model.net.G.abs().sum()) > 0
    # the fit data are affine views of one 3-D cloud: 3 components carry (almost) everything
    pca = model.loss_factory_unsup.loss_instance_dict["pca_multiview"].pca
    assert pca.parameters["kept_eigenvectors"].shape[0] == 3


def test_hbm_rooflines_runs_the_heatmap_kernels(stack_backend):
    out = bench.hbm_rooflines(stack_backend, 64, 3, 4, reps=1)
    assert out["bound"] == "hbm" and out["algorithmic_bytes_per_frame"] == 3 * 16 * 16 * 4
    assert set(out["kernels"]) == {"decode_fwd", "decode_fwd_bwd", "heatmap_gen", "heatmap_mse_fwd_bwd"}
    for v in out["kernels"].values():
        assert v["us"] > 0 and v["achieved"] >= 0  # (the emulator moves kilobytes per millisecond)
