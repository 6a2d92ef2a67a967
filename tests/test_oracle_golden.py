"""Pin oracle.restated against golden vectors produced by the VERBATIM reference modules
(tests/golden/make_golden.py).  CPU only."""

import os

import numpy as np
import pytest
import torch

from oracle import restated as O


def close(a, b, atol=1e-5, rtol=1e-5):
    a = torch.as_tensor(np.asarray(a)) if not torch.is_tensor(a) else a
    b = torch.as_tensor(np.asarray(b)) if not torch.is_tensor(b) else b
    torch.testing.assert_close(a.float(), b.float(), atol=atol, rtol=rtol, equal_nan=True)


@pytest.mark.parametrize("tag,dss", [("a", (1, 2, 3)), ("b", (1, 2, 3)), ("c", (1, 2)), ("flat", (2,)), ("edge", (2,))])
def test_soft_argmax(golden, tag, dss):
    g = golden("decode")
    for ds in dss:
        kp, conf = O.soft_argmax(g.t(f"{tag}_in"), ds, 1000.0)
        # T=1000 amplifies fp32 rounding of the upsampled logits; 2e-4 px covers op-order differences
        close(kp, g[f"{tag}_kp_ds{ds}"], atol=2e-4, rtol=0)
        close(conf, g[f"{tag}_conf_ds{ds}"], atol=2e-5, rtol=0)


def test_upsample_matrix_form(golden):
    g = golden("decode")
    x = g.t("up_in")
    close(O.upsample2x(x), g["up_out"], atol=1e-6)
    uh = O.upsample_matrix(x.shape[2], 1)
    uw = O.upsample_matrix(x.shape[3], 1)
    y = torch.einsum("ih,bkhw,jw->bkij", uh, x.double(), uw)
    close(y, g["up_out"], atol=1e-5)


def test_generate_heatmaps(golden):
    g = golden("heatmaps")
    kp, vis = g.t("kp"), g.t("vis")
    close(O.generate_heatmaps(kp, 128, 128, (32, 32)), g["hm_novis"], atol=1e-7)
    close(O.generate_heatmaps(kp, 128, 128, (32, 32), visibility=vis), g["hm_vis"], atol=1e-7)
    close(O.generate_heatmaps(kp, 128, 160, (32, 40), sigma=2.0), g["hm_rect"], atol=1e-7)


def test_confidence_window(golden):
    g = golden("heatmaps")
    close(O.confidence_window(g.t("cw_p"), g.t("cw_locs")), g["cw_out"], atol=1e-7)


def test_geometry(golden):
    g = golden("geometry")
    kp = g.t("kp")
    close(O.undo_affine(kp, g.t("A")), g["undo_single"], atol=1e-4)
    close(O.undo_affine(kp, g.t("As")), g["undo_perframe"], atol=1e-4)
    close(O.undo_affine(kp, g.t("As")[:2], is_multiview=True), g["undo_multiview"], atol=1e-4)
    close(O.undo_affine(kp, torch.tensor([-1.0])), g["undo_sentinel"], atol=0)
    close(O.model_to_frame(kp, 128, 160, g.t("bbox")), g["m2f_single"], atol=1e-4)
    close(O.model_to_frame(kp, 128, 160, g.t("bbox")), g["m2f_labeled"], atol=1e-4)
    close(O.model_to_frame(kp, 128, 160, g.t("bbox2"), num_views=2), g["m2f_multiview"], atol=1e-4)


def test_losses(golden):
    g = golden("losses")
    t, p = g.t("hm_targ"), g.t("hm_pred")
    close(O.heatmap_mse_loss(t, p), g["heatmap_mse"], atol=1e-7)
    close(O.heatmap_kl_loss(t, p), g["heatmap_kl"], atol=1e-6)
    close(O.heatmap_js_loss(t, p), g["heatmap_js"], atol=1e-6)
    kp, conf = g.t("t_kp"), g.t("t_conf")
    close(O.temporal_loss(kp), g["temporal_plain"], atol=1e-5)
    close(O.temporal_loss(kp, None, 5.0), g["temporal_eps"], atol=1e-5)
    close(O.temporal_loss(kp, conf, 3.0, 0.3), g["temporal_conf"], atol=1e-5)
    close(O.temporal_loss(kp, conf, g.t("t_eps_list"), 0.3), g["temporal_epslist"], atol=1e-5)
    close(O.rmse_loss(g.t("r_targ"), g.t("r_pred")), g["rmse"], atol=1e-6)


@pytest.mark.parametrize("tag,ctk", [("sv99", 0.99), ("sv3", 3)])
def test_pca_singleview(golden, tag, ctk):
    g = golden("losses")
    cols = [int(c) for c in g["pca_cols"]]
    fit = O.fit_pca(O.pca_format_singleview(g.t("pca_fit_data"), cols).numpy(), ctk)
    close(fit["mean"], g[f"pca_{tag}_mean"], atol=1e-4)
    close(fit["kept_eigenvectors"], g[f"pca_{tag}_kept"], atol=1e-5)
    close(fit["epsilon"], g[f"pca_{tag}_eps"], atol=1e-4)
    data = O.pca_format_singleview(g.t(f"pca_{tag}_test"), cols)
    mean, kept = g.t(f"pca_{tag}_mean"), g.t(f"pca_{tag}_kept")
    close(O.pca_reprojection_error(data, mean, kept), g[f"pca_{tag}_err"], atol=1e-4)
    close(O.pca_loss(data, mean, kept, float(g[f"pca_{tag}_eps"])), g[f"pca_{tag}_loss"], atol=1e-5)


def test_pca_multiview(golden):
    g = golden("losses")
    mcm = [[int(c) for c in row] for row in g["pca_mv_mcm"]]
    fit = O.fit_pca(O.pca_format_multiview(g.t("pca_mv_fit_data"), mcm).numpy(), 3, loss_type="pca_multiview")
    close(fit["kept_eigenvectors"], g["pca_mv_kept"], atol=1e-5)
    close(fit["epsilon"], g["pca_mv_eps"], atol=1e-4)
    data = O.pca_format_multiview(g.t("pca_mv_test"), mcm)
    close(O.pca_loss(data, g.t("pca_mv_mean"), g.t("pca_mv_kept"), float(g["pca_mv_eps"])), g["pca_mv_loss"], atol=1e-5)


def test_factory_totals(golden):
    g = golden("losses")
    mse = O.heatmap_mse_loss(g.t("hm_targ"), g.t("hm_pred"))
    for aw in (None, 0.0, 0.5):
        close(O.factory_total({"heatmap_mse": (mse, 0.0)}, aw), g[f"fac_sup_aw{aw}"], atol=1e-7)
    tl = O.temporal_loss(g.t("t_kp"), g.t("t_conf"), 3.0, 0.3)
    for aw in (None, 0.0, 0.5, 1.0):
        close(O.factory_total({"temporal": (tl, 5.0)}, aw), g[f"fac_unsup_aw{aw}"], atol=1e-7)


@pytest.mark.parametrize("tag,kind,eps,thr", [("mse_plain", "mse", 0.0, 0.0), ("kl_plain", "kl", 0.0, 0.0), ("mse_thr", "mse", 0.0, 0.4),
                                              ("kl_thr_eps", "kl", 0.5, 0.4), ("mse_eps_list", "mse", [0.0, 2e-5, 1e-4, 1.0], 0.2)])
def test_temporal_heatmap_loss(golden, tag, kind, eps, thr):
    """oracle restatement vs the verbatim TemporalHeatmapLoss (value and gradient)"""
    g = golden("temporal_heatmap")
    p = g.t("hm").clone().requires_grad_(True)
    val = O.temporal_heatmap_loss(p, g.t("conf"), eps, thr, kind)
    (0.7 * val).backward()
    assert float(val) == pytest.approx(float(g[f"{tag}_loss"]), rel=1e-5, abs=1e-10)
    torch.testing.assert_close(p.grad, g.t(f"{tag}_grad"), rtol=1e-4, atol=1e-9)


@pytest.mark.skipif(not os.path.isdir("/root/reference/lightning_pose"), reason="/root/reference not present (build container only)")
def test_committed_fixtures_are_what_the_reference_produces_now(tmp_path):
    """tests/golden/make_golden.py re-run against the reference tree reproduces EVERY committed fixture bit for bit
    (seeded inputs, verbatim reference modules): the fixtures are outputs of the reference, not of this repository."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = ["decode", "heatmaps", "geometry", "losses", "callbacks", "tracker_step", "predictions", "labeled_targets", "temporal_heatmap"]
    env = dict(os.environ, LP_GOLDEN_OUT=str(tmp_path))
    res = subprocess.run([sys.executable, os.path.join(root, "tests", "golden", "make_golden.py"), *names], env=env, capture_output=True,
                         text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    for name in names:
        with np.load(os.path.join(root, "tests", "golden", name + ".npz")) as a, np.load(tmp_path / (name + ".npz")) as b:
            assert sorted(a.files) == sorted(b.files), name
            for k in a.files:
                np.testing.assert_array_equal(a[k], b[k], err_msg=f"{name}:{k}")
