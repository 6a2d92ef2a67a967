"""Host-side logic that needs no device: registry surface, callbacks, PCA fit, decode tables, plan layout."""

import inspect

import numpy as np
import pytest
import torch

from oracle import restated as O


def test_callbacks_match_reference_golden(golden):
    from lightning_pose_amd.callbacks import AnnealWeight, UnfreezeBackbone

    g = golden("callbacks")

    class M:
        current_epoch = 0
        global_step = 0

    m = M()
    cb = AnnealWeight("total_unsupervised_importance", init_val=0.0, increase_factor=0.01, final_val=1.0, freeze_until_epoch=3)
    cb.on_train_start(None, m)
    vals = []
    for e in range(120):
        m.current_epoch = e
        cb.on_train_epoch_start(None, m)
        vals.append(float(m.total_unsupervised_importance))
    np.testing.assert_allclose(vals, g["anneal"], rtol=1e-6)
    ub = UnfreezeBackbone(unfreeze_epoch=5, initial_ratio=0.1, warm_up_ratio=1.5)
    lrs, head_lr = [], 1e-3
    for e in range(20):
        if e == 12:
            head_lr *= 0.5
        lrs.append(ub._get_backbone_lr(None, e, head_lr) if not ub._warmed_up else -1.0)
    np.testing.assert_allclose(lrs, g["unfreeze_lr"], rtol=1e-9)


@pytest.mark.parametrize("tag,ctk", [("sv99", 0.99), ("sv3", 3)])
def test_keypoint_pca_fit_matches_reference(golden, tag, ctk):
    from lightning_pose_amd.utils.pca import KeypointPCA

    g = golden("losses")
    cols = [int(c) for c in g["pca_cols"]]
    kp = KeypointPCA("pca_singleview", components_to_keep=ctk, columns_for_singleview_pca=cols, data_arr=g["pca_fit_data"])
    kp()
    np.testing.assert_allclose(kp.parameters["mean"].numpy(), g[f"pca_{tag}_mean"], atol=1e-4)
    np.testing.assert_allclose(kp.parameters["kept_eigenvectors"].numpy(), g[f"pca_{tag}_kept"], atol=1e-5)
    assert float(kp.parameters["epsilon"]) == pytest.approx(float(g[f"pca_{tag}_eps"]), abs=1e-4)
    assert kp.index_table(7).tolist() == [cols]


def test_keypoint_pca_multiview_fit(golden):
    from lightning_pose_amd.utils.pca import KeypointPCA

    g = golden("losses")
    mcm = [[int(c) for c in r] for r in g["pca_mv_mcm"]]
    with pytest.warns(UserWarning):
        kp = KeypointPCA("pca_multiview", components_to_keep=0.9, mirrored_column_matches=mcm, data_arr=g["pca_mv_fit_data"])
        kp()
    np.testing.assert_allclose(kp.parameters["kept_eigenvectors"].numpy(), g["pca_mv_kept"], atol=1e-5)
    assert kp.index_table(8).tolist() == [[0, 4], [1, 5], [2, 6]]


@pytest.mark.parametrize("n,ds", [(8, 1), (8, 2), (12, 2), (16, 3), (96, 2), (64, 2)])
def test_decode_tables_reproduce_the_composite_operator(n, ds):
    from lightning_pose_amd import _tables

    u = O.upsample_matrix(n, ds).numpy()
    np.testing.assert_allclose(_tables.upsample_matrix(n, ds), u, atol=1e-12)
    t = _tables.axis_tables(n, ds)
    r = 1 << ds
    rebuilt = np.zeros_like(u)
    for j in range(n):
        rebuilt[j * r:(j + 1) * r, t["row_base"][j]:t["row_base"][j] + t["ty"]] = t["row_taps"][j]
    np.testing.assert_allclose(rebuilt, u, atol=1e-7)
    rebuilt = np.zeros_like(u)
    for c in range(n * r):
        rebuilt[c, t["col_start"][c]:t["col_start"][c] + t["tx"]] = t["col_taps"][c, :t["tx"]]
    np.testing.assert_allclose(rebuilt, u, atol=1e-7)
    rebuilt = np.zeros_like(u)
    for q in range(n):
        rebuilt[t["colT_start"][q]:t["colT_start"][q] + t["tc"], q] = t["colT_taps"][q]
    np.testing.assert_allclose(rebuilt, u, atol=1e-7)


def test_plan_layout_and_parameter_count():
    from lightning_pose_amd.engine import build_plan

    plan = build_plan(17, 2)
    assert len(plan.blocks) == 16 and len(plan.head) == 2 and len(plan.bns) == 53
    logical = 0
    for c in plan.convs:
        logical += c.cout * c.cin * c.k * c.k + (c.cout if c.kind == "convT" else 0)
    logical += sum(2 * b.C for b in plan.bns)
    ref = sum(p.numel() for p in O.OracleTracker(17, 2).parameters())
    assert logical == ref == 23508032 + 80971
    # contiguous, non-overlapping, backbone before head
    offs = sorted([(c.w_off, c.numel + (64 if c.kind == "convT" else 0)) for c in plan.convs] + [(b.g_off, 2 * b.C) for b in plan.bns])
    pos = 0
    for o, n in offs:
        assert o == pos
        pos += n
    assert pos == plan.n_total and plan.head[0].w_off == plan.n_backbone


def test_registry_surface_and_validation():
    from lightning_pose_amd.losses import LossFactory, get_loss_classes
    from lightning_pose_amd.models import get_model_class
    from lightning_pose_amd.models.factory import _validate_loss_model_compatibility
    from lightning_pose_amd.models.heatmap_tracker import HeatmapTracker, SemiSupervisedHeatmapTracker

    assert get_model_class("heatmap", False) is HeatmapTracker and get_model_class("heatmap", True) is SemiSupervisedHeatmapTracker
    with pytest.raises(NotImplementedError):
        get_model_class("regression", False)
    assert {"heatmap_mse", "temporal", "pca_singleview", "pca_multiview", "unimodal_mse"} <= set(get_loss_classes())
    facs = {"supervised": LossFactory({"heatmap_mse": {"log_weight": 0.0}}, None),
            "unsupervised": LossFactory({"temporal": {"log_weight": 5.0}, "unimodal_mse": {"log_weight": 5.0}}, None)}
    _validate_loss_model_compatibility(SemiSupervisedHeatmapTracker, facs)

    class NeedsMissing:
        def __call__(self, embedding, stage=None, **kw):
            return None

    facs["unsupervised"].loss_instance_dict["bad"] = NeedsMissing()
    with pytest.raises(ValueError, match="embedding"):
        _validate_loss_model_compatibility(SemiSupervisedHeatmapTracker, facs)
    assert {"heatmap_kl", "heatmap_js"} <= set(get_loss_classes())
    assert {"temporal_heatmap_mse", "temporal_heatmap_kl"} <= set(get_loss_classes())
    _validate_loss_model_compatibility(SemiSupervisedHeatmapTracker, {
        "supervised": facs["supervised"],
        "unsupervised": LossFactory({"temporal_heatmap_kl": {"loss_name": "temporal_heatmap_kl", "log_weight": 5.0}}, None)})
    with pytest.raises(NotImplementedError):
        LossFactory({"regression": {"log_weight": 0.0}}, None)
    # constructor signature of the reference classes is preserved
    sig = inspect.signature(SemiSupervisedHeatmapTracker.__init__)
    for name in ("num_keypoints", "loss_factory", "loss_factory_unsupervised", "backbone", "downsample_factor", "pretrained",
                 "torch_seed", "optimizer", "optimizer_params", "lr_scheduler", "lr_scheduler_params"):
        assert name in sig.parameters


def test_loss_weight_and_log_names():
    from lightning_pose_amd.losses import TemporalLoss

    t = TemporalLoss(log_weight=5.0, epsilon=[1.0, 2.0])
    assert float(t.weight) == pytest.approx(O.loss_weight(5.0))
    names = [d["name"] for d in t.log_loss(torch.tensor(1.0), "train")]
    assert names == ["train_temporal_loss", "temporal_weight"]


def test_trainer_flushes_logged_scalars_once_per_logging_step_under_gradient_accumulation():
    """with accumulate_grad_batches > 1 the step counter stands still on the non-stepping micro-batches; the flush (a host <->
    device synchronisation) must happen when an optimiser step LANDED on a multiple of log_every_n_steps, not on every micro-batch while
    the counter sits on one - and never twice for the same step."""
    from lightning_pose_amd.trainer import Trainer

    class Opt:
        def zero_grad(self):
            pass

        def step(self):
            pass

    class Sched:
        def step(self):
            pass

    class Module:
        device = torch.device("cpu")
        training = True

        def __init__(self):
            self.global_step, self.current_epoch, self.logged, self._opt = 0, 0, {}, Opt()
            self.w = torch.zeros(1, requires_grad=True)

        def train(self, mode=True):
            self.training = mode

        def optimizers(self):
            return self._opt

        def get_scheduler(self, opt):
            return Sched()

        def training_step(self, batch, batch_idx):
            self.logged = {"total_loss": torch.tensor(float(batch_idx))}
            return {"loss": (self.w * 0).sum()}

    model = Module()
    trainer = Trainer(max_epochs=1, data_parallel=False, accumulate_grad_batches=3, log_every_n_steps=2)
    trainer.fit(model, [{} for _ in range(14)])   # 14 micro-batches: steps after batches 2, 5, 8, 11 and the trailing partial group (13)
    assert model.global_step == 5
    steps = [h["step"] for h in trainer.logged_history]
    assert steps == [2.0, 4.0, 5.0], steps         # multiples of 2 as they are reached, + the end of the epoch; no duplicates, nothing at step 0
    assert [h["total_loss"] for h in trainer.logged_history] == [5.0, 11.0, 13.0]
