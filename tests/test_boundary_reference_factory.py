"""The drop-in boundary, executed from the REFERENCE's side (missing #4).

1. The reference's own ``lightning_pose/models/factory.py`` - loaded verbatim - builds the PRODUCT's trackers: ``get_model(cfg, data_module,
   loss_factories)`` (:195-319) with INTEGRATION.md section 4's registry switch applied (``get_model_class`` returns
   ``lightning_pose_amd.models.get_model_class(...)``; here the two-line source patch is applied as an attribute assignment on the loaded
   module, nothing else is touched).  That runs, unmodified: the reference's optimizer / scheduler defaults (``models/base.py:105-156``), its
   ``_validate_loss_model_compatibility`` (:139-192 - it introspects the product classes' TypedDict return annotations and the product loss
   classes' ``__call__`` signatures), its constructor-argument assembly (``image_size``, ``num_targets``, ``backbone_checkpoint``,
   ``downsample_factor``) and its checkpoint warm start (:299-317, ``torch.load(...)["state_dict"]`` -> ``load_state_dict(strict=False)``).
2. N4, both directions through FILES: a ``.ckpt`` written by the product loads ``strict=True`` into the verbatim ``HeatmapTracker`` built by the
   UNPATCHED reference factory, and gives the same heat-maps; the reference's ``api/model_runtime.py:143-150`` legacy-key remap
   (``upsampling_layers.*`` -> ``head.upsampling_layers.*``) is restated and its output loads into the product.
3. The branch of ``lightning_pose_amd/models/base.py:18-23`` that subclasses a real ``lightning.pytorch.LightningModule`` runs once, in a
   subprocess whose ``lightning.pytorch`` is a stand-in package with Lightning's module path: the trackers then inherit from IT, and one
   training step goes through its ``log`` / ``save_hyperparameters``."""

import os
import subprocess
import sys
import textwrap

import pytest
import torch

from oracle import ref_loader as R
from tests.conftest import ROOT, needs_reference

pytestmark = [needs_reference, pytest.mark.reference]


def _cfg(losses_to_use, HW=64, K=3, checkpoint=None, backbone="resnet50", model_type="heatmap"):
    """a reference-style config: the fields get_model reads (config_default.yaml's names), as the DictConfig stand-in the reference's modules see"""
    import omegaconf   # (ref_loader's stand-in: attribute access + .get, like DictConfig)

    return omegaconf.OmegaConf.create({
        "data": {"image_resize_dims": {"height": HW, "width": HW}, "num_keypoints": K, "downsample_factor": 2},
        "model": {"model_type": model_type, "backbone": backbone, "backbone_pretrained": False, "losses_to_use": losses_to_use,
                  "checkpoint": checkpoint},
        "training": {"rng_seed_model_pt": 5, "optimizer": "AdamW", "optimizer_params": {"learning_rate": 5e-4},
                     "lr_scheduler": "multisteplr", "lr_scheduler_params": {"multisteplr": {"milestones": [10, 20], "gamma": 0.25}}},
    })


def _ref_factory(patched: bool):
    F = R.load("models.factory")
    import lightning_pose.models as ns   # the namespace package ref_loader registered: give it the names `from lightning_pose.models import X` needs
    T = R.load("models.heatmap_tracker")
    ns.HeatmapTracker, ns.SemiSupervisedHeatmapTracker = T.HeatmapTracker, T.SemiSupervisedHeatmapTracker
    if patched:   # `if os.environ.get("LP_BACKEND") == "mi355x": return lightning_pose_amd.models.get_model_class(...)`
        from lightning_pose_amd.models import get_model_class
        return F, get_model_class
    return F, None


@pytest.fixture()
def on_emulator(stack_backend, monkeypatch):
    """the reference's get_model passes no `device=`: the product takes cuda:$LOCAL_RANK; on the CPU suite that default becomes the emulated device"""
    from lightning_pose_amd.models import heatmap_tracker as HT

    monkeypatch.setattr(HT, "_default_device", lambda: torch.device(stack_backend))
    return stack_backend


def test_reference_get_model_builds_the_product_trackers(on_emulator, monkeypatch, tmp_path):
    from lightning_pose_amd.losses import LossFactory
    from lightning_pose_amd.models import HeatmapTracker, SemiSupervisedHeatmapTracker
    from lightning_pose_amd.optim import FusedAdam

    F, switch = _ref_factory(patched=True)
    monkeypatch.setattr(F, "get_model_class", switch)
    sup = LossFactory({"heatmap_mse": {"log_weight": 0.0}}, None)
    unsup = LossFactory({"temporal": {"log_weight": 5.0, "epsilon": 2.0, "prob_threshold": 0.05}, "unimodal_mse": {"log_weight": 5.0}}, None)
    # supervised
    m = F.get_model(_cfg([]), None, {"supervised": sup, "unsupervised": None})
    assert type(m) is HeatmapTracker and m.device == torch.device(on_emulator)
    assert m.num_keypoints == 3 and m.downsample_factor == 2 and m.torch_seed == 5 and m.optimizer == "AdamW"
    assert dict(m.optimizer_params) == {"learning_rate": 5e-4} and dict(m.lr_scheduler_params) == {"milestones": [10, 20], "gamma": 0.25}
    cfgd = m.configure_optimizers()
    assert isinstance(cfgd["optimizer"], FusedAdam) and cfgd["monitor"] == "val_supervised_loss"
    assert [g["name"] for g in cfgd["optimizer"].param_groups] == ["backbone", "head"] and cfgd["optimizer"].param_groups[1]["lr"] == 5e-4
    assert cfgd["optimizer"].param_groups[1]["weight_decay"] == 0.01 and list(cfgd["lr_scheduler"].milestones) == [10, 20]
    # semi-supervised: the reference's validation walks the product's TypedDicts and loss signatures and accepts the pairing ...
    m2 = F.get_model(_cfg(["temporal", "unimodal_mse"]), None, {"supervised": sup, "unsupervised": unsup})
    assert type(m2) is SemiSupervisedHeatmapTracker and m2.loss_factory_unsup is unsup
    # ... and refuses, with ITS error, a loss whose inputs the model does not produce (reference :188-192)
    class NeedsDepth(type(sup.loss_instance_dict["heatmap_mse"])):
        def __call__(self, heatmaps_targ, heatmaps_pred, depth_maps, stage=None, **kwargs):   # noqa: D102
            raise AssertionError
    bad = LossFactory({"heatmap_mse": {"log_weight": 0.0}}, None)
    bad.loss_instance_dict["needs_depth"] = NeedsDepth()
    with pytest.raises(ValueError, match=r"loss 'needs_depth' requires \['depth_maps'\], but HeatmapTracker.get_loss_inputs_labeled\(\) produces"):
        F.get_model(_cfg([]), None, {"supervised": bad, "unsupervised": None})
    # the reference's other refusals reach the caller unchanged
    with pytest.raises(NotImplementedError):
        F.get_model(_cfg([], model_type="regression"), None, {"supervised": sup, "unsupervised": None})
    with pytest.raises(RuntimeError, match="ViT model requires"):
        c = _cfg([], backbone="vits_dino")
        c["data"]["image_resize_dims"]["width"] = 96
        F.get_model(c, None, {"supervised": sup, "unsupervised": None})
    # warm start through the reference's own checkpoint branch (:299-317): a Lightning-layout file written by the product
    from lightning_pose_amd.checkpoint import save_checkpoint

    with torch.no_grad():
        getattr(m.head.upsampling_layers, "2").bias.add_(0.25)
        m.net.refresh_weight_copies()
    ck = tmp_path / "run" / "tb_logs" / "version_0" / "checkpoints"
    ck.mkdir(parents=True)
    save_checkpoint(m, str(ck / "epoch=3-step=40.ckpt"))
    m3 = F.get_model(_cfg([], checkpoint=str(tmp_path / "run")), None, {"supervised": sup, "unsupervised": None})   # (a directory: the glob branch)
    for (k, a), (_, b) in zip(m.state_dict().items(), m3.state_dict().items()):
        assert torch.equal(a, b), k
    x = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(1)).to(on_emulator)
    m.eval(), m3.eval()
    with torch.no_grad():
        assert torch.equal(m(x), m3(x))


def test_product_checkpoint_loads_strict_into_the_verbatim_tracker(on_emulator, tmp_path):
    """N4 (the strict load was only ever checked against the restated oracle): the file, not just the dict"""
    from lightning_pose_amd.checkpoint import load_weights, save_checkpoint
    from lightning_pose_amd.models import HeatmapTracker

    F, _ = _ref_factory(patched=False)
    prod = HeatmapTracker(num_keypoints=3, loss_factory=None, backbone="resnet50", pretrained=False, torch_seed=9, device=on_emulator, precision="fp32")
    path = str(tmp_path / "model.ckpt")
    save_checkpoint(prod, path)
    ckpt = torch.load(path, weights_only=False)
    assert "state_dict" in ckpt and "hyper_parameters" in ckpt   # Lightning's layout (api/model_runtime.py:137-141 reads exactly these)
    ref = F.get_model(_cfg([]), None, {"supervised": None, "unsupervised": None})     # the UNPATCHED reference factory -> the verbatim tracker
    assert type(ref).__module__ == "lightning_pose.models.heatmap_tracker"
    missing, unexpected = ref.load_state_dict(ckpt["state_dict"], strict=True)
    assert not missing and not unexpected
    x = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(2))
    ref.eval(), prod.eval()
    with torch.no_grad():
        want, got = ref(x), prod(x.to(on_emulator)).cpu()
    torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-4 * float(want.max()))
    # the other direction, through the reference's legacy-key remap (api/model_runtime.py:143-150: old checkpoints kept the head at the top level)
    legacy = {(k[len("head."):] if k.startswith("head.upsampling_layers.") else k): v.clone() for k, v in ref.state_dict().items()}
    assert any(k.startswith("upsampling_layers.") for k in legacy)
    with torch.no_grad():
        legacy["upsampling_layers.2.bias"] += 0.5
    for key in list(legacy.keys()):   # (the remap, restated from the reference)
        if key.startswith("upsampling_layers."):
            legacy["head." + key] = legacy.pop(key)
    lpath = str(tmp_path / "legacy.ckpt")
    torch.save({"state_dict": legacy}, lpath)
    load_weights(prod, lpath)
    assert torch.allclose(getattr(prod.head.upsampling_layers, "2").bias.cpu(), ref.state_dict()["head.upsampling_layers.2.bias"] + 0.5)
    # ... and the product's own loader takes the un-remapped legacy file as well (checkpoint.py)
    torch.save({"state_dict": {(k[len("head."):] if k.startswith("head.upsampling_layers.") else k): v for k, v in ref.state_dict().items()}}, lpath)
    load_weights(prod, lpath)
    assert torch.allclose(getattr(prod.head.upsampling_layers, "2").bias.cpu(), ref.state_dict()["head.upsampling_layers.2.bias"])


_LIGHTNING_BRANCH = '''
import sys, types, torch
sys.path.insert(0, {root!r})
# a stand-in PACKAGE with Lightning's module path: models/base.py:18-23 accepts it as the real thing and subclasses it
calls = []
core = types.ModuleType("lightning.pytorch.core.module")
class LightningModule(torch.nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
        self._logged = {{}}
        self.hparams = {{}}
        self.current_epoch = self.global_step = 0
    def log(self, name, value, *a, **k):
        calls.append(("log", name, sorted(k)))
        self._logged[name] = value
    def save_hyperparameters(self, *a, ignore=None, **k):
        calls.append(("save_hyperparameters", tuple(ignore or ())))
    def optimizers(self):
        return self._opt
LightningModule.__module__ = "lightning.pytorch.core.module"
core.LightningModule = LightningModule
class Callback: pass
Callback.__module__ = "lightning.pytorch.callbacks.callback"
pkg, plm, cbm = types.ModuleType("lightning"), types.ModuleType("lightning.pytorch"), types.ModuleType("lightning.pytorch.callbacks")
plm.LightningModule, cbm.Callback, pkg.pytorch, plm.callbacks, plm.core = LightningModule, Callback, plm, cbm, core
sys.modules.update({{"lightning": pkg, "lightning.pytorch": plm, "lightning.pytorch.callbacks": cbm, "lightning.pytorch.core.module": core}})
import _lp_bootstrap
from lightning_pose_amd import _lib, ops
from tests.hipemu import emu
_lib._lib = emu.emu_lib(); ops.require_device = lambda *a: None; ops.require_device_type = lambda d: None; ops._stream = lambda: None
from lightning_pose_amd.models import base, SemiSupervisedHeatmapTracker
from lightning_pose_amd.callbacks import AnnealWeight
from lightning_pose_amd.losses import LossFactory
assert base.LightningModule is LightningModule, "the stand-in branch ran instead of the Lightning branch"
assert LightningModule in SemiSupervisedHeatmapTracker.__mro__ and issubclass(AnnealWeight, Callback)
K, HW = 3, 64
sup = LossFactory({{"heatmap_mse": {{"log_weight": 0.0}}}}, None)
unsup = LossFactory({{"temporal": {{"log_weight": 5.0, "epsilon": 1.0, "prob_threshold": 0.0}}}}, None)
m = SemiSupervisedHeatmapTracker(num_keypoints=K, loss_factory=sup, loss_factory_unsupervised=unsup, pretrained=False, torch_seed=1, device="cpu")
assert ("save_hyperparameters", ("loss_factory", "loss_factory_unsupervised")) in calls
g = torch.Generator().manual_seed(0)
kp = torch.rand(2, 2 * K, generator=g) * HW
batch = {{"labeled": {{"images": torch.randn(2, 3, HW, HW, generator=g), "keypoints": kp,
                     "heatmaps": ops.generate_heatmaps(kp.reshape(2, K, 2), HW, HW, (HW // 4, HW // 4)), "bbox": torch.tensor([[0.0, 0.0, HW, HW]]).repeat(2, 1),
                     "idxs": torch.arange(2)}},
         "unlabeled": {{"frames": torch.randn(3, 3, HW, HW, generator=g), "transforms": torch.tensor([-1.0]), "bbox": torch.tensor([[0.0, 0.0, HW, HW]]).repeat(3, 1),
                       "is_multiview": False}}}}
m.train()
m._opt = m.configure_optimizers()["optimizer"]
m._opt.zero_grad()
loss = m.training_step(batch, 0)["loss"]
loss.backward()
m._opt.step()
names = [c[1] for c in calls if c[0] == "log"]
assert names[0] == "total_unsupervised_importance" and names[-1] == "total_loss" and "train_temporal_loss" in names, names
assert all("sync_dist" in c[2] for c in calls if c[0] == "log" and c[1] not in ("total_unsupervised_importance",)), calls
assert torch.isfinite(loss).item() and float(m.net.G.abs().sum()) > 0
print("LIGHTNING_BRANCH_OK", len(names))
'''


def test_the_real_lightning_subclass_branch_runs_once():
    """INTEGRATION.md section 4 said this branch "has NEVER executed".  (Real Lightning is not installed anywhere this suite runs: the
    stand-in only has to BE a class from a `lightning.*` module for the branch to take it, which is all the branch checks.)"""
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", textwrap.dedent(_LIGHTNING_BRANCH).format(root=ROOT)], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0 and "LIGHTNING_BRANCH_OK" in out.stdout, out.stderr[-3000:]
