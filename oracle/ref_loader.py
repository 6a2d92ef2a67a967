"""Import the reference's OWN hot-path modules, verbatim, from /root/reference under stubs.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  ``/root/reference`` exists in the build container
only; on the GPU box the loader falls back to ``oracle/_ref/`` - the same hot-path files, copied
verbatim by ``oracle/make_ref.py`` (git-ignored, shipped with the snapshot).  Used by
``tests/golden/make_golden.py`` (fixture generation), by the tests that pin ``oracle.restated`` and the
product directly against the reference (skipped when neither tree is present) and by ``bench.py``'s
``cpu_baseline`` leg (the reference's own training step timed on the host cores).

Recipe: the reference package cannot be imported as a package here
(omegaconf, lightning, kornia, torchvision, jaxtyping are not installed).  Its hot-path
*modules* import unchanged once
  * bare namespace packages replace the heavy ``__init__`` files of ``lightning_pose`` and
    its ``data``/``utils``/``losses``/``models`` sub-packages,
  * tiny stand-ins exist for jaxtyping / omegaconf / lightning / the data-module classes, and
  * kornia / torchvision.models resolve to the restatements in ``oracle.thirdparty``.
No reference source is copied: modules are executed from where they lie.
"""

from __future__ import annotations

import importlib
import os
import sys
import types

import torch
import torch.nn as nn

from . import thirdparty as tp

# /root/reference in the build container; on the GPU box the verbatim copy oracle/make_ref.py shipped (oracle/_ref/: git-ignored, travels
# with the snapshot) - the hot-path modules only, which is all this loader executes
_SHIPPED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")
REFERENCE_ROOT = os.environ.get("LP_REFERENCE_ROOT") or ("/root/reference" if os.path.isdir("/root/reference/lightning_pose") else _SHIPPED)
_PKG = os.path.join(REFERENCE_ROOT, "lightning_pose")


def available() -> bool:
    return os.path.isdir(_PKG)


# ---------------------------------------------------------------------------- stubs


class _Subscriptable:
    def __class_getitem__(cls, item):
        return torch.Tensor


class _AttrDict(dict):
    """Minimal DictConfig: dict with attribute access, recursive."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return v

    def __setattr__(self, k, v):
        self[k] = v


class _ListConfig(list):
    pass


def _wrap(obj):
    if isinstance(obj, dict):
        return _AttrDict({k: _wrap(v) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)):
        return _ListConfig([_wrap(v) for v in obj])
    return obj


def _unwrap(obj):
    if isinstance(obj, dict):
        return {k: _unwrap(v) for k, v in obj.items()}
    if isinstance(obj, list):
        return [_unwrap(v) for v in obj]
    return obj


class _OmegaConf:
    @staticmethod
    def create(obj=None):
        return _wrap(obj if obj is not None else {})

    @staticmethod
    def merge(*cfgs):
        out = {}
        for c in cfgs:
            out.update(_unwrap(c))
        return _wrap(out)

    @staticmethod
    def to_object(cfg):
        return _unwrap(cfg)

    @staticmethod
    def register_new_resolver(*a, **k):
        return None


class _LightningModule(nn.Module):
    """nn.Module with the sliver of the LightningModule protocol the step touches."""

    def __init__(self, *a, **k):
        super().__init__()
        self.logged: dict[str, torch.Tensor] = {}
        self.current_epoch = 0
        self.global_step = 0

    local_rank = 0

    @property
    def device(self):
        try:
            return next(self.parameters()).device
        except StopIteration:
            return torch.device("cpu")

    def log(self, name, value, *a, **k):
        self.logged[name] = value.detach().clone() if torch.is_tensor(value) else torch.tensor(float(value))

    def save_hyperparameters(self, *a, **k):
        return None


class _Callback:
    pass


def _module(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _namespace_pkg(name: str, path: str) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__path__ = [path]  # type: ignore[attr-defined]
    m.__package__ = name
    sys.modules[name] = m
    return m


_installed = False


def install_stubs() -> None:
    """Idempotently install every stub needed to import the reference's hot-path modules."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")

    _module("jaxtyping", Float=_Subscriptable, Int=_Subscriptable, Bool=_Subscriptable,
            Shaped=_Subscriptable)
    _module("omegaconf", DictConfig=_AttrDict, ListConfig=_ListConfig, OmegaConf=_OmegaConf)

    # kornia -> restatements
    _module("kornia")
    _module("kornia.losses", kl_div_loss_2d=tp.kl_div_loss_2d, js_div_loss_2d=tp.js_div_loss_2d)
    _module("kornia.filters", filter2d=tp.filter2d)
    _module("kornia.geometry")
    _module("kornia.geometry.subpix", spatial_softmax2d=tp.spatial_softmax2d,
            spatial_expectation2d=tp.spatial_expectation2d)
    _module("kornia.geometry.transform")
    _module("kornia.geometry.transform.pyramid",
            _get_pyramid_gaussian_kernel=tp.get_pyramid_gaussian_kernel)

    # torchvision.models -> restated ResNet-50
    _module("torchvision")
    _module("torchvision.models", resnet50=tp.resnet50)

    # lightning
    lp = _module("lightning", LightningModule=_LightningModule, Trainer=object)
    plm = _module("lightning.pytorch", LightningModule=_LightningModule,
                  LightningDataModule=object, Trainer=object, Callback=_Callback)
    lp.pytorch = plm
    _module("lightning.pytorch.callbacks", Callback=_Callback, EarlyStopping=object,
            LearningRateMonitor=object, ModelCheckpoint=object)
    _module("lightning.pytorch.utilities", CombinedLoader=object)

    # namespace packages that bypass the heavy __init__ files
    _namespace_pkg("lightning_pose", _PKG)
    for sub in ("data", "utils", "losses", "models"):
        _namespace_pkg(f"lightning_pose.{sub}", os.path.join(_PKG, sub))

    # data-module classes only used for type hints / isinstance / the PCA fit driver
    class BaseDataModule:  # noqa: D401
        pass

    class UnlabeledDataModule(BaseDataModule):
        pass

    class MultiviewHeatmapDataset:
        pass

    class DataExtractor:
        def __init__(self, *a, **k):
            raise RuntimeError("oracle drives the PCA fit directly; DataExtractor is a stub")

    _module("lightning_pose.data.datamodules", BaseDataModule=BaseDataModule,
            UnlabeledDataModule=UnlabeledDataModule)
    _module("lightning_pose.data.datasets", MultiviewHeatmapDataset=MultiviewHeatmapDataset)
    _module("lightning_pose.data.extractor", DataExtractor=DataExtractor)
    _installed = True


def load(name: str) -> types.ModuleType:
    """``load('losses.losses')`` -> the reference's lightning_pose/losses/losses.py, unchanged."""
    install_stubs()
    return importlib.import_module(f"lightning_pose.{name}")


def fit_keypoint_pca(loss_type: str, data_arr: torch.Tensor, *, components_to_keep=0.99,
                     empirical_epsilon_percentile: float = 99.0, mirrored_column_matches=None,
                     columns_for_singleview_pca=None, centering_method=None):
    """Drive the reference's KeypointPCA fit on an in-memory (N, 2K) array.

    Mirrors ``KeypointPCA.__call__`` (lightning_pose/utils/pca.py:311-328) minus ``_get_data``
    (which needs a real data module); every other step runs the reference's own code.
    """
    pca_mod = load("utils.pca")
    dm = sys.modules["lightning_pose.data.datamodules"].BaseDataModule()
    kp = pca_mod.KeypointPCA(
        loss_type=loss_type, data_module=dm, components_to_keep=components_to_keep,
        empirical_epsilon_percentile=empirical_epsilon_percentile,
        mirrored_column_matches=mirrored_column_matches,
        columns_for_singleview_pca=columns_for_singleview_pca, device="cpu",
        centering_method=centering_method,
    )
    kp.data_arr = kp._format_data(data_arr=data_arr.clone())
    kp._check_data()
    kp._fit_pca()
    kp._choose_n_components()
    kp._set_parameter_dict()
    return kp
