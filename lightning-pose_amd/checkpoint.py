"""Checkpoint interchange with the reference.

The reference trains under Lightning, whose ``.ckpt`` is a ``torch.save``d dict: ``state_dict`` (keys ``backbone.{0,1,4..7}.*`` /
``backbone.vision_encoder.*`` and ``head.upsampling_layers.{1,2}.{weight,bias}``), ``hyper_parameters`` (the constructor arguments
kept by ``save_hyperparameters``), ``epoch``, ``global_step``, ``pytorch-lightning_version``, optimizer / scheduler states.  The
trackers here expose exactly those parameter names and shapes, so weights move in both directions without renaming:

* ``save_checkpoint``            writes a file ``lightning_pose.api.model_runtime.load_model_from_checkpoint`` (:79-187) and
                                 ``models/factory.py:299-317`` accept (they read ``state_dict`` and ``hyper_parameters`` only);
* ``read_state_dict``            reads a reference checkpoint, with the legacy ``upsampling_layers.*`` key remap of
                                 ``api/model_runtime.py:143-148``;
* ``load_weights``               ``cfg.model.checkpoint`` semantics of ``models/factory.py:299-317``: a ``.ckpt`` file or a directory
                                 searched recursively, ``strict=False``, and - when shapes disagree (e.g. a different number of
                                 keypoints) - the backbone-only retry;
* ``load_model_from_checkpoint`` rebuilds a tracker from ``hyper_parameters`` like ``ModelClass.load_from_checkpoint``.

Everything here is host-side file handling; tensors land in the engine's device-resident flat buffers through the model's own
``load_state_dict``.
"""

from __future__ import annotations

import glob
import os
from collections import OrderedDict
from typing import Any

import torch

LIGHTNING_VERSION = "2.5.0"  # the reference pins lightning ~=2.5.0 (pyproject.toml); readers only look at the keys below
_PLAIN = (int, float, str, bool, type(None))


def _plain(v: Any) -> Any:
    """hyper-parameters as plain python (what ``torch.load(weights_only=True)`` can read back)"""
    if isinstance(v, _PLAIN):
        return v
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    if hasattr(v, "items"):
        return {str(k): _plain(x) for k, x in v.items()}
    if isinstance(v, torch.device):
        return str(v)
    return str(v)


def save_checkpoint(model, path: str, optimizer=None, epoch: int | None = None, global_step: int | None = None) -> str:
    sd = OrderedDict((k, v.detach().to("cpu").contiguous().clone()) for k, v in model.state_dict().items())
    hp = {k: _plain(v) for k, v in dict(getattr(model, "hparams", {}) or {}).items() if k != "device"}
    ckpt = {
        "epoch": int(model.current_epoch if epoch is None else epoch),
        "global_step": int(model.global_step if global_step is None else global_step),
        "pytorch-lightning_version": LIGHTNING_VERSION,
        "state_dict": sd,
        "hyper_parameters": hp,
        "hparams_name": "kwargs",
    }
    if optimizer is not None:  # this optimizer's state is one flat (exp_avg, exp_avg_sq) pair, not Lightning's per-parameter list
        ckpt["lp_amd_optimizer_state"] = optimizer.state_dict()
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(ckpt, path)
    return path


def _load_file(path: str) -> dict:
    try:
        return torch.load(path, map_location="cpu")
    except Exception:  # noqa: BLE001 - older checkpoints pickle omegaconf containers etc. (reference model_runtime.py:131-136)
        return torch.load(path, map_location="cpu", weights_only=False)


def resolve_checkpoint_path(ckpt: str) -> str:
    """a ``.ckpt`` file, or a directory searched recursively for one (reference models/factory.py:302-303)"""
    if ckpt.endswith(".ckpt"):
        return ckpt
    found = sorted(glob.glob(os.path.join(ckpt, "**", "*.ckpt"), recursive=True))
    if not found:
        raise FileNotFoundError(f"no .ckpt file under {ckpt}")
    return found[0]


def read_checkpoint(path: str) -> dict:
    ckpt = _load_file(resolve_checkpoint_path(path))
    if "state_dict" not in ckpt:  # a bare state_dict
        ckpt = {"state_dict": ckpt}
    sd = ckpt["state_dict"]
    for key in list(sd.keys()):  # checkpoints from before the head was its own module
        if key.startswith("upsampling_layers."):
            sd["head." + key] = sd.pop(key)
    return ckpt


def read_state_dict(path: str) -> dict[str, torch.Tensor]:
    return read_checkpoint(path)["state_dict"]


def load_weights(model, ckpt: str):
    """``cfg.model.checkpoint``: non-strict load; on a shape clash keep the checkpoint's backbone only (factory.py:310-317)."""
    state_dict = read_state_dict(ckpt)
    try:
        return model.load_state_dict(state_dict, strict=False)
    except RuntimeError:
        backbone_only = OrderedDict((k, v) for k, v in state_dict.items() if "backbone" in k)
        return model.load_state_dict(backbone_only, strict=False)


def load_model_from_checkpoint(ckpt_file: str | None, model_class=None, strict: bool = False, **overrides: Any):
    """Rebuild a tracker from a checkpoint's ``hyper_parameters`` (+ ``overrides``: loss factories, device, ...) and load its weights."""
    if ckpt_file is None:
        raise ValueError("ckpt_file must be provided to load a model from checkpoint")
    ckpt = read_checkpoint(ckpt_file)
    hp = dict(ckpt.get("hyper_parameters", {}))
    hp.update(overrides)
    hp["pretrained"] = False  # the weights come from the checkpoint
    hp.pop("backbone_checkpoint", None)
    if model_class is None:
        from .models import HeatmapTracker, SemiSupervisedHeatmapTracker
        model_class = SemiSupervisedHeatmapTracker if hp.get("loss_factory_unsupervised") is not None else HeatmapTracker
    model = model_class(**hp)
    model.load_state_dict(ckpt["state_dict"], strict=strict)
    model.current_epoch = int(ckpt.get("epoch", 0))
    model.global_step = int(ckpt.get("global_step", 0))
    return model
