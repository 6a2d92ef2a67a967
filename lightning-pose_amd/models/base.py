"""LightningModule-protocol bases of the trackers (reference: lightning_pose/models/base.py).

``lightning`` is imported when available so the module plugs straight into ``pl.Trainer``; otherwise a minimal stand-in
implementing the slice of the protocol the step touches (``log``, ``save_hyperparameters``, ``device``, ``local_rank``,
``current_epoch``, ``global_step``, ``optimizers``) is used together with ``lightning_pose_amd.trainer.Trainer``.
"""

from __future__ import annotations

import inspect
import os
from typing import Any, Literal

import torch
from torch import nn
from torch.optim.lr_scheduler import MultiStepLR

try:  # pragma: no cover - lightning is not installed in the build image
    from lightning.pytorch import LightningModule  # type: ignore

    if not str(getattr(LightningModule, "__module__", "")).startswith(("lightning", "pytorch_lightning")):
        raise ImportError("sys.modules['lightning.pytorch'] is a stand-in registered by other code, not Lightning")
except Exception:  # noqa: BLE001

    class LightningModule(nn.Module):  # type: ignore[no-redef]
        def __init__(self, *args: Any, **kwargs: Any) -> None:
            super().__init__()
            self.logged: dict[str, torch.Tensor] = {}
            self.sync_logged: set[str] = set()   # names logged with sync_dist=True: the trainer averages them across ranks
            self.current_epoch = 0
            self.global_step = 0
            self._optimizer = None
            self.hparams: dict[str, Any] = {}

        @property
        def local_rank(self) -> int:
            return int(os.environ.get("LOCAL_RANK", "0"))

        @property
        def device(self) -> torch.device:
            for p in self.parameters():
                return p.device
            return torch.device("cpu")

        def log(self, name: str, value: Any, *args: Any, sync_dist: bool = False, **kwargs: Any) -> None:
            """Lightning's ``self.log``: the value is recorded; ``sync_dist=True`` (reference models/base.py:531-544,651-656,697) marks it
            for the cross-rank mean, which the trainer takes for ALL marked scalars of a step in one packed all-reduce
            (``DataParallel.mean_scalars``) instead of one collective per scalar."""
            self.logged[name] = value.detach() if torch.is_tensor(value) else torch.tensor(float(value))
            if sync_dist:
                self.sync_logged.add(name)

        def save_hyperparameters(self, *args: Any, ignore: list[str] | None = None, **kwargs: Any) -> None:
            """Keep the calling ``__init__``'s arguments in ``self.hparams`` (what Lightning stores as ``hyper_parameters``)."""
            frame = inspect.currentframe().f_back
            info = inspect.getargvalues(frame)
            hp = {n: info.locals[n] for n in info.args if n != "self"}
            if info.keywords:
                hp.update(info.locals[info.keywords])
            for n in ignore or []:
                hp.pop(n, None)
            self.hparams.update(hp)

        def optimizers(self):
            return self._optimizer


DEFAULT_LR_SCHEDULER_PARAMS = {"milestones": [150, 200, 250], "gamma": 0.5}
DEFAULT_OPTIMIZER_PARAMS = {"learning_rate": 1e-3}


class LrNotImplementedError(NotImplementedError):
    def __init__(self, lr_scheduler: str) -> None:
        super().__init__(f"'{lr_scheduler}' is an invalid LR scheduler. Must be multisteplr.")


class OptimizerNotImplementedError(NotImplementedError):
    def __init__(self, optimizer: str) -> None:
        super().__init__(f"'{optimizer}' is an invalid optimizer. Must be Adam or AdamW.")


def check_if_semi_supervised(losses_to_use: list | None = None) -> bool:
    """reference :46-62"""
    if losses_to_use is None or len(losses_to_use) == 0:
        return False
    if len(losses_to_use) == 1 and losses_to_use[0] == "":
        return False
    return True


def _merged(defaults: dict, user: Any) -> dict:
    out = dict(defaults)
    if user is not None:
        out.update({k: user[k] for k in user})
    return out


class BaseSupervisedTracker(LightningModule):
    """Optimiser / scheduler plumbing + labeled evaluation (reference :199-479, :482-599)."""

    loss_factory: Any
    rmse_loss: Any

    def __init__(self, optimizer: str = "Adam", optimizer_params: Any = None, lr_scheduler: str = "multisteplr",
                 lr_scheduler_params: Any = None, **kwargs: Any) -> None:
        super().__init__()
        if lr_scheduler not in ("multistep_lr", "multisteplr"):
            raise LrNotImplementedError(lr_scheduler)
        if optimizer not in ("Adam", "AdamW"):
            raise OptimizerNotImplementedError(optimizer)
        self.lr_scheduler = lr_scheduler
        self.lr_scheduler_params = _merged(DEFAULT_LR_SCHEDULER_PARAMS, lr_scheduler_params)
        self.optimizer = optimizer
        self.optimizer_params = _merged(DEFAULT_OPTIMIZER_PARAMS, optimizer_params)

    # -- optimiser --------------------------------------------------------------------------------------------
    def get_parameters(self) -> list[dict]:
        raise NotImplementedError

    def get_scheduler(self, optimizer: torch.optim.Optimizer) -> MultiStepLR:
        return MultiStepLR(optimizer, milestones=list(self.lr_scheduler_params["milestones"]), gamma=self.lr_scheduler_params["gamma"])

    def configure_optimizers(self) -> dict:
        from ..optim import FusedAdam

        lr = float(self.optimizer_params["learning_rate"])
        if self.optimizer == "Adam":
            optimizer = FusedAdam(self.net, self.get_parameters(), lr=lr)
        else:  # AdamW: torch default weight_decay = 0.01, decoupled
            optimizer = FusedAdam(self.net, self.get_parameters(), lr=lr, weight_decay=0.01, decoupled_weight_decay=True)
        self._optimizer = optimizer
        return {"optimizer": optimizer, "lr_scheduler": self.get_scheduler(optimizer), "monitor": "val_supervised_loss"}

    # -- labeled step -------------------------------------------------------------------------------------------
    def get_loss_inputs_labeled(self, batch_dict: dict) -> dict:
        raise NotImplementedError

    def evaluate_labeled(self, batch_dict: dict, stage: Literal["train", "val", "test"] | None = None,
                         anneal_weight: torch.Tensor | float | None = None) -> torch.Tensor:
        data_dict = self.get_loss_inputs_labeled(batch_dict=batch_dict)
        loss, log_list = self.loss_factory(stage=stage, anneal_weight=anneal_weight, **data_dict)
        loss_rmse, _ = self.rmse_loss(stage=stage, **data_dict)
        if stage:
            self.log(f"{stage}_supervised_loss", loss, prog_bar=True, sync_dist=True)
            self.log(f"{stage}_supervised_rmse", loss_rmse, sync_dist=True)
            for log_dict in log_list:
                self.log(log_dict["name"], log_dict["value"], prog_bar=log_dict.get("prog_bar", False), sync_dist=True)
        return loss

    def training_step(self, batch_dict: dict, batch_idx: int) -> dict[str, torch.Tensor]:
        if hasattr(self, "total_unsupervised_importance"):
            anneal_weight = self.total_unsupervised_importance
            self.log("total_unsupervised_importance", anneal_weight, prog_bar=True)
        else:
            anneal_weight = None
        if getattr(self, "net", None) is not None:
            self.net.single_backward = True  # one pass, one backward: gradient buckets may leave while it runs
        return {"loss": self.evaluate_labeled(batch_dict, "train", anneal_weight=anneal_weight)}

    def validation_step(self, batch_dict: dict, batch_idx: int) -> None:
        self.evaluate_labeled(batch_dict, "val")

    def test_step(self, batch_dict: dict, batch_idx: int) -> None:
        self.evaluate_labeled(batch_dict, "test")


class SemiSupervisedTrackerMixin:
    """training_step = supervised + unsupervised losses (reference :602-701)."""

    loss_factory_unsup: Any
    total_unsupervised_importance: torch.Tensor

    def get_loss_inputs_unlabeled(self, batch_dict: dict) -> dict:
        raise NotImplementedError

    def evaluate_unlabeled(self, batch_dict: dict, stage: Literal["train", "val", "test"] | None = None,
                           anneal_weight: float | torch.Tensor = 1.0) -> torch.Tensor:
        data_dict = self.get_loss_inputs_unlabeled(batch_dict=batch_dict)
        loss, log_list = self.loss_factory_unsup(stage=stage, anneal_weight=anneal_weight, **data_dict)
        if stage:
            for log_dict in log_list:
                self.log(log_dict["name"], log_dict["value"], prog_bar=log_dict.get("prog_bar", False), sync_dist=True)
        return loss

    def training_step(self, batch_dict: dict, batch_idx: int) -> dict[str, torch.Tensor]:
        unsup_importance = self.total_unsupervised_importance
        self.log("total_unsupervised_importance", unsup_importance, prog_bar=True)
        # both batches share ONE pass through the network (two BatchNorm segments per launch - same statistics, running-statistics
        # order and gradients as the reference's two forward calls, half the launches, fuller tile rounds); the two evaluate_* calls
        # below then pick up their heat-maps instead of running the network.  (An earlier opt-in mode ran the two passes on two
        # streams: +5.5 % on the device against +4.8 % / +14 % at 384 / 256 px for this one, which also survives SyncBatchNorm.)
        joint = getattr(self, "joint_forward", None)
        try:
            joined = joint is not None and bool(joint(batch_dict["labeled"]["images"], batch_dict["unlabeled"]["frames"]))
            if getattr(self, "net", None) is not None:
                self.net.single_backward = joined  # two separate passes accumulate into G twice: nothing may be sent before both ran
            loss_super = self.evaluate_labeled(batch_dict=batch_dict["labeled"], stage="train", anneal_weight=unsup_importance)
            loss_unsuper = self.evaluate_unlabeled(batch_dict=batch_dict["unlabeled"], stage="train", anneal_weight=unsup_importance)
        finally:
            if joint is not None:
                self._joint = {}
        total_loss = loss_super + loss_unsuper
        self.log("total_loss", total_loss, prog_bar=True, sync_dist=True)
        return {"loss": total_loss}
