"""AnnealWeight / UnfreezeBackbone (reference: lightning_pose/callbacks.py:32-196): the two callbacks that mutate
hot-path state.  Same constructor arguments and hook names, so they work under ``pl.Trainer`` and under
``lightning_pose_amd.trainer.Trainer``."""

from __future__ import annotations

from typing import Any

import torch

try:  # pragma: no cover
    from lightning.pytorch.callbacks import Callback  # type: ignore
except Exception:  # noqa: BLE001
    class Callback:  # type: ignore[no-redef]
        pass


class AnnealWeight(Callback):
    """Linearly raise ``pl_module.<attr_name>`` from init_val to final_val, one increment per epoch after the freeze."""

    def __init__(self, attr_name: str, init_val: float = 0.0, increase_factor: float = 0.01, final_val: float = 1.0,
                 freeze_until_epoch: int = 0) -> None:
        super().__init__()
        self.attr_name, self.init_val, self.increase_factor = attr_name, init_val, increase_factor
        self.final_val, self.freeze_until_epoch = final_val, freeze_until_epoch

    def on_train_start(self, trainer: Any, pl_module: Any) -> None:
        setattr(pl_module, self.attr_name, torch.tensor(self.init_val))

    def on_train_epoch_start(self, trainer: Any, pl_module: Any) -> None:
        if pl_module.current_epoch > self.freeze_until_epoch:
            eff_epoch = pl_module.current_epoch - self.freeze_until_epoch
            setattr(pl_module, self.attr_name, torch.tensor(min(self.init_val + eff_epoch * self.increase_factor, self.final_val)))


class UnfreezeBackbone(Callback):
    """Backbone lr: 0 until the unfreeze epoch/step, then initial_ratio * head_lr, then x warm_up_ratio per epoch/step until
    it reaches the head lr.  Needs optimizer.param_groups == [backbone, head] (reference :79-196)."""

    def __init__(self, unfreeze_epoch: int | None = None, unfreeze_step: int | None = None, initial_ratio: float = 0.1,
                 warm_up_ratio: float = 1.5) -> None:
        assert (unfreeze_epoch is None) != (unfreeze_step is None), "Exactly one must be provided."
        self.unfreeze_epoch, self.unfreeze_step = unfreeze_epoch, unfreeze_step
        self.initial_ratio, self.warm_up_ratio = initial_ratio, warm_up_ratio
        self._warmed_up = False
        self._initial_lr = 0.0

    def on_train_batch_start(self, trainer: Any, pl_module: Any, batch: Any, batch_idx: int) -> None:
        if self._warmed_up:
            return
        optimizer = pl_module.optimizers()
        assert optimizer.param_groups[0]["name"] == "backbone"
        head_lr = optimizer.param_groups[1]["lr"]
        optimizer.param_groups[0]["lr"] = self._get_backbone_lr(pl_module.global_step, pl_module.current_epoch, head_lr)

    def _get_backbone_lr(self, current_step: int | None, current_epoch: int, upsampling_lr: float) -> float:
        assert not self._warmed_up
        thaw, now = (self.unfreeze_step, current_step) if self.unfreeze_step is not None else (self.unfreeze_epoch, current_epoch)
        if now < thaw:
            return 0.0
        if now == thaw:
            self._initial_lr = self.initial_ratio * upsampling_lr
            return self._initial_lr
        next_lr = min(self._initial_lr * self.warm_up_ratio ** (now - thaw), upsampling_lr)
        if next_lr == upsampling_lr:
            self._warmed_up = True
        return next_lr
