"""Minimal fit loop with Lightning's hook order for the heatmap-tracker step (Lightning is not a dependency here).

Order per batch (reference: Lightning 2.5 fit loop as used by train.py:411-431): callbacks.on_train_batch_start ->
training_step -> backward -> gradient all-reduce -> optimizer.step -> zero_grad; per epoch: on_train_epoch_start ...
lr_scheduler.step().  The module is anything with the LightningModule surface of models/base.py.
"""

from __future__ import annotations

from typing import Any, Callable, Iterable

import torch

from .distributed import DataParallel


class Trainer:
    def __init__(self, max_epochs: int = 1, callbacks: list | None = None, limit_train_batches: int | None = None,
                 data_parallel: bool | None = None, sync_batchnorm: bool = True, accumulate_grad_batches: int = 1):
        self.max_epochs = max_epochs
        self.callbacks = callbacks or []
        self.limit_train_batches = limit_train_batches
        self.sync_batchnorm = sync_batchnorm
        self.accumulate_grad_batches = accumulate_grad_batches
        self._want_dp = data_parallel
        self.dp: DataParallel | None = None
        self.logged_history: list[dict[str, float]] = []
        self.validation_history: list[dict[str, float]] = []

    def _hook(self, name: str, *args: Any) -> None:
        for cb in self.callbacks:
            fn = getattr(cb, name, None)
            if fn is not None:
                fn(self, *args)

    def setup(self, model) -> None:
        import torch.distributed as dist

        if self.dp is None:
            want = self._want_dp if self._want_dp is not None else dist.is_initialized()
            self.dp = DataParallel(model.net, sync_bn=self.sync_batchnorm) if want else None
            if self.dp is not None:
                self.dp.broadcast_parameters()
        if model.optimizers() is None:
            cfg = model.configure_optimizers()
            self.scheduler = cfg["lr_scheduler"]
        opt = model.optimizers()
        if self.dp is not None:
            opt.grad_scale = 1.0 / self.dp.world

    def training_batch(self, model, batch: dict, batch_idx: int) -> torch.Tensor:
        """One optimisation step; returns the (detached) loss."""
        opt = model.optimizers()
        self._hook("on_train_batch_start", model, batch, batch_idx)
        if batch_idx % self.accumulate_grad_batches == 0:
            opt.zero_grad()
        loss = model.training_step(batch, batch_idx)["loss"]
        loss.backward()
        if (batch_idx + 1) % self.accumulate_grad_batches == 0:
            if self.dp is not None:
                self.dp.all_reduce_gradients()
                self.dp.wait()
            opt.step()
            model.global_step += 1
        return loss.detach()

    @torch.no_grad()
    def validate(self, model, batches: Iterable[dict], step_name: str = "validation_step") -> dict[str, float]:
        """Lightning's validation / test loop for this module: eval mode (BatchNorm running statistics - the folded inference forward),
        no autograd tape, ``validation_step`` per batch, every logged value averaged over the batches (what ``self.log(on_epoch=True)``
        reports, e.g. the ``val_supervised_loss`` the scheduler / checkpointing monitor)."""
        was_training = model.training
        model.eval()
        totals: dict[str, float] = {}
        n = 0
        try:
            for batch_idx, batch in enumerate(batches):
                model.logged = {}
                getattr(model, step_name)(batch, batch_idx)
                for k, v in model.logged.items():
                    totals[k] = totals.get(k, 0.0) + float(v)
                n += 1
        finally:
            model.train(was_training)
        means = {k: v / max(n, 1) for k, v in totals.items()}
        self.validation_history.append(means)
        return means

    def fit(self, model, batches: Callable[[int], Iterable[dict]] | Iterable[dict],
            val_batches: Callable[[int], Iterable[dict]] | None = None) -> None:
        self.setup(model)
        model.train()
        self._hook("on_train_start", model)
        for epoch in range(self.max_epochs):
            model.current_epoch = epoch
            self._hook("on_train_epoch_start", model)
            it = batches(epoch) if callable(batches) else batches
            for batch_idx, batch in enumerate(it):
                if self.limit_train_batches is not None and batch_idx >= self.limit_train_batches:
                    break
                self.training_batch(model, batch, batch_idx)
                self.logged_history.append({k: float(v) for k, v in getattr(model, "logged", {}).items()})
            if val_batches is not None:  # check_val_every_n_epoch = 1
                self.validate(model, val_batches(epoch))
            self.scheduler.step()
