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
    # Nothing in a step makes the host wait for the device, so unthrottled it would enqueue steps as fast as the launch queue accepts them,
    # each with its own ~30 GB of activations allocated ahead of time.  Two steps in flight keep the device busy across the step boundary
    # and bound the memory the caching allocator has to hold.
    MAX_STEPS_IN_FLIGHT = int(__import__("os").environ.get("LP_MAX_STEPS_IN_FLIGHT", "2"))

    def __init__(self, max_epochs: int = 1, callbacks: list | None = None, limit_train_batches: int | None = None,
                 data_parallel: bool | None = None, sync_batchnorm: bool = True, accumulate_grad_batches: int = 1,
                 log_every_n_steps: int = 50):
        self.max_epochs = max_epochs
        self.callbacks = callbacks or []
        self.limit_train_batches = limit_train_batches
        self.sync_batchnorm = sync_batchnorm
        self.accumulate_grad_batches = accumulate_grad_batches
        self._want_dp = data_parallel
        # The step's logged scalars stay ON THE DEVICE; they reach the host every log_every_n_steps optimiser steps (Lightning's default 50;
        # the reference passes cfg.training.log_every_n_steps, train.py:420) and at the end of every epoch, as ONE packed copy.  Round 3
        # called float() on each of ~15 scalars after every batch: a full host <-> device synchronisation per step, which made
        # MAX_STEPS_IN_FLIGHT meaningless for anyone who trained through fit() instead of training_batch().
        self.log_every_n_steps = max(1, int(log_every_n_steps))
        self._inflight: list = []             # "step finished" events: the host stays at most MAX_STEPS_IN_FLIGHT steps ahead of the device
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
        elif getattr(self, "scheduler", None) is None:  # a model whose optimiser was configured by the caller: its scheduler only
            self.scheduler = model.get_scheduler(model.optimizers())
        opt = model.optimizers()
        if self.dp is not None:
            opt.grad_scale = 1.0 / self.dp.world

    def _sync_logged(self, model) -> None:
        """``self.log(..., sync_dist=True)``: every marked scalar of the step becomes its mean over the ranks - ONE packed all-reduce"""
        names = getattr(model, "sync_logged", None)
        if self.dp is not None and self.dp.active and names:
            model.logged.update(self.dp.mean_scalars({k: model.logged[k] for k in names if k in model.logged}))

    def training_batch(self, model, batch: dict, batch_idx: int, last_in_epoch: bool = False) -> torch.Tensor:
        """One optimisation step; returns the (detached) loss.  Every kernel is enqueued from here; nothing in it makes the host wait for the
        device.  (Rounds 2 - 3 could also replay the step as one captured HIP graph: host 0.2 instead of 10 - 26 ms per step, but the replay
        was ~3 ms SLOWER on the device than stream launches and the device bounds the step - measured in profiles/archive/r02d_*, removed in round 4,
        profiles/retired/r04_graph_step.py.txt.)"""
        return self._eager_batch(model, batch, batch_idx, last_in_epoch)

    def _eager_batch(self, model, batch: dict, batch_idx: int, last_in_epoch: bool = False, count: bool = True) -> torch.Tensor:
        """With gradient accumulation the loss is scaled by 1 / accumulate_grad_batches (Lightning's rule) and a trailing partial group is
        stepped at the end of the epoch (``last_in_epoch``).  ``count=False``: hooks and step counters are the caller's."""
        opt = model.optimizers()
        acc = self.accumulate_grad_batches
        if count:
            self._hook("on_train_batch_start", model, batch, batch_idx)
        if self.dp is not None and not getattr(self, "_batches_checked", False):
            self._batches_checked = True
            lab = batch["labeled"] if "labeled" in batch else batch
            sizes = [int(lab["images"].shape[0])] + ([int(batch["unlabeled"]["frames"].shape[0])] if "unlabeled" in batch else [])
            self.dp.check_equal_batches(*sizes)
        if batch_idx % acc == 0:
            opt.zero_grad()
        loss = model.training_step(batch, batch_idx)["loss"]
        if self.dp is not None and acc == 1:
            self.dp.begin_step()  # buckets leave as soon as backward has finished their layers (single-backward steps only)
        (loss / acc if acc > 1 else loss).backward()
        if (batch_idx + 1) % acc == 0 or last_in_epoch:
            if self.dp is not None:
                # the gradient buckets go out first (tail of the flat buffer = what backward finished first); the logged scalars' mean
                # rides behind them on the same communicator, and the optimiser waits for the buckets only
                self.dp.all_reduce_gradients()
                self._sync_logged(model)
                self.dp.wait()
            opt.step()
            if count:
                model.global_step += 1
        elif self.dp is not None:
            self._sync_logged(model)
        if count:   # Lightning's hook order: after the optimiser step of the batch (callbacks.Callback.on_train_batch_end(trainer, module, outputs, batch, idx))
            self._hook("on_train_batch_end", model, {"loss": loss.detach()}, batch, batch_idx)
        if count and model.device.type == "cuda":
            ev = torch.cuda.Event()
            ev.record()
            self._inflight.append(ev)
            if len(self._inflight) > self.MAX_STEPS_IN_FLIGHT:
                self._inflight.pop(0).synchronize()
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
                self._sync_logged(model)  # val_supervised_loss (the scheduler / checkpoint monitor) is the mean over ranks
                for k, v in model.logged.items():
                    totals[k] = totals.get(k, 0.0) + float(v)
                n += 1
        finally:
            model.train(was_training)
        means = {k: v / max(n, 1) for k, v in totals.items()}
        self.validation_history.append(means)
        return means

    def _flush_logged(self, model) -> None:
        """the current step's logged values -> one host record: device scalars travel as ONE stacked copy (the only synchronisation)"""
        logged = getattr(model, "logged", {})
        if not logged:
            return
        dev_keys = [k for k, v in logged.items() if torch.is_tensor(v) and v.device.type != "cpu"]
        rec = {k: float(v) for k, v in logged.items() if k not in dev_keys}
        if dev_keys:
            vals = torch.stack([logged[k].detach().float().reshape(()) for k in dev_keys]).tolist()
            rec.update(zip(dev_keys, vals))
        rec["step"] = float(model.global_step)
        self.logged_history.append({k: rec[k] for k in list(logged) + ["step"]})

    def fit(self, model, batches: Callable[[int], Iterable[dict]] | Iterable[dict],
            val_batches: Callable[[int], Iterable[dict]] | None = None) -> None:
        self.setup(model)
        model.train()
        self._hook("on_train_start", model)
        for epoch in range(self.max_epochs):
            model.current_epoch = epoch
            self._hook("on_train_epoch_start", model)
            it = iter(batches(epoch) if callable(batches) else batches)
            nxt = next(it, None)
            if self.limit_train_batches is not None and self.limit_train_batches <= 0:
                nxt = None   # Lightning: limit_train_batches = 0 means no training batches (and no optimiser step)
            batch_idx = 0
            while nxt is not None:  # one batch of look-ahead: the last batch of the epoch closes a partial accumulation group
                batch, nxt = nxt, next(it, None)
                last = nxt is None or (self.limit_train_batches is not None and batch_idx + 1 >= self.limit_train_batches)
                step_before = model.global_step
                self.training_batch(model, batch, batch_idx, last_in_epoch=last)
                # flush when an optimiser step HAPPENED in this batch and landed on a multiple of N (with accumulate_grad_batches > 1 the
                # step counter stands still on the non-stepping micro-batches: testing only its value flushed - and synchronised - on every
                # one of them while it sat on a multiple, and appended duplicate records), and at the end of the epoch
                stepped = model.global_step != step_before
                if last or (stepped and model.global_step % self.log_every_n_steps == 0):
                    self._flush_logged(model)
                batch_idx += 1
                if last:
                    break
            if val_batches is not None:  # check_val_every_n_epoch = 1
                self.validate(model, val_batches(epoch))
            self.scheduler.step()
