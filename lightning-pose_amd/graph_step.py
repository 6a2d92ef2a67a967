"""One optimisation step as ONE captured HIP graph.

A step of the heatmap tracker is ~1000 kernel launches issued from Python through ctypes (26 ms of host time per step at 384 px, and the
bound of the whole step at 256 px, where the device needs only ~21 ms).  The network is static, every buffer a step touches can be
allocated once, and no value computed on the device steers the host - so after a few eager steps the whole thing (zero_grad -> joint
forward -> decode -> losses -> backward incl. the side-stream weight gradients -> fused Adam) is captured with ``torch.cuda.graph`` (a
hipGraph on ROCm: torch supplies the capture stream and the private memory pool, the nodes are the lp_hip kernels) and replayed.  Per
step the host then copies the new batch into the captured input buffers, rewrites 32 bytes of optimiser scalars (FusedAdam.advance) and
launches the graph.

What is baked into a capture is its KEY: batch tensor shapes, which parameter groups have a non-zero learning rate (they refresh their
data-gradient weight copies), the anneal weight of the unsupervised losses, train/eval mode.  When the key changes (an epoch boundary:
AnnealWeight / UnfreezeBackbone / MultiStepLR - reference callbacks.py:32-196) the step is captured again.  Learning rates themselves
and Adam's step count are NOT baked in: they are read from device memory (lp_adam_step_dev).

Opt-in: ``Trainer(hip_graph=True)`` or ``LP_HIP_GRAPH=1``.  Single-process only by default; with a process group ``LP_HIP_GRAPH_DIST=1``
captures the collectives too (verified on RCCL 2.26 in loop-back on one MI355X; never run across GPUs).  Never while
bench.py's per-launch events are on."""

from __future__ import annotations

import os

import torch


def requested() -> bool:
    return os.environ.get("LP_HIP_GRAPH", "0") == "1"


def _tensors(batch, prefix=""):
    out = []
    for k, v in batch.items():
        if isinstance(v, dict):
            out += _tensors(v, prefix + k + ".")
        elif torch.is_tensor(v):
            out.append((prefix + k, v))
    return out


class GraphedStep:
    WARMUP = 2  # eager steps before the capture (lazy allocations: workspaces, side stream, decode tables, the optimiser's buffers)

    def __init__(self, trainer, model) -> None:
        self.trainer, self.model = trainer, model
        self.graph: torch.cuda.CUDAGraph | None = None
        self.key = None
        self.static: dict | None = None
        self.loss: torch.Tensor | None = None
        self.eager_steps = 0
        self.captures = 0
        self.replays = 0

    def _key(self, batch) -> tuple:
        opt = self.model.optimizers()
        shapes = tuple((n, tuple(t.shape), str(t.dtype), str(t.device)) for n, t in _tensors(batch))
        anneal = getattr(self.model, "total_unsupervised_importance", None)
        return (shapes, opt.refresh_signature(), None if anneal is None else float(anneal), bool(self.model.training),
                tuple(sorted((k, str(v)) for k, v in _flat_scalars(batch))))

    def _adopt(self, batch) -> dict:
        """The captured graph reads its inputs from PRIVATE copies of the first batch's tensors: _load overwrites them with every later
        batch, and a caller may hand the same batch objects in again next epoch (a list of batches reused every epoch, bench.py) - writing
        through the caller's own tensors would silently replace batch 0's data with the last batch's."""
        def clone(d):
            return {k: (clone(v) if isinstance(v, dict) else v.clone() if torch.is_tensor(v) else v) for k, v in d.items()}
        return clone(batch)

    def _load(self, batch) -> None:
        for (n, dst), (_, src) in zip(_tensors(self.static), _tensors(batch)):
            dst.copy_(src, non_blocking=True)   # (one batch of D2D copies per step: negligible next to the step)

    def step(self, batch: dict, batch_idx: int) -> torch.Tensor:
        trainer, model = self.trainer, self.model
        opt = model.optimizers()
        if self.eager_steps < self.WARMUP:
            self.eager_steps += 1
            return trainer._eager_batch(model, batch, batch_idx)
        trainer._hook("on_train_batch_start", model, batch, batch_idx)  # (UnfreezeBackbone moves the learning rates here)
        key = self._key(batch)
        if self.graph is None or key != self.key:
            opt.enable_device_hyper()
            self.static = self._adopt(batch)
            net = model.net
            nbt0 = int(net.nbt) if hasattr(net, "nbt") else 0
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            # with a process group alive, ProcessGroupNCCL's watchdog thread polls events while this thread captures: only THIS thread's
            # calls are part of the capture ("thread_local"); the default mode would fail the capture on the watchdog's query
            mode = "thread_local" if (trainer.dp is not None and trainer.dp.active) else "global"
            with torch.cuda.graph(self.graph, capture_error_mode=mode):
                self.loss = trainer._eager_batch(model, self.static, batch_idx, count=False)
            # the Python side of the step ran once while its kernels were only recorded: keep what it did per step, undo this instance
            if hasattr(net, "nbt"):
                self._nbt_per_step = int(net.nbt) - nbt0
                net.nbt.fill_(nbt0)
            self._logged = dict(model.logged)
            self.key = key
            self.captures += 1
        else:
            self._load(batch)
        opt.advance()
        self.graph.replay()
        self.replays += 1
        # the replayed kernels changed the parameters and the running statistics: whatever the inference forward folded out of the old
        # ones is stale (the eager step drops it in FusedAdam.step / the training forward, neither of which runs on a replay)
        inval = getattr(model.net, "invalidate_inference_copies", None)
        if inval is not None:
            inval()
        # host-side bookkeeping the captured kernels do not carry
        model.logged = dict(self._logged)
        model.global_step += 1
        if hasattr(model.net, "nbt") and model.training:
            model.net.nbt += self._nbt_per_step
        return self.loss

    _nbt_per_step = 0


def _flat_scalars(batch, prefix=""):
    for k, v in batch.items():
        if isinstance(v, dict):
            yield from _flat_scalars(v, prefix + k + ".")
        elif not torch.is_tensor(v):
            yield prefix + k, v
