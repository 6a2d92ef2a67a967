"""One process per GPU over RCCL (torch.distributed backend "nccl" on ROCm): the data-parallel wiring of the step.

The reference relies on Lightning DDP (train.py:411-431): per-rank data shards, NCCL all-reduce of 25 MB gradient
buckets, SyncBatchNorm, parameter broadcast at start, ``sync_dist=True`` metric means (SURVEY.md sections 3.5, 8e).
Here gradients already live in ONE flat fp32 buffer, so the exchange is a handful of large all-reduces sized for
xGMI's per-link bandwidth rather than DDP's 25 MB default; nothing else crosses GPUs (losses are local means, as in
the reference), and the unlabeled window is never split across ranks.
"""

from __future__ import annotations

import os

import torch
import torch.distributed as dist

GRAD_BUCKET_BYTES = 64 << 20  # 94 MB of fp32 gradients -> 2 buckets; large messages keep the xGMI ring bandwidth-bound


def loopback() -> bool:
    """``LP_DIST_LOOPBACK=1``: form the process group and issue every collective of the step even on a world of ONE rank - the whole RCCL
    path (communicator on the right device, stream hand-offs of the buckets and SyncBatchNorm messages, the logged-scalar message) can then
    be run on a single-GPU box; the arithmetic is unchanged (sums over one rank, divided by one)."""
    return os.environ.get("LP_DIST_LOOPBACK", "0") == "1"


def init_process_group_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """Read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun contract) and initialise the default group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available() and os.environ.get("LP_FORCE_DEVICE") is None:
        # every backend, and before the communicator exists: kernels launch on the current device's stream (ops._stream), and RCCL binds
        # its communicator to the device that is current at its first collective
        torch.cuda.set_device(local_rank)
    if (world > 1 or loopback()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("LP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        # (no device_id=: binding the communicator eagerly was measured in loop-back at +1.8 ms per step - 56.8 vs 55.0 ms on the same box,
        # profiles/archive/r02_final_bench_loopback_rccl*.json.log; the device is already current, which is what RCCL's lazy initialisation uses)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def labeled_batch_per_gpu(train_batch_size: int, num_gpus: int) -> int:
    """reference data/factory.py:252-255"""
    return -(-train_batch_size // num_gpus)


def sequence_length_per_gpu(sequence_length: int, num_gpus: int) -> int:
    """reference data/factory.py:274-276: whole windows per rank, never a split window"""
    return -(-sequence_length // num_gpus)


class DataParallel:
    """Wraps an Engine: parameter broadcast, SyncBatchNorm switch, bucketed gradient all-reduce (mean)."""

    def __init__(self, engine, process_group=None, sync_bn: bool = True, bucket_bytes: int = GRAD_BUCKET_BYTES):
        self.engine = engine
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (loopback() and dist.is_initialized())   # False: every method below is a no-op
        self.bucket_elems = max(1, bucket_bytes // 4)
        engine.process_group = process_group
        engine.sync_bn = bool(sync_bn and self.active)
        self._works: list = []
        self._next_hi: int | None = None   # overlapped mode: upper end of the next bucket to send (None: no step in flight)
        self.buckets_during_backward = 0

    def check_equal_batches(self, *sizes: int) -> None:
        """SyncBatchNorm divides the all-reduced sums by (local rows x world size) and the gradient mean by world size: both assume every
        rank holds equally many frames per step - what DistributedSampler's padding and the per-rank DALI windows give the reference.  One
        MIN / MAX all-reduce at the first step turns a silently wrong normalisation into an error."""
        if not self.active:
            return
        t = torch.tensor([float(v) for v in sizes] + [-float(v) for v in sizes], device=self.engine.device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.pg)
        n = len(sizes)
        lo, hi = t[:n].tolist(), (-t[n:]).tolist()
        if lo != hi:
            raise ValueError(f"ranks hold different batch sizes (min {lo}, max {hi}): SyncBatchNorm and the gradient mean need equal "
                             "per-rank batches (pad the last batch like DistributedSampler does)")

    def broadcast_parameters(self, src: int = 0) -> None:
        if not self.active:
            return
        e = self.engine
        dist.broadcast(e.P, src=src, group=self.pg)
        if e.R.numel():  # BatchNorm running statistics (none for the ViT engine)
            dist.broadcast(e.R, src=src, group=self.pg)
        e.refresh_weight_copies()

    def begin_step(self) -> None:
        """Arm the overlap of the gradient exchange with backward: the engine reports how far its (single) backward pass has come
        (Engine.grad_progress) and every bucket that is final goes out at once - the tail bucket (head, layer4, part of layer3: 64 of the
        94 MB) is on the wire while layers 3..1 and the stem, i.e. most of backward's time, are still being computed."""
        if not self.active:
            return
        self._next_hi = self.engine.G.numel()
        self.buckets_during_backward = 0
        self.engine.grad_progress = self._on_progress

    def _send(self, lo: int, hi: int, async_op: bool = True) -> None:
        e = self.engine
        side = getattr(e, "_side", None)
        if async_op and side is not None and getattr(e, "_side_busy", False) and e.device.type == "cuda":
            # The weight gradients of this range run on the engine's side stream.  The collective is issued FROM that stream (ordered behind
            # the main stream's current point as well: the BatchNorm / bias gradients of the range are main-stream kernels), so RCCL's
            # stream waits for exactly what the bucket needs and the main stream - the data-gradient chain - never waits for anything.
            side.wait_stream(torch.cuda.current_stream(e.device))
            with torch.cuda.stream(side):
                self._works.append(dist.all_reduce(e.G[lo:hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
            return
        join = getattr(e, "_join_side_stream", None)
        if join is not None:
            join()  # the weight gradients of this range run on the engine's side stream: the collective is ordered after them
        self._works.append(dist.all_reduce(e.G[lo:hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=async_op))

    def _on_progress(self, lo_done: int) -> None:
        while self._next_hi is not None and self._next_hi > 0:
            lo = max(0, self._next_hi - self.bucket_elems)
            if lo < lo_done:
                return
            self._send(lo, self._next_hi)
            self.buckets_during_backward += 1
            self._next_hi = lo

    def all_reduce_gradients(self, async_op: bool = True) -> None:
        """SUM all-reduce of the flat gradient buffer in large buckets (whatever an armed backward pass has not sent yet); pair with
        optimizer.grad_scale = 1/world."""
        if not self.active:
            return
        # backward produces the tail of the buffer (head, layer4) first: reduce from the end
        hi = self._next_hi if self._next_hi is not None else self.engine.G.numel()
        self._next_hi = None
        while hi > 0:
            lo = max(0, hi - self.bucket_elems)
            self._send(lo, hi, async_op)
            hi = lo

    def wait(self) -> None:
        for w in self._works:
            if w is not None:
                w.wait()
        self._works.clear()

    def mean_scalars(self, values: dict[str, torch.Tensor]) -> dict[str, torch.Tensor]:
        """``self.log(..., sync_dist=True)`` for all logged scalars in ONE all-reduce (reference base.py:535-544).  Only device-resident
        scalars (the losses and metrics the step computed) travel: host-side values are configuration constants (the loss weights the
        reference logs next to each loss), identical on every rank, so their mean is themselves - and uploading a pageable host scalar
        would block the host until the stream has drained, once per step (and is not capturable in a HIP graph)."""
        if not self.active or not values:
            return values
        dev = self.engine.device
        keys = sorted(k for k, v in values.items() if torch.is_tensor(v) and v.device.type == dev.type)
        if not keys:
            return values
        packed = torch.stack([values[k].detach().float().reshape(()) for k in keys])
        dist.all_reduce(packed, group=self.pg)
        packed /= self.world
        out = dict(values)
        out.update({k: packed[i] for i, k in enumerate(keys)})
        return out
