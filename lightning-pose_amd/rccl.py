"""A second RCCL communicator driven straight through librccl's C API, for the SyncBatchNorm messages (opt-in: ``LP_SYNCBN_DIRECT=1``).

The reference's ``SyncBatchNorm`` (train.py:427) exchanges per-layer statistics through torch.distributed; so does the default path here
(engine.py: one ``dist.all_reduce`` per BatchNorm layer and direction).  ProcessGroupNCCL runs every collective on its own internal stream,
so each of the 106 tiny messages of a ResNet-50 step costs two cross-stream event hand-offs (~17 us each way on this stack, measured in
loop-back: profiles/r02_loopback_*) on top of the collective itself - in the middle of a strictly sequential chain of kernels.  This
communicator enqueues ``ncclAllReduce`` ON THE COMPUTE STREAM, between the kernel that produces the sums and the kernel that consumes
them: no hand-off, no Work object, nothing for the caching allocator to track (the buffer is only ever used on that one stream), and the
call is capturable in a HIP graph like any other launch.  The gradient buckets stay on torch.distributed, whose separate stream is what
overlaps them with backward.

Both communicators are used by every rank in the same program order (the step is deterministic), which is what RCCL requires of
concurrent communicators.
"""

from __future__ import annotations

import ctypes as C
import os

import torch
import torch.distributed as dist

NCCL_FLOAT32, NCCL_SUM = 7, 0   # ncclDataType_t / ncclRedOp_t (nccl.h; identical in rccl.h)
_UID_BYTES = 128                # ncclUniqueId { char internal[128]; }


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_ubyte * _UID_BYTES)]   # (not c_char: ctypes would cut a c_char field at its first NUL)


def requested() -> bool:
    return os.environ.get("LP_SYNCBN_DIRECT", "0") == "1"


def _load() -> C.CDLL:
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    lib = C.CDLL(path)  # already mapped by torch: this only hands out the handle
    lib.ncclGetErrorString.restype = C.c_char_p
    lib.ncclGetErrorString.argtypes = [C.c_int]
    lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.ncclCommDestroy.argtypes = [C.c_void_p]
    return lib


class DirectComm:
    """One communicator over the ranks of ``process_group`` (default group if None), bound to the current device."""

    def __init__(self, device: torch.device, process_group=None):
        if device.type != "cuda":
            raise RuntimeError("the direct RCCL communicator needs the ranks' GPUs (LP_SYNCBN_DIRECT=1 with a CPU engine)")
        self._lib = _load()
        self.rank, self.world = dist.get_rank(process_group), dist.get_world_size(process_group)
        uid = _UniqueId()
        if self.rank == 0:
            self._check(self._lib.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        # the id travels over the existing process group (a byte tensor on whatever device that backend moves)
        backend = dist.get_backend(process_group)
        carrier = torch.frombuffer(bytearray(C.string_at(C.addressof(uid), _UID_BYTES)), dtype=torch.uint8).clone()
        carrier = carrier.to(device) if backend == "nccl" else carrier
        src = dist.get_global_rank(process_group, 0) if process_group is not None else 0
        dist.broadcast(carrier, src=src, group=process_group)
        C.memmove(C.addressof(uid), carrier.cpu().numpy().tobytes(), _UID_BYTES)
        self._comm = C.c_void_p()
        with torch.cuda.device(device):
            self._check(self._lib.ncclCommInitRank(C.byref(self._comm), self.world, uid, self.rank), "ncclCommInitRank")
        self.device = device
        self.messages = 0

    def _check(self, rc: int, what: str) -> None:
        if rc != 0:
            raise RuntimeError(f"{what}: {self._lib.ncclGetErrorString(rc).decode()} (rc {rc})")

    def all_reduce_sum_(self, t: torch.Tensor, stream) -> None:
        """In-place SUM of a contiguous fp32 device tensor over the ranks, enqueued on ``stream`` (ops._stream(): the raw hipStream_t as a c_void_p)"""
        if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
            raise ValueError("DirectComm.all_reduce_sum_ takes a contiguous fp32 tensor on the communicator's device")
        p = C.c_void_p(t.data_ptr())
        self._check(self._lib.ncclAllReduce(p, p, t.numel(), NCCL_FLOAT32, NCCL_SUM, self._comm, stream), "ncclAllReduce")
        self.messages += 1

    def destroy(self) -> None:
        if self._comm:
            self._lib.ncclCommDestroy(self._comm)
            self._comm = C.c_void_p()
