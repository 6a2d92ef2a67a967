"""world_size-2 gloo test of the data-parallel wiring (DataParallel: broadcast, bucketed gradient all-reduce, scalar
means, SyncBatchNorm switch) on CPU, with a stand-in for the engine's flat buffers."""

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class FlatEngine:
    def __init__(self, n):
        self.device = torch.device("cpu")
        self.P = torch.zeros(n)
        self.G = torch.zeros(n)
        self.R = torch.zeros(8)
        self.sync_bn = False
        self.process_group = None
        self.refreshed = 0

    def refresh_weight_copies(self):
        self.refreshed += 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import _lp_bootstrap  # noqa: F401
    from lightning_pose_amd.distributed import DataParallel, labeled_batch_per_gpu, sequence_length_per_gpu

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 1000
    eng = FlatEngine(n)
    eng.P[:] = float(rank + 1)
    eng.G[:] = torch.arange(n, dtype=torch.float32) * (rank + 1)
    dp = DataParallel(eng, sync_bn=True, bucket_bytes=4 * 300)  # 300-element buckets -> 4 buckets, from the tail
    dp.broadcast_parameters()
    dp.all_reduce_gradients()
    dp.wait()
    means = dp.mean_scalars({"b": torch.tensor(float(rank)), "a": torch.tensor(10.0 * rank)})
    ok = (
        bool((eng.P == 1.0).all()) and eng.refreshed == 1 and eng.sync_bn is True
        and torch.allclose(eng.G, torch.arange(n, dtype=torch.float32) * 3.0)
        and float(means["a"]) == pytest.approx(5.0) and float(means["b"]) == pytest.approx(0.5)
        and labeled_batch_per_gpu(513, 8) == 65 and sequence_length_per_gpu(128, 8) == 16
    )
    q.put((rank, ok))
    dist.destroy_process_group()


def test_data_parallel_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]
