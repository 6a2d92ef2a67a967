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
    dp.check_equal_batches(4, 8)
    try:
        dp.check_equal_batches(4, 8 + rank)
        unequal_caught = False
    except ValueError:
        unequal_caught = True
    ok = (
        bool((eng.P == 1.0).all()) and eng.refreshed == 1 and eng.sync_bn is True
        and torch.allclose(eng.G, torch.arange(n, dtype=torch.float32) * 3.0)
        and float(means["a"]) == pytest.approx(5.0) and float(means["b"]) == pytest.approx(0.5)
        and labeled_batch_per_gpu(513, 8) == 65 and sequence_length_per_gpu(128, 8) == 16 and unequal_caught
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


def _engine_worker(rank, world, port, q, gather=False):
    """one rank of 2-rank data-parallel steps of the REAL engine on the emulated kernels: SyncBatchNorm (statistics from the conv store
    passes, the fused stem, every BatchNorm backward) + bucketed gradient all-reduce, against single-process runs on rank 0.

    A random-init 50-layer BatchNorm network at batch 4 amplifies 1-ulp differences (another summation order is enough) by ~1.5x per
    block, so "2 ranks x 2 frames == 1 process x 4 frames" can only be checked near the input.  The test therefore has three parts:
    (A) different frames per rank: the synchronised moments of the stem and the first blocks equal the whole-batch ones and differ
    from the local ones; (B) the SAME frames on both ranks: sums double and counts double exactly, so every moment, every gradient
    (x 2) must be BIT-identical to the single-process run - this pins the world-size scaling everywhere; (C) every BatchNorm layer
    issues exactly one all-reduce per direction."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["HIPEMU_THREADS"] = "1"  # one emulator thread: workgroups (and their fp32 atomics) run in a fixed order, so (B) can be exact
    os.environ["LP_SYNCBN_GATHER"] = "1" if gather else "0"   # the one-shot exchange (all-gather + local add in rank order) or an all-reduce
    import _lp_bootstrap  # noqa: F401
    from lightning_pose_amd import _lib, ops
    from lightning_pose_amd.distributed import DataParallel
    from lightning_pose_amd.engine import Engine
    from lightning_pose_amd.models.backbones._init import seeded_state_dict
    from tests.hipemu import emu

    _lib._lib = emu.emu_lib()
    ops.require_device = lambda *a: None
    ops.require_device_type = lambda d: None
    ops._stream = lambda: None
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    calls = {"n": 0}
    real_all_reduce = dist.all_reduce

    def counting_all_reduce(*a, **k):
        calls["n"] += 1
        return real_all_reduce(*a, **k)

    dist.all_reduce = counting_all_reduce
    real_all_gather = dist.all_gather_into_tensor

    def counting_all_gather(*a, **k):   # (a SyncBatchNorm message in gather mode: counted like an all-reduce)
        calls["n"] += 1
        return real_all_gather(*a, **k)

    dist.all_gather_into_tensor = counting_all_gather
    K, dev = 3, torch.device("cpu")
    torch.manual_seed(7)
    sd = seeded_state_dict(K, 2)
    gen = torch.Generator().manual_seed(0)
    images = torch.randn(2 * world, 3, 64, 64, generator=gen)   # (world 2: the 4 frames of rounds 1 - 3, same draw)
    g_heat = torch.randn(2 * world, K, 16, 16, generator=gen)

    def make(sync):
        e = Engine(K, 2, dev)
        e.load_state_dict(sd, strict=False)
        d = DataParallel(e, sync_bn=True) if sync else None
        if d is not None:
            d.broadcast_parameters()
        return e, d

    bad = []
    # ---- (A) different frames per rank, forward only
    eng, dp = make(True)
    _, tape = eng.forward(images[2 * rank:2 * rank + 2], True)
    if rank == 0:
        whole, _ = make(False)
        _, t_whole = whole.forward(images, True)
        local, _ = make(False)
        _, t_local = local.forward(images[:2], True)
        for key in ("stem.mu", "stem.iv", "b0.m1", "b0.v1", "b0.m3", "b0.v3", "b0.md", "b1.m2", "b1.v2", "b2.m1"):
            if not torch.allclose(tape.t[key], t_whole.t[key], atol=5e-3, rtol=5e-3):
                bad.append(f"A:{key} != whole batch")
        if torch.allclose(tape.t["stem.mu"], t_local.t["stem.mu"], atol=1e-4, rtol=1e-3):
            bad.append("A:stem.mu equals the LOCAL statistics")
    # ---- (B) + (C) the same frames on both ranks, forward + backward + gradient all-reduce
    eng, dp = make(True)
    n_bn = len(eng.plan.bns)
    calls["n"] = 0
    _, tape = eng.forward(images[:2], True)
    n_fwd, calls["n"] = calls["n"], 0
    eng.zero_grad()
    # the gradient exchange is armed before backward (DataParallel.begin_step): with 8 MB buckets the tail buckets (head, layer4) must
    # leave WHILE backward is still running, and the result must not change
    dp.bucket_elems = (8 << 20) // 4
    eng.single_backward = True
    dp.begin_step()
    eng.backward(tape, g_heat[:2])
    early = dp.buckets_during_backward
    n_bwd = calls["n"] - early
    dp.all_reduce_gradients()
    dp.wait()
    if (n_fwd, n_bwd) != (n_bn, n_bn):   # (one message per layer and direction whatever the world size)
        bad.append(f"C:{n_fwd} forward / {n_bwd} backward all-reduces for {n_bn} BatchNorm layers")
    if not (3 <= early < -(-eng.G.numel() * 4 // (8 << 20))):
        bad.append(f"overlap: {early} gradient buckets left during backward")
    # ---- (D) joint pass (two BatchNorm segments): ONE message carries both segments' sums; the moments are the global per-segment ones
    calls["n"] = 0
    b0 = eng.plan.blocks[0].bn1
    zz = torch.randn(4, 6, 6, b0.C, generator=torch.Generator().manual_seed(5)).to(torch.bfloat16)   # same on both ranks
    sums = torch.zeros(2 * 2 * b0.C * 2, dtype=torch.int64)   # (segments, 2, C) fixed-point sums: two int64 words each
    mu, iv = eng._bn_moments(b0, zz, 4 * 36, True, sums, have_sums=False, seg=1)
    if calls["n"] != 1:
        bad.append(f"D:{calls['n']} all-reduces for one two-segment BatchNorm layer")
    zf = zz.float().reshape(4, 36, b0.C)
    for si, sl in enumerate((slice(0, 1), slice(1, 4))):
        want = zf[sl].reshape(-1, b0.C).mean(0)
        if not torch.allclose(mu[si * b0.C:(si + 1) * b0.C], want, atol=1e-5, rtol=1e-4):
            bad.append(f"D:segment {si} mean")
    if rank == 0:
        solo, _ = make(False)
        _, t_solo = solo.forward(images[:2], True)
        solo.zero_grad()
        solo.backward(t_solo, g_heat[:2])
        for key, v in t_solo.t.items():
            if key.split(".")[-1] in ("mu", "iv", "m1", "v1", "m2", "v2", "m3", "v3", "md", "vd") and not torch.equal(tape.t[key], v):
                bad.append(f"B:{key} not bit-identical")
        # world 2: x + x is exact.  Beyond it the gradient buckets go through gloo's / RCCL's ring, whose partial sums 3x, 5x, 7x of equal
        # terms round (the SyncBatchNorm sums above do not: they are integers) - equal to fp32 rounding then
        same = torch.equal(eng.G, world * solo.G) if world == 2 else torch.allclose(eng.G, world * solo.G, rtol=2e-6, atol=1e-7 * float(solo.G.abs().max()))
        if not same:
            bad.append(f"B:summed gradient != {world} x single-process gradient (max diff {float((eng.G - world * solo.G).abs().max()):.3e})")
        sb = eng.plan.stem_bn
        if not torch.equal(eng.running_view(sb, "running_mean"), solo.running_view(sb, "running_mean")):
            bad.append("B:stem running_mean")
        m = 2 * 32 * 32  # stem pixels per rank: unbiased variance uses the GLOBAL count world * m
        ratio = (eng.running_view(sb, "running_var") - 0.9) / (solo.running_view(sb, "running_var") - 0.9)
        if not torch.allclose(ratio, torch.full_like(ratio, (world * m / (world * m - 1)) / (m / (m - 1))), rtol=1e-4):
            bad.append("B:stem running_var (global count)")
    # ---- (E) the ViT engine: no BatchNorm messages, gradient buckets leave layer by layer during its backward, the sum is exact
    from lightning_pose_amd.vit_engine import ViTEngine

    def make_vit():
        torch.manual_seed(11)
        e = ViTEngine(K, 2, dev, hidden=128, depth=3, heads=2, mlp=256, patch=16, pretrain_grid=3)
        e.P.copy_(torch.randn(e.P.shape, generator=torch.Generator().manual_seed(3)) * 0.05)
        for l in e.plan.norms():
            e.P[l.g_off:l.g_off + 128] = 1.0
        e.refresh_weight_copies()
        return e

    vit = make_vit()
    vdp = DataParallel(vit, sync_bn=True, bucket_bytes=256 << 10)
    vdp.broadcast_parameters()
    calls["n"] = 0
    _, vt = vit.forward(images[:2], True)
    vit.zero_grad()
    vit.single_backward = True
    vdp.begin_step()
    vit.backward(vt, g_heat[:2])
    v_early = vdp.buckets_during_backward
    vdp.all_reduce_gradients()
    vdp.wait()
    n_buckets = -(-vit.G.numel() * 4 // (256 << 10))
    if calls["n"] != n_buckets:
        bad.append(f"E:{calls['n']} all-reduces for {n_buckets} gradient buckets (a ViT has no SyncBatchNorm traffic)")
    if not (2 <= v_early < n_buckets):
        bad.append(f"E:{v_early} of {n_buckets} ViT gradient buckets left during backward")
    if rank == 0:
        vsolo = make_vit()
        _, vts = vsolo.forward(images[:2], True)
        vsolo.zero_grad()
        vsolo.backward(vts, g_heat[:2])
        if not (torch.equal(vit.G, world * vsolo.G) if world == 2 else torch.allclose(vit.G, world * vsolo.G, rtol=2e-6, atol=1e-7 * float(vsolo.G.abs().max()))):
            bad.append(f"E:summed ViT gradient != {world} x single-process gradient (max diff {float((vit.G - world * vsolo.G).abs().max()):.3e})")
    q.put((rank, not bad, "; ".join(bad[:6])))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,gather", [(2, False), (2, True), (8, True)], ids=["all_reduce", "one_shot_gather", "world8_one_shot_gather"])
def test_sync_batchnorm_engine_gloo_world2(world, gather):
    """SyncBatchNorm + summed gradients of the real engine across gloo ranks (see _engine_worker), with the messages as all-reduces and as
    the one-shot exchange (LP_SYNCBN_GATHER=1: all-gather into per-rank rows, added locally).  The messages are fixed-point sums (int64
    words), so either form gives the same bits on every rank at any world size and part (B)'s exact comparisons hold for both.  World 8
    (round 4: what the driver's SCALE run launches) runs the one-shot form."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_engine_worker, args=(r, world, port, q, gather)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=1500) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, True, "") for r in range(world)], res


def _tracker_worker(rank, world, port, q):
    """One supervised training step of the tracker in fp32, sharded by frame over 2 ranks (SyncBatchNorm, summed gradient, grad_scale =
    1 / world) against the same step over all frames in one process - the definition of data parallelism the reference gets from Lightning
    DDP + SyncBatchNorm (train.py:411-431).  fp32 so that the comparison is not limited by the bf16 policy's own noise."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import _lp_bootstrap  # noqa: F401
    from lightning_pose_amd import _lib, ops
    from lightning_pose_amd.distributed import DataParallel
    from lightning_pose_amd.losses import LossFactory
    from lightning_pose_amd.models import HeatmapTracker
    from oracle import restated as O
    from tests.hipemu import emu

    _lib._lib = emu.emu_lib()
    ops.require_device = lambda *a: None
    ops.require_device_type = lambda d: None
    ops._stream = lambda: None
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    dev, K, HW, Bper = torch.device("cpu"), 3, 64, (2 if world <= 4 else 1)
    gen = torch.Generator().manual_seed(5)
    images = torch.randn(world * Bper, 3, HW, HW, generator=gen)
    kp = torch.rand(world * Bper, K, 2, generator=gen) * HW
    heat = O.generate_heatmaps(kp, HW, HW, (HW // 4, HW // 4))

    def batch(lo, hi):
        return {"images": images[lo:hi], "keypoints": kp[lo:hi].reshape(hi - lo, 2 * K), "heatmaps": heat[lo:hi],
                "bbox": torch.tensor([[0.0, 0.0, float(HW), float(HW)]]).repeat(hi - lo, 1), "idxs": torch.arange(lo, hi)}

    def make():
        return HeatmapTracker(num_keypoints=K, loss_factory=LossFactory({"heatmap_mse": {"log_weight": 0.0}}, None), backbone="resnet50",
                              pretrained=False, torch_seed=3, device=dev, precision="fp32")

    bad = []
    model = make()
    dp = DataParallel(model.net, sync_bn=True)
    dp.broadcast_parameters()
    model.train()
    opt = model.configure_optimizers()["optimizer"]
    opt.grad_scale = 1.0 / world
    opt.zero_grad()
    out = model.training_step(batch(rank * Bper, (rank + 1) * Bper), 0)
    out["loss"].backward()
    dp.all_reduce_gradients()
    dp.wait()
    mean_loss = dp.mean_scalars({"loss": out["loss"].detach()})["loss"]
    if rank == 0:
        solo = make()
        solo.train()
        solo.configure_optimizers()["optimizer"].zero_grad()
        want = solo.training_step(batch(0, world * Bper), 0)
        want["loss"].backward()
        # every map is labeled, so the mean over all frames is the mean of the per-rank means
        if abs(float(mean_loss) - float(want["loss"].detach())) > 1e-5 * abs(float(want["loss"].detach())):
            bad.append(f"loss: mean over ranks {float(mean_loss):.8f} vs all frames in one process {float(want['loss']):.8f}")
        g_dp, g_solo = model.net.G / world, solo.net.G
        n_bb = model.net.plan.n_backbone
        # (backbone: fp32 rounding differences - the ranks' partial sums are added in another order than one process adds its rows - grow through
        # 50 BatchNorm layers whose deepest normalise over 16 values per channel; observed 1.5e-2 .. 2.5e-2 from run to run of the threaded
        # emulator, against 2e-4 .. 1e-3 for the head)
        for name, sl, tol in (("head", slice(n_bb, None), 2e-3), ("backbone", slice(0, n_bb), 6e-2)):
            a, b = g_dp[sl], g_solo[sl]
            rel = float((a - b).norm() / b.norm())
            if not rel < tol:
                bad.append(f"{name} gradient: |mean over ranks - single process| / |single process| = {rel:.2e}")
        rm_dp = model.net.running_view(model.net.plan.stem_bn, "running_mean")
        rm_solo = solo.net.running_view(solo.net.plan.stem_bn, "running_mean")
        if not torch.allclose(rm_dp, rm_solo, atol=1e-6, rtol=1e-4):
            bad.append("stem running_mean differs from the single-process (global batch) statistics")
    q.put((rank, not bad, "; ".join(bad)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_tracker_step_sharded_over_ranks_equals_one_process(world):
    """(world 4 and 8: the SCALE runs go to 8 ranks - bucketed gradient sum, SyncBatchNorm counts and the 1 / world scale beyond two ranks;
    world 8 shards one frame per rank, the smallest per-rank batch `ceil(batch / n)` produces)"""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_tracker_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in results), [msg for _, ok, msg in results if not ok]
