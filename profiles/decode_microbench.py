"""Fused decode (lp_decode_fwd / lp_decode_bwd) on FLAT maps (an untrained network: nothing can be pruned) and on PEAKED maps (a trained
network: Gaussians of the target's width), with the exact high-temperature pruning on and off (LP_DECODE_PRUNE):
    python profiles/decode_microbench.py        # on the GPU box"""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _lp_bootstrap  # noqa: E402,F401
from lightning_pose_amd import _lib, ops  # noqa: E402
from lightning_pose_amd.ops import _p  # noqa: E402

lib = _lib.lib()
dev = torch.device("cuda:0")
frames, K, h = 192, 17, 96
g = torch.Generator().manual_seed(0)
flat = torch.softmax(torch.randn(frames, K, h * h, generator=g) * 0.05, -1).reshape(frames, K, h, h).to(dev)
ys = torch.arange(h).view(1, 1, h, 1).float()
xs = torch.arange(h).view(1, 1, 1, h).float()
cx, cy = torch.rand(frames, K, 1, 1, generator=g) * (h - 1), torch.rand(frames, K, 1, 1, generator=g) * (h - 1)
peak = torch.exp(-((xs - cx) ** 2 + (ys - cy) ** 2) / (2 * 1.25 ** 2))
peak = (peak / peak.sum(dim=(2, 3), keepdim=True)).to(dev)
fm = ops.DecodeFrameMap(None, False, None, 1, 4 * h, 4 * h, K)
tables, _keep = ops._device_tables(h, h, 2, dev)
kp_aug, kp_frame = torch.empty(frames, K, 2, device=dev), torch.empty(frames, K, 2, device=dev)
conf, stats = torch.empty(frames, K, device=dev), torch.empty(frames, K, 4, device=dev)
g_frame, g_heat = torch.ones(frames, K, 2, device=dev), torch.empty_like(flat)


def timeit(fn, reps=10):
    for _ in range(2):
        assert fn() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1000.0 * e0.elapsed_time(e1) / reps


for name, hm in (("flat", flat), ("peaked", peak)):
    ref = None
    for prune in ("0", "1"):
        pm = int(prune)   # (the `prune` argument of the two calls: a call argument since round 5)
        st = ops._stream()
        fwd = timeit(lambda: lib.lp_decode_fwd(_p(hm), frames, K, h, h, 2, 1000.0, C.byref(tables), C.byref(fm.struct), _p(kp_aug), _p(kp_frame),
                                               _p(conf), _p(stats), pm, st))
        bwd = timeit(lambda: lib.lp_decode_bwd(_p(hm), frames, K, h, h, 2, 1000.0, C.byref(tables), C.byref(fm.struct), _p(stats), None, _p(g_frame),
                                               _p(g_heat), 0, pm, st))
        torch.cuda.synchronize()
        cur = (kp_aug.clone(), g_heat.clone())
        diff = None if ref is None else (float((cur[0] - ref[0]).abs().max()), float((cur[1] - ref[1]).abs().max() / ref[1].abs().max()))
        ref = ref or cur
        print(json.dumps({"maps": name, "prune": int(prune), "frames": frames, "K": K, "h": h, "decode_fwd_us": round(fwd, 1), "decode_bwd_us": round(bwd, 1),
                          "max_diff_vs_unpruned(kp px, grad rel)": diff}), flush=True)
