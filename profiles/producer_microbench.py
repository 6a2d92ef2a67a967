"""HBM roofline of the streaming kernels written after round 1's GPU budget was spent (csrc/frames.hip, the temporal heat-map loss,
the inference conv): algorithmic bytes / HIP-event time per launch.  Usage on the GPU box:  python profiles/producer_microbench.py [reps]
Prints one JSON line per kernel: {"kernel", "us", "GB/s", "frac_of_8TBs"} (algorithmic bytes as stated in DESIGN.md)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _lp_bootstrap  # noqa: E402,F401
from lightning_pose_amd import _lib, ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]


def timed(name, nbytes, fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = 1000.0 * e0.elapsed_time(e1) / reps
    gbs = nbytes / us / 1e3
    print(json.dumps({"kernel": name, "us": round(us, 1), "GB/s": round(gbs, 1), "frac_of_8TBs": round(gbs / 8000.0, 4)}), flush=True)


S, H, W = 128, 384, 384
for hs, ws in ((406, 396), (1080, 1920)):
    src = torch.randint(0, 256, (S, hs, ws, 3), device=dev, dtype=torch.uint8)
    timed(f"frames_resize<finish> {hs}x{ws}->{H}x{W}", S * (hs * ws * 3 + H * W * 12), lambda: ops.frames_resize(src, H, W, "clamp", mean=MEAN, std=STD))
    timed(f"frames_resize<raw> {hs}x{ws}->{H}x{W}", S * (hs * ws * 3 + H * W * 12), lambda: ops.frames_resize(src, H, W, "clamp"))
    del src
raw = torch.rand(S, H, W, 3, device=dev) * 255
m = [[1.05, 0.08, -10.0], [-0.08, 0.95, 12.0]]
timed("frames_augment warp+colour", S * H * W * 24, lambda: ops.frames_augment(raw, MEAN, STD, matrix=m, brightness=1.1, contrast=0.9))
timed("frames_augment warp+colour+shot", S * H * W * 24, lambda: ops.frames_augment(raw, MEAN, STD, matrix=m, brightness=1.1, contrast=0.9, shot_factor=5.0, seed=1))
del raw

K, h, w = 17, 96, 96
hm = torch.softmax(torch.randn(S, K, h * w, device=dev), -1).reshape(S, K, h, w).requires_grad_(True)
conf = torch.rand(S, K, device=dev)
eps = torch.zeros(1)
for kind, name in ((_lib.HM_MSE, "mse"), (_lib.HM_KL, "kl")):
    timed(f"temporal_heatmap_{name} fwd", 2 * (S - 1) * K * h * w * 4, lambda: ops.temporal_heatmap_loss(hm.detach(), conf, eps, 0.05, kind))

    def fwd_bwd():
        hm.grad = None
        ops.temporal_heatmap_loss(hm, conf, eps, 0.05, kind).backward()
    timed(f"temporal_heatmap_{name} fwd+bwd", (2 * (S - 1) + 4 * S) * K * h * w * 4, fwd_bwd)
