"""The stem's fused BatchNorm / ReLU / max-pool passes through the C ABI at the bench's shape (one BatchNorm segment of `B` frames, 192 x 192 x 64
pre-normalisation tensor): time per launch and bytes / time against the 5.4 TB/s a read + write stream reaches on this part.
    python profiles/pool_microbench.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _lp_bootstrap  # noqa: E402,F401
from lightning_pose_amd import _lib  # noqa: E402
from lightning_pose_amd.ops import _p, _stream  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
H = W = 192
C_ = 64
Ho = Wo = 96
dev = torch.device("cuda:0")
lib = _lib.lib()
z = torch.randn(B, H, W, C_, device=dev).to(torch.bfloat16)
mean, invstd = torch.randn(C_, device=dev) * 0.1, torch.rand(C_, device=dev) + 0.5
gamma, beta = torch.rand(C_, device=dev) + 0.5, torch.randn(C_, device=dev) * 0.3
y = torch.empty(B, Ho, Wo, C_, device=dev, dtype=torch.bfloat16)
arg = torch.empty(B, Ho, Wo, C_, device=dev, dtype=torch.uint8)
dy = torch.randn(B, Ho, Wo, C_, device=dev).to(torch.bfloat16)
sums = torch.zeros(2 * C_, device=dev)
dbeta, dgamma = torch.zeros(C_, device=dev), torch.zeros(C_, device=dev)
dz = torch.empty_like(z)
nz, ny = z.numel() * 2, y.numel() * 2


def timed(name, fn, nbytes, reps=10):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        assert fn() == 0
    e1.record()
    torch.cuda.synchronize()
    us = 1000 * e0.elapsed_time(e1) / reps
    print(f"{name:24s} {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s", flush=True)


timed("bn_relu_maxpool_fwd", lambda: lib.lp_bn_relu_maxpool_fwd(_p(z), _p(mean), _p(invstd), _p(gamma), _p(beta), B, H, W, C_, _p(y), _p(arg), _stream()),
      nz + ny + arg.numel())
timed("bn_pool_bwd_reduce", lambda: lib.lp_bn_pool_bwd_reduce(_p(arg), _p(dy), _p(z), _p(mean), _p(invstd), _p(gamma), _p(beta), B, H, W, C_, _p(sums),
                                                               _p(dbeta), _p(dgamma), _stream()), nz + ny + arg.numel())
timed("bn_pool_bwd_apply", lambda: lib.lp_bn_pool_bwd_apply(_p(arg), _p(dy), _p(z), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(sums),
                                                             float(B * H * W), B, H, W, C_, _p(dz), _stream()), 2 * nz + ny + arg.numel())
