"""The un-fused BatchNorm backward reductions of the step (lp_bn_bwd_reduce = colreduce_kernel<1>: the downsample BatchNorms and the inputs of the
stride-2 blocks) through the C ABI at the bench's shapes, one BatchNorm segment of 128 frames each: time per launch and bytes / time.
    python profiles/colreduce_microbench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _lp_bootstrap  # noqa: E402,F401
from lightning_pose_amd import _lib  # noqa: E402
from lightning_pose_amd.ops import _p, _stream  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.lib()
tot = 0.0
for name, hw, C_ in (("l1 (96x96x256)", 96, 256), ("l2 (48x48x512)", 48, 512), ("l3 (24x24x1024)", 24, 1024), ("l4 (12x12x2048)", 12, 2048)):
    M = 128 * hw * hw
    dy = torch.randn(M, C_, device=dev).to(torch.bfloat16)
    z = torch.randn(M, C_, device=dev).to(torch.bfloat16)
    mean, invstd = torch.randn(C_, device=dev) * 0.1, torch.rand(C_, device=dev) + 0.5
    sums, db, dg = torch.zeros(2 * C_, device=dev), torch.zeros(C_, device=dev), torch.zeros(C_, device=dev)
    fn = lambda: lib.lp_bn_bwd_reduce(_p(dy), None, _p(z), _p(mean), _p(invstd), M, C_, _p(sums), _p(db), _p(dg), _stream())  # noqa: E731
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        assert fn() == 0
    e1.record()
    torch.cuda.synchronize()
    us = 100 * e0.elapsed_time(e1)
    tot += us
    print(f"{name:18s} {us:8.1f} us  {4 * M * C_ / us / 1e6:6.2f} TB/s", flush=True)
print(f"sum {tot:.1f} us")
