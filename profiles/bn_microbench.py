"""BatchNorm streaming kernels at ResNet-50's real shapes (192 frames of 384 x 384): lp_bn_apply (plain / + residual + 1-bit ReLU mask) and
lp_bn_bwd_apply (two-launch form / self-contained form), us per call and algorithmic TB/s.     python profiles/bn_microbench.py   # on the GPU box"""
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
B = 192
SHAPES = [("layer1 bn1/bn2 (64 ch, 96x96)", B * 96 * 96, 64), ("layer1 bn3 (256 ch, 96x96)", B * 96 * 96, 256), ("layer2 bn3 (512 ch, 48x48)", B * 48 * 48, 512),
          ("layer3 bn3 (1024 ch, 24x24)", B * 24 * 24, 1024), ("layer4 bn3 (2048 ch, 12x12)", B * 12 * 12, 2048), ("layer3 bn1 (256 ch, 24x24)", B * 24 * 24, 256)]


def timeit(fn, reps=20):
    for _ in range(3):
        assert fn() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1000.0 * e0.elapsed_time(e1) / reps


for name, M, Cn in SHAPES:
    z = torch.randn(M, Cn, device=dev).to(torch.bfloat16)
    res = torch.randn(M, Cn, device=dev).to(torch.bfloat16)
    y, dz = torch.empty_like(z), torch.empty_like(z)
    bits = torch.empty(M * Cn // 8, device=dev, dtype=torch.uint8)
    mean, invstd = torch.zeros(Cn, device=dev), torch.ones(Cn, device=dev)
    gam, bet = torch.ones(Cn, device=dev), torch.zeros(Cn, device=dev)
    sums = torch.zeros(2 * Cn * 2, device=dev, dtype=torch.int64)      # lp_fxsum (hi, lo) x [2][C]
    terms = torch.empty(2 * Cn, device=dev)
    st = ops._stream()
    n = M * Cn * 2
    t_plain = timeit(lambda: lib.lp_bn_apply(_p(z), _p(mean), _p(invstd), _p(gam), _p(bet), None, 1, M, Cn, _p(y), None, st))
    t_res = timeit(lambda: lib.lp_bn_apply(_p(z), _p(mean), _p(invstd), _p(gam), _p(bet), _p(res), 1, M, Cn, _p(y), None, st))
    t_bits = timeit(lambda: lib.lp_bn_apply(_p(z), _p(mean), _p(invstd), _p(gam), _p(bet), _p(res), 1, M, Cn, _p(y), _p(bits), st))
    t_bwd = timeit(lambda: lib.lp_bn_bwd_apply(_p(res), None, _p(z), _p(mean), _p(invstd), _p(gam), _p(sums), float(M), M, Cn, _p(dz), None, None, None, None, _p(terms), st))
    t_bwd0 = timeit(lambda: lib.lp_bn_bwd_apply(_p(res), None, _p(z), _p(mean), _p(invstd), _p(gam), _p(sums), float(M), M, Cn, _p(dz), None, None, None, None, None, st))
    t_bwdy = timeit(lambda: lib.lp_bn_bwd_apply(_p(res), _p(y), _p(z), _p(mean), _p(invstd), _p(gam), _p(sums), float(M), M, Cn, _p(dz), _p(dz), None, None, None, _p(terms), st))
    print(json.dumps({"shape": name, "MB per tensor": round(n / 1e6, 1),
                      "bn_apply": [round(t_plain, 1), round(2 * n / t_plain / 1e6, 2)],
                      "bn_apply + residual": [round(t_res, 1), round(3 * n / t_res / 1e6, 2)],
                      "bn_apply + residual + bits": [round(t_bits, 1), round((3 * n + n / 16) / t_bits / 1e6, 2)],
                      "bn_bwd_apply (terms)": [round(t_bwd, 1), round(3 * n / t_bwd / 1e6, 2)],
                      "bn_bwd_apply (self-contained)": [round(t_bwd0, 1), round(3 * n / t_bwd0 / 1e6, 2)],
                      "bn_bwd_apply + y mask + dres (terms)": [round(t_bwdy, 1), round(5 * n / t_bwdy / 1e6, 2)]}), flush=True)
