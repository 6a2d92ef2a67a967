"""Streaming rate of the BatchNorm element-wise kernels at the step's own shapes:
    python profiles/bn_microbench.py          # on the GPU box
bytes = every tensor the launch reads or writes once (bf16 activations, 1-bit masks).

Round 2 used it to A/B kernel variants behind LP_BN_VARIANT (1: deeper unroll, 2: non-temporal stores, 3: both; results in
profiles/archive/r02f_bn_microbench.jsonl): tensors of 900 MB stream at 4.4 - 4.7 TB/s (the practical mixed read/write HBM rate of this part), tensors
that fit the 256 MB Infinity Cache at 5.9 - 6.4 TB/s in this loop only because the loop re-reads them.  Non-temporal stores gained 5 - 8 % here and
NOTHING in the step (3541 vs 3553 frames/s, profiles/archive/r02g_bench_*.json.log), so the variants were not kept; the library now ignores the variable."""
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
SHAPES = [("l1 wide", B * 96 * 96, 256), ("l1 narrow", B * 96 * 96, 64), ("l2 wide", B * 48 * 48, 512), ("l3 wide", B * 24 * 24, 1024),
          ("l4 wide", B * 12 * 12, 2048)]


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1000.0 * e0.elapsed_time(e1) / reps


for variant in (os.environ.get("LP_BN_VARIANTS", "0,1,2,3").split(",")):
    os.environ["LP_BN_VARIANT"] = variant
    for name, M, Cn in SHAPES:
        z = torch.randn(M, Cn, device=dev).to(torch.bfloat16)
        res = torch.randn(M, Cn, device=dev).to(torch.bfloat16)
        y = torch.empty_like(z)
        bits = torch.empty(M * Cn // 8, device=dev, dtype=torch.uint8)
        mean, invstd = torch.zeros(Cn, device=dev), torch.ones(Cn, device=dev)
        gam, bet = torch.ones(Cn, device=dev), torch.zeros(Cn, device=dev)
        sums = torch.zeros(2 * Cn, device=dev)
        dz = torch.empty_like(z)
        st = ops._stream()
        n = M * Cn * 2
        t_app = timeit(lambda: lib.lp_bn_apply(_p(z), _p(mean), _p(invstd), _p(gam), _p(bet), _p(res), 1, M, Cn, _p(y), _p(bits), st))
        t_app0 = timeit(lambda: lib.lp_bn_apply(_p(z), _p(mean), _p(invstd), _p(gam), _p(bet), None, 1, M, Cn, _p(y), None, st))
        t_bwd = timeit(lambda: lib.lp_bn_bwd_apply(_p(res), None, _p(z), _p(mean), _p(invstd), _p(gam), _p(sums), float(M), M, Cn, _p(dz), None, None, None, None, _p(terms), st))
        print(json.dumps({"variant": variant, "shape": name, "M": M, "C": Cn,
                          "bn_apply+res+bits": {"us": round(t_app, 1), "TB/s": round((3 * n + n / 16) / t_app / 1e6, 2)},
                          "bn_apply": {"us": round(t_app0, 1), "TB/s": round(2 * n / t_app0 / 1e6, 2)},
                          "bn_bwd_apply": {"us": round(t_bwd, 1), "TB/s": round(3 * n / t_bwd / 1e6, 2)}}), flush=True)
        del z, res, y, bits, dz
