"""Helper of tests/test_widen_bench_helpers.py: ONE rank of `bench.py --gpus 2` on the CPU with the emulated kernel library and gloo
(the launch contract the driver uses for N > 1: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment, rank 0 prints the line)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _lp_bootstrap  # noqa: E402,F401
from lightning_pose_amd import _lib, ops  # noqa: E402
from tests.hipemu import emu  # noqa: E402

_lib._lib = emu.emu_lib()
ops.require_device = lambda *a: None
ops.require_device_type = lambda d: None
ops._stream = lambda: None
import bench  # noqa: E402

bench.main(sys.argv[1:], device=torch.device("cpu"))
