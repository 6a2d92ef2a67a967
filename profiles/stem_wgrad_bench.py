"""The stem's weight gradient (7x7 / 2, NHWC4, 64 channels) at BASELINE's shape, conv_wgrad_kernel<64, stem> (LP_STEM_WGRAD_NB=0) against
stem_wgrad_nb_kernel (csrc/conv_stem_wgrad.h), alternating in one process; prints us per launch (kernel + its reduction), the byte-floor
fraction and the largest difference between the two gradients.
    python profiles/stem_wgrad_bench.py [B] [HW] [reps]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _lp_bootstrap  # noqa: E402,F401
from lightning_pose_amd import _lib  # noqa: E402
from lightning_pose_amd.ops import _p, _stream  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 192
HW = int(sys.argv[2]) if len(sys.argv) > 2 else 384
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
splits = [int(v) for v in os.environ.get("SPLITS", "0").split(",")]   # pixel slices of the new kernel (0 = the library's choice)
dev = torch.device("cuda:0")
lib = _lib.lib()
g = _lib.ConvGeom(B, HW, HW, 4, HW // 2, HW // 2, 64, 7, 7, 2, 3)
x4 = torch.randn(B, HW, HW, 4, device=dev).to(torch.bfloat16)
x4[..., 3] = 0
dy = torch.randn(B, HW // 2, HW // 2, 64, device=dev).to(torch.bfloat16)
nbytes = x4.numel() * 2 + dy.numel() * 2
out = {}
for rnd in range(3):
    for mode, split in [("0", 0)] + [("1", s_) for s_ in splits]:
        os.environ["LP_STEM_WGRAD_NB"] = mode
        lib.lp_config_reload_env()
        nws = lib.lp_conv_wgrad_workspace_bytes(C.byref(g), split)
        ws = torch.empty(nws, device=dev, dtype=torch.uint8)
        dw = torch.zeros(64, 256, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(reps + 2):
            if it == 2:
                e0.record()
            if it < 2:
                dw.zero_()
            assert lib.lp_stem_wgrad(_p(x4), _p(dy), C.byref(g), _p(dw), split, _p(ws), nws, _stream()) == 0
            if it == 0:
                out[mode] = dw.clone()
        e1.record()
        torch.cuda.synchronize()
        us = 1000 * e0.elapsed_time(e1) / reps
        print(f"round {rnd} LP_STEM_WGRAD_NB={mode} split {split:4d} kernel id {lib.lp_conv_last_kernel()}: {us:8.1f} us per launch   {nbytes / us / 1e6:5.2f} TB/s of the {nbytes / 1e6:.0f} MB the operands hold", flush=True)
a, b = out["0"], out["1"]
print(f"max |old - new| = {float((a - b).abs().max()):.3e}   max |old| = {float(a.abs().max()):.3e}   padding entries zero: "
      f"{not bool(b.reshape(64, 8, 8, 4)[:, 7].any()) and not bool(b.reshape(64, 8, 8, 4)[:, :, 7].any())}")
