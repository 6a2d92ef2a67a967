"""Run a few representative MFMA convolution launches (real ResNet-50 layer shapes at B=128, 384x384) so rocprofv3 PMC
passes can attribute counters to them.  Usage on the GPU box:  python profiles/conv_microbench.py [reps]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _lp_bootstrap  # noqa: E402,F401
from lightning_pose_amd import _lib  # noqa: E402
from lightning_pose_amd.ops import _p, _stream  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
lib = _lib.lib()
B = 128
SHAPES = [  # name, Hi, Ci, Co, k, stride, pad
    ("l1_conv1_1x1_256_64", 96, 256, 64, 1, 1, 0),
    ("l1_conv2_3x3_64", 96, 64, 64, 3, 1, 1),
    ("l1_conv3_1x1_64_256", 96, 64, 256, 1, 1, 0),
    ("l2_conv2_3x3_128", 48, 128, 128, 3, 1, 1),
    ("l3_conv2_3x3_256", 24, 256, 256, 3, 1, 1),
    ("l3_conv3_1x1_256_1024", 24, 256, 1024, 1, 1, 0),
    ("l4_conv2_3x3_512", 12, 512, 512, 3, 1, 1),
    ("l3_0_conv2_3x3_s2", 48, 256, 256, 3, 2, 1),
]
for name, H, Ci, Co, k, st, pad in SHAPES:
    Ho = (H + 2 * pad - k) // st + 1
    g = _lib.ConvGeom(B, H, H, Ci, Ho, Ho, Co, k, k, st, pad)
    x = torch.randn(B, H, H, Ci, device=dev).to(torch.bfloat16)
    w = (torch.randn(Co, k, k, Ci, device=dev) * 0.05).to(torch.bfloat16)
    wd = w.permute(3, 1, 2, 0).contiguous()
    dy = torch.randn(B, Ho, Ho, Co, device=dev).to(torch.bfloat16)
    out = torch.empty(B, Ho, Ho, Co, device=dev, dtype=torch.bfloat16)
    dx = torch.empty(B, H, H, Ci, device=dev, dtype=torch.bfloat16)
    dw = torch.zeros(Co, k * k * Ci, device=dev)
    nws = lib.lp_conv_wgrad_workspace_bytes(C.byref(g), 0)
    ws = torch.empty(nws, device=dev, dtype=torch.uint8)
    flops = 2.0 * B * Ho * Ho * Co * k * k * Ci
    for kind in ("fwd", "dgrad", "wgrad"):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(reps + 1):
            if it == 1:
                e0.record()
            if kind == "fwd":
                rc = lib.lp_conv_fwd(_p(x), _p(w), C.byref(g), None, _p(out), None, Co, 0, _stream())
            elif kind == "dgrad":
                rc = lib.lp_conv_dgrad(_p(dy), _p(wd), C.byref(g), None, None, None, _p(dx), None, Ci, 0, 0, _stream())
            else:
                rc = lib.lp_conv_wgrad(_p(x), _p(dy), C.byref(g), _p(dw), 0, _p(ws), nws, _stream())
            assert rc == 0
        e1.record()
        torch.cuda.synchronize()
        us = 1000 * e0.elapsed_time(e1) / reps
        print(f"{name:24s} {kind:6s} {us:9.1f} us  {flops / us / 1e6:8.1f} TFLOP/s", flush=True)
