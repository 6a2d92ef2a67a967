"""One convolution layer at a time through the C ABI (forward / data gradient, plain store pass), for kernel tuning and PMC passes.
    python profiles/conv_layer_bench.py [reps] name:B:H:Ci:Co:k:stride:pad ...      (default: a few trunk shapes at B = 192, 384 px)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _lp_bootstrap  # noqa: E402,F401
from lightning_pose_amd import _lib  # noqa: E402
from lightning_pose_amd.ops import _p, _stream  # noqa: E402

args = sys.argv[1:]
reps = int(args.pop(0)) if args and args[0].isdigit() else 5
DEFAULT = ["l1.c3:192:96:64:256:1:1:0", "l1.c2:192:96:64:64:3:1:1", "l2.c2:192:48:128:128:3:1:1", "l3.c1:192:24:1024:256:1:1:0",
           "l3.c2:192:24:256:256:3:1:1", "l3.c3:192:24:256:1024:1:1:0", "l4.c2:192:12:512:512:3:1:1", "l4.c3:192:12:512:2048:1:1:0"]
dev = torch.device("cuda:0")
lib = _lib.lib()
kinds = os.environ.get("KINDS", "fwd,dgrad").split(",")
for spec in (args or DEFAULT):
    name, B, H, Ci, Co, k, st, pad = spec.split(":")
    B, H, Ci, Co, k, st, pad = map(int, (B, H, Ci, Co, k, st, pad))
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
    nbytes = 2.0 * (B * H * H * Ci + B * Ho * Ho * Co + Co * k * k * Ci)
    for kind in kinds:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(reps + 1):
            if it == 1:
                e0.record()
            if kind == "wgrad":
                rc = lib.lp_conv_wgrad(_p(x), _p(dy), C.byref(g), _p(dw), 0, _p(ws), nws, _stream())
            elif kind == "fwd":
                rc = lib.lp_conv_fwd(_p(x), _p(w), C.byref(g), None, _p(out), None, Co, 0, _stream())
            else:
                rc = lib.lp_conv_dgrad(_p(dy), _p(wd), C.byref(g), None, None, None, _p(dx), None, Ci, 0, 0, _stream())
            assert rc == 0
        e1.record()
        torch.cuda.synchronize()
        us = 1000 * e0.elapsed_time(e1) / reps
        print(f"{name:8s} {kind:6s} {us:9.1f} us  {flops / us / 1e6:8.1f} TFLOP/s  {nbytes / us / 1e6:6.2f} TB/s (algorithmic)", flush=True)
