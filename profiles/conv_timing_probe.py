"""Where a K step's time goes inside the forward convolution kernels: drives the TIMING BUILD of the library (profiles/r05k_timing_build.patch
applied to a copy of csrc/: s_memtime around the waits of wave 0 [and of producer wave 4 in conv_spec_kernel], per-workgroup sums in a
__device__ array read back by lp_debug_conv_timing) over one launch per layer and prints mean shader-clock cycles per K step.
    LP_HIP_LIB=build/liblp_hip_timing.so python profiles/conv_timing_probe.py [name:B:H:Ci:Co:k:stride:pad ...]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _lp_bootstrap  # noqa: E402,F401
from lightning_pose_amd import _lib  # noqa: E402
from lightning_pose_amd.ops import _p, _stream  # noqa: E402

DEFAULT = ["l1.c2:192:96:64:64:3:1:1", "l2.c2:192:48:128:128:3:1:1", "l3.c2:192:24:256:256:3:1:1", "l4.c2:192:12:512:512:3:1:1",
           "l3.c1:192:24:1024:256:1:1:0", "l3.c3:192:24:256:1024:1:1:0", "l4.c3:192:12:512:2048:1:1:0"]
dev = torch.device("cuda:0")
lib = _lib.lib()
lib.lp_debug_conv_timing.argtypes = [C.c_void_p]
lib.lp_debug_conv_timing.restype = C.c_int
buf = np.zeros((256, 8, 8), dtype=np.uint64)
kinds = os.environ.get("KINDS", "fwd").split(",")
for spec in (sys.argv[1:] or DEFAULT):
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
    flops = 2.0 * B * Ho * Ho * Co * k * k * Ci
    for kind in kinds:
        for mode in (("0", "1") if kind == "fwd" else ("0",)):
            os.environ["LP_CONV_SPEC"] = mode
            lib.lp_config_reload_env()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for it in range(6):
                if it == 5:
                    e0.record()
                if kind == "fwd":
                    rc = lib.lp_conv_fwd(_p(x), _p(w), C.byref(g), None, _p(out), None, Co, 0, _stream())
                else:
                    rc = lib.lp_conv_dgrad(_p(dy), _p(wd), C.byref(g), None, None, None, _p(dx), None, Ci, 0, 0, _stream())
                assert rc == 0
            e1.record()
            torch.cuda.synchronize()
            us = 1000 * e0.elapsed_time(e1)
            kern = lib.lp_conv_last_kernel()
            assert lib.lp_debug_conv_timing(buf.ctypes.data) == 0
            d = buf.astype(np.float64)
            if kern in (6, 7):    # conv_spec_kernel: row of wave 0 = consumer wave 0 {barrier, body, tile end, steps, total}, producer wave 4 {vm, barrier, body}
                d = d[:, 0, :]
                steps = d[:, 3]
                ok = steps > 0
                sN = steps[ok]
                print(f"{name:7s} {kind} spec kern={kern} {us:8.1f} us {flops / us / 1e6:7.1f} TF | cycles per K step (mean of {ok.sum()} WGs x {sN.mean():.0f} steps): "
                      f"consumer barrier {np.mean(d[ok, 0] / sN):6.0f} body {np.mean(d[ok, 1] / sN):6.0f} tile-end/step {np.mean(d[ok, 2] / sN):5.0f} | "
                      f"producer vmwait {np.mean(d[ok, 5] / sN):6.0f} barrier {np.mean(d[ok, 6] / sN):6.0f} body {np.mean(d[ok, 7] / sN):6.0f} | "
                      f"kernel cycles {d[ok, 4].mean():9.0f}", flush=True)
            else:                 # conv_pipe_kernel, every wave {vm, barrier, body, store pass, steps, cycles, 100 MHz ticks, first-slice fetch}
                steps = d[:, 0, 4]
                ok = steps > 0
                if not ok.any():
                    print(f"{name:7s} {kind} kern={kern} {us:8.1f} us {flops / us / 1e6:7.1f} TF (not an instrumented kernel)", flush=True)
                    continue
                sN = steps[ok][:, None]
                dd = d[ok]
                clk = dd[:, 0, 5].sum() / dd[:, 0, 6].sum() * 0.1   # GHz: shader cycles per 100 MHz tick
                print(f"{name:7s} {kind} pipe kern={kern} {us:8.1f} us {flops / us / 1e6:7.1f} TF | {ok.sum()} WGs x {sN.mean():.0f} K steps, kernel "
                      f"{dd[:, 0, 5].mean():9.0f} cycles = {dd[:, 0, 6].mean() / 100:7.1f} us in-kernel, shader clock {clk:5.3f} GHz; cycles per K step by wave:", flush=True)
                for nm, j in (("vmwait", 0), ("barrier", 1), ("fetch0", 7), ("body", 2), ("store/step", 3)):
                    print(f"          {nm:10s} " + " ".join(f"{v:6.0f}" for v in (dd[:, :, j] / sN).mean(0)), flush=True)
