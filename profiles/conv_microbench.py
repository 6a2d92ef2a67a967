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

# ---- the launches DESIGN.md section 9 singles out: conv1 data gradients of the non-first blocks, with everything their store pass
# fuses (residual addend, 1-bit ReLU mask of the previous block's output, that block's BatchNorm-backward reductions).  Reports the
# launch time, TFLOP/s and the ALGORITHMIC traffic rate (dz + addend + z + bits read, dx written).
C1 = [  # name, H (= W), C_mid (K of the GEMM), 4 * C_mid (N)
    ("l1.c1_dgrad_fused", 96, 64, 256),
    ("l2.c1_dgrad_fused", 48, 128, 512),
    ("l3.c1_dgrad_fused", 24, 256, 1024),
    ("l4.c1_dgrad_fused", 12, 512, 2048),
]
for name, H, Cm, Cn in C1:
    M = B * H * H
    g = _lib.ConvGeom(B, H, H, Cn, H, H, Cm, 1, 1, 1, 0)          # forward conv1: Cn -> Cm; its data gradient: M x Cm -> M x Cn
    dz = torch.randn(M, Cm, device=dev).to(torch.bfloat16)
    wd = (torch.randn(Cn, Cm, device=dev) * 0.05).to(torch.bfloat16)   # [Ci][R][S][Co] for a 1x1
    addend = torch.randn(M, Cn, device=dev).to(torch.bfloat16)
    z = torch.randn(M, Cn, device=dev).to(torch.bfloat16)
    bits = torch.randint(0, 256, (M * Cn // 8,), device=dev, dtype=torch.uint8)
    mean, invstd = torch.zeros(Cn, device=dev), torch.ones(Cn, device=dev)
    gamma, beta = torch.ones(Cn, device=dev), torch.zeros(Cn, device=dev)
    sums, dbeta, dgamma = torch.zeros(2 * Cn, device=dev), torch.zeros(Cn, device=dev), torch.zeros(Cn, device=dev)
    dx = torch.empty(M, Cn, device=dev, dtype=torch.bfloat16)
    need = int(lib.lp_conv_bn_workspace_bytes(C.byref(g), 1))
    wsb = torch.empty(max(need, 16), device=dev, dtype=torch.uint8)
    f = _lib.BnFuse()
    f.sums, f.workspace, f.workspace_bytes = sums.data_ptr(), wsb.data_ptr(), need
    f.z, f.mean, f.invstd, f.gamma, f.beta = z.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr()
    f.mask_from_z, f.relu_bits = 0, bits.data_ptr()
    f.dbeta_acc, f.dgamma_acc = dbeta.data_ptr(), dgamma.data_ptr()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(reps + 1):
        if it == 1:
            e0.record()
        sums.zero_()
        rc = lib.lp_conv_dgrad_bn(_p(dz), _p(wd), C.byref(g), _p(addend), None, _p(dx), C.byref(f), _stream())
        assert rc == 0, rc
    e1.record()
    torch.cuda.synchronize()
    us = 1000 * e0.elapsed_time(e1) / reps
    nbytes = 2.0 * M * Cm + 3 * 2.0 * M * Cn + M * Cn / 8
    print(f"{name:24s} dgrad+ {us:9.1f} us  {2.0 * M * Cm * Cn / us / 1e6:8.1f} TFLOP/s  {nbytes / us / 1e6:6.2f} TB/s", flush=True)
