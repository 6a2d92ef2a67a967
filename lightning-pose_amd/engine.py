"""Explicit forward/backward executor for the heatmap tracker's network (ResNet-50 trunk + PixelShuffle /
ConvTranspose head + spatial softmax) on top of the lp_hip kernels.

Design (MI355X-first, not a translation of the reference's module tree):
  * ONE flat fp32 buffer each for parameters, gradients and Adam moments (plus a flat bf16 operand copy and a
    flat buffer of transposed copies for the data-gradient GEMMs).  Gradients therefore all-reduce as a few large
    RCCL buckets and the optimiser is one launch per parameter group.
  * activations NHWC bf16, conv weights [Cout][R][S][Cin] (torch ``channels_last``), so every contraction is a
    K-contiguous GEMM on the MFMA units (csrc/conv.hip); BatchNorm / ReLU / residual / pooling are 16-byte-per-lane
    streaming kernels (csrc/bn.hip).
  * the network is static, so forward records a plain Python "tape" of tensors and backward walks it in reverse
    calling dgrad / wgrad kernels directly - no autograd graph, no tracing compiler.  288 GB of HBM lets every
    activation stay resident (about 0.13 GB per 384x384 frame); the one exception is the stem's BatchNorm -> ReLU -> max-pool,
    fused so that the full-resolution activation between them (and its gradient) never exists.

Reference behaviour reproduced (paths relative to the reference tree): torchvision ResNet-50 children[:-2]
(models/backbones/factory.py:322-348), HeatmapHead (models/heads/heatmap.py:20-83,147-212), training-mode
BatchNorm per forward call with running-stat updates, SyncBatchNorm across ranks (train.py:427).
"""

from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass, field

import torch
import torch.distributed as dist

from . import _lib
from ._lib import check
from . import ops
from .ops import _p

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
CPAD = 64  # channel padding of the head's narrow tensors (K = 17 -> 64) so they are MFMA K/N operands


@dataclass
class ConvP:
    name: str
    kind: str            # "conv" | "stem" | "convT"
    cin: int             # logical channels (torch view)
    cout: int
    k: int
    stride: int
    pad: int
    Co: int              # storage dims of the GEMM weight [Co][k][k][Ci]
    Ci: int
    w_off: int = 0
    wd_off: int = 0
    bias_off: int = -1   # convT only (padded to Ci)

    @property
    def numel(self) -> int:
        kk = 8 if self.kind == "stem" else self.k
        return self.Co * kk * kk * self.Ci


@dataclass
class BNP:
    name: str
    C: int
    g_off: int = 0
    b_off: int = 0
    r_off: int = 0       # running_mean at r_off, running_var at r_off + C (in the running-stat buffer)


@dataclass
class Block:
    conv1: ConvP
    bn1: BNP
    conv2: ConvP
    bn2: BNP
    conv3: ConvP
    bn3: BNP
    down: ConvP | None = None
    dbn: BNP | None = None


@dataclass
class Plan:
    stem: ConvP
    stem_bn: BNP
    blocks: list[Block]
    head: list[ConvP]
    n_backbone: int = 0   # flat range [0, n_backbone) is the "backbone" optimiser group
    n_total: int = 0
    n_wd: int = 0
    n_running: int = 0
    convs: list[ConvP] = field(default_factory=list)
    bns: list[BNP] = field(default_factory=list)


def build_head(feat_channels: int, stride: int, num_keypoints: int, downsample_factor: int, int_channels: int | None = None) -> list[ConvP]:
    """PixelShuffle(2) + n x ConvTranspose2d(k3,s2,p1,op1) of HeatmapHead (reference models/heads/heatmap.py:20-71,186-196):
    n = log2(stride) - downsample_factor - 1; every layer but the last has ``int_channels`` outputs (``deconv_out_channels``; default: the
    keypoints).  Each ConvTranspose2d(cin -> cout) is stored as the mirrored convolution's weight
    [Co = cin rounded up to 64][3][3][Ci = cout rounded up to 64 (CPAD for the keypoints)]."""
    n_layers = int(math.log2(stride)) - downsample_factor - 1
    if n_layers < 1:
        raise NotImplementedError(f"downsample_factor={downsample_factor} leaves no upsampling layer for a stride-{stride} backbone")
    head: list[ConvP] = []
    cin = feat_channels // 4
    for i in range(n_layers):
        cout = num_keypoints if i == n_layers - 1 else (int_channels or num_keypoints)
        co_store, ci_store = -(-cin // 64) * 64, -(-cout // 64) * 64
        head.append(ConvP(f"head.upsampling_layers.{i + 1}", "convT", cin, cout, 3, 2, 1, co_store, ci_store))
        cin = cout
    return head


def build_plan(num_keypoints: int, downsample_factor: int) -> Plan:
    """Layer list + flat offsets, parameters in torchvision ``state_dict`` order (backbone first, then head)."""
    if num_keypoints > CPAD:
        raise NotImplementedError(f"at most {CPAD} heat-map channels (keypoints x views) are supported, got {num_keypoints}")
    stem = ConvP("backbone.0", "stem", 3, 64, 7, 2, 3, 64, 4)
    stem_bn = BNP("backbone.1", 64)
    blocks: list[Block] = []
    inplanes = 64
    for li, (planes, nblk, stride) in enumerate(((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))):
        for bi in range(nblk):
            pre = f"backbone.{4 + li}.{bi}"
            st = stride if bi == 0 else 1
            blk = Block(
                ConvP(f"{pre}.conv1", "conv", inplanes, planes, 1, 1, 0, planes, inplanes), BNP(f"{pre}.bn1", planes),
                ConvP(f"{pre}.conv2", "conv", planes, planes, 3, st, 1, planes, planes), BNP(f"{pre}.bn2", planes),
                ConvP(f"{pre}.conv3", "conv", planes, planes * 4, 1, 1, 0, planes * 4, planes), BNP(f"{pre}.bn3", planes * 4))
            if bi == 0 and (st != 1 or inplanes != planes * 4):
                blk.down = ConvP(f"{pre}.downsample.0", "conv", inplanes, planes * 4, 1, st, 0, planes * 4, inplanes)
                blk.dbn = BNP(f"{pre}.downsample.1", planes * 4)
            blocks.append(blk)
            inplanes = planes * 4
    head = build_head(2048, 32, num_keypoints, downsample_factor)

    plan = Plan(stem, stem_bn, blocks, head)
    off = 0
    wd = 0
    run = 0

    def add_conv(c: ConvP):
        nonlocal off, wd
        c.w_off = off
        off += c.numel
        if c.kind != "stem":
            c.wd_off = wd
            wd += c.numel
        if c.kind == "convT":
            c.bias_off = off
            off += c.Ci
        plan.convs.append(c)

    def add_bn(b: BNP):
        nonlocal off, run
        b.g_off = off
        b.b_off = off + b.C
        off += 2 * b.C
        b.r_off = run
        run += 2 * b.C
        plan.bns.append(b)

    add_conv(stem)
    add_bn(stem_bn)
    for blk in blocks:
        add_conv(blk.conv1); add_bn(blk.bn1)
        add_conv(blk.conv2); add_bn(blk.bn2)
        add_conv(blk.conv3); add_bn(blk.bn3)
        if blk.down is not None:
            add_conv(blk.down); add_bn(blk.dbn)
    plan.n_backbone = off
    for c in head:
        add_conv(c)
    plan.n_total = off
    plan.n_wd = wd
    plan.n_running = run
    return plan


class Tape:
    """Tensors one forward pass leaves behind for its backward pass."""

    def __init__(self):
        self.t: dict[str, torch.Tensor] = {}
        self.meta: dict[str, object] = {}


class Engine:
    # weight gradients on a side stream (measured on the ResNet step: +2 %; on the ViT step, whose main stream is already
    # MFMA-bound, -8 %, so ViTEngine turns it off)
    wgrad_side_stream = True
    dgrad_mask_bits = True
    bn_bwd_ds = True
    bn_apply_rbn = True
    residual_fp32 = False
    dgrad_half_addend = True

    final_softmax = True   # (HeadEngine: HeatmapHead(final_softmax=False) leaves the last layer's output un-normalised)

    def __init__(self, num_keypoints: int, downsample_factor: int = 2, device: torch.device | str = "cuda:0", plan: Plan | None = None):
        self.device = torch.device(device)
        ops.require_device_type(self.device)
        _lib.lib()  # fail loudly now if liblp_hip.so is missing
        self.K = num_keypoints
        self.ds = downsample_factor
        self.plan = build_plan(num_keypoints, downsample_factor) if plan is None else plan
        n = self.plan.n_total
        dev = self.device
        self.P = torch.zeros(n, device=dev, dtype=torch.float32)      # master parameters
        self.G = torch.zeros(n, device=dev, dtype=torch.float32)      # gradients
        self.Wb = torch.zeros(n, device=dev, dtype=torch.bfloat16)    # bf16 operand copy (same offsets)
        self.Wd = torch.zeros(self.plan.n_wd, device=dev, dtype=torch.bfloat16)  # [Ci][R][S][Co] copies for dgrad
        self.R = torch.zeros(self.plan.n_running, device=dev, dtype=torch.float32)
        for b in self.plan.bns:
            self.P[b.g_off:b.g_off + b.C] = 1.0
            self.R[b.r_off + b.C:b.r_off + 2 * b.C] = 1.0
        self.nbt = torch.zeros((), dtype=torch.long)  # shared num_batches_tracked of every BatchNorm
        self.sync_bn = False          # set by the DDP wrapper when world_size > 1 (reference: train.py:427)
        # gradient all-reduce overlapped with backward (DataParallel): called with the lowest flat offset whose gradients are FINAL, as the
        # backward pass moves from the head towards the stem; only when the step has ONE backward pass (single_backward: the joint pass
        # of a semi-supervised step, or a supervised step), since two passes accumulate into the same buffer
        self.grad_progress = None
        self.single_backward = False
        self._bwd_training = True     # mode of the forward pass whose backward is running (eval: no batch-statistics terms)
        self.sync_bn_messages = 0     # SyncBatchNorm all-reduces issued so far (bench.py reports them per step)
        self.process_group = None
        # SyncBatchNorm transport.  Default: one SUM all-reduce per message.  LP_SYNCBN_GATHER=1: one-shot exchange - every rank's
        # [segments][2][C] sums are all-gathered (over xGMI each rank writes its 4 - 16 KB straight into every peer's slot: one hop, no
        # ring) and added locally in RANK ORDER, so every rank holds the same bits whatever the collective's internal order
        self.sync_bn_gather = os.environ.get("LP_SYNCBN_GATHER", "0") == "1"
        # lp_bn_bwd_apply in two launches (the correction terms converted by a one-thread-per-value kernel into a small workspace, round 5) or
        # - LP_BN_BWD_TERMS=0, A/B runs - in the self-contained form that converts them per workgroup into LDS
        self.bn_bwd_terms = os.environ.get("LP_BN_BWD_TERMS", "1") != "0"
        self.dgrad_half_addend = os.environ.get("LP_DGRAD_HALF_ADDEND", "1") != "0"   # (0: conv1 first, the shortcut accumulates in place, stand-alone reduction)
        self.bn_apply_rbn = os.environ.get("LP_BN_APPLY_RBN", "1") != "0"   # (0: the projection shortcut normalised by a pass of its own, lp_bn_apply_seg)
        self.bn_bwd_ds = os.environ.get("LP_BN_BWD_DS", "1") != "0"   # (0: the projection shortcut's BatchNorm reductions as a pass of their own, lp_bn_bwd_reduce)
        # The optional "fp32 residual stream" policy (round 6): block outputs that the next block adds as an identity shortcut
        # are kept as a bf16 pair (hi + lo), a projection shortcut is added unrounded.  NOT the benchmarked default: +3.9 % of the step's HBM time.
        self.residual_fp32 = os.environ.get("LP_RESIDUAL_FP32", "0") == "1"
        self.dgrad_mask_bits = os.environ.get("LP_DGRAD_MASK_BITS", "1") != "0"   # (0: the two data gradients into a layer's first block read the bf16 activation as mask)
        self._gather_buf: torch.Tensor | None = None
        self._lib = _lib.lib()
        self.profile: list | None = None  # bench.py: [(kernel tag, algorithmic flops, start event, end event)]
        self._wgrad_ws: torch.Tensor | None = None  # split-K partial tiles of the weight-gradient kernels
        self._red_ws: torch.Tensor | None = None    # per-workgroup rows of the stand-alone BatchNorm reductions
        self._side = None                            # side stream of the weight-gradient launches (created on first use)
        self._side_busy = False
        self._side_keep: list = []                   # operands of the side-stream launches in flight (released at the join)
        self._fold: tuple[torch.Tensor, torch.Tensor] | None = None  # inference copies: BatchNorm folded into (bf16 weights, biases)

    def _zeros_f32(self, n: int) -> torch.Tensor:
        """n zeroed fp32 words for a reduction target.  Inside backward() they are carved from ONE arena zeroed with a single fill (the
        step had ~25 separate fills for its BatchNorm / bias sums); outside, an ordinary allocation."""
        ar = getattr(self, "_zero_arena", None)
        if ar is not None and self._zero_off + n <= ar.numel():
            t = ar[self._zero_off:self._zero_off + n]
            self._zero_off += (n + 3) // 4 * 4   # (16-B aligned pieces)
            return t
        return torch.zeros(n, device=self.device, dtype=torch.float32)

    def _zeros_fx(self, n: int) -> torch.Tensor:
        """n zeroed fixed-point sums (include/lp_hip.h: lp_fxsum = two int64 words each) as an int64 tensor of 2 n words: the targets of
        every BatchNorm reduction.  Integer addition commutes, so the totals - and an all-reduce over them - do not depend on the order
        workgroups or ranks arrive in."""
        return self._zeros_f32(4 * n).view(torch.int64)

    def _timed(self, tag: str, flops: float, fn, nbytes: float = 0.0, layer: str = ""):
        """Run one kernel launch; when profiling is on, bracket it with HIP events on the launch stream.  ``nbytes``: the launch's
        ALGORITHMIC bytes (operands read once + result written once), the denominator the measured HBM traffic is held against.
        ``layer``: the parameter name of the layer the launch belongs to (profiles/layer_table.py groups by it, not by launch order)."""
        if self.profile is None:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        if tag.startswith("conv_") and hasattr(self._lib, "lp_conv_last_kernel"):   # label the launch with the kernel that actually ran
            k = self._lib.lp_conv_last_kernel()
            if k == _lib.CONV_KERNEL_PIPE:
                tag = tag.replace("conv_igemm_kernel", "conv_pipe_kernel")
            elif k == _lib.CONV_KERNEL_PIPE_HALO:
                tag = tag.replace("conv_igemm_kernel", "conv_pipe_kernel").replace(">", ",halo>")
            elif k == _lib.CONV_KERNEL_RES2D:
                tag = tag.replace("conv_igemm_kernel<64,", "conv_res2d_kernel<").replace("conv_igemm_kernel<64>", "conv_res2d_kernel<fwd>")
            elif k == _lib.CONV_KERNEL_WGRAD_PIPE:
                tag = tag.replace("conv_wgrad_kernel", "conv_wgrad_pipe_kernel")
        self.profile.append((tag, flops, e0, e1, nbytes, layer))
        return out

    def _wgrad(self, x, dy, g, dw: torch.Tensor, stem: bool = False, dbias: torch.Tensor | None = None) -> None:
        """Weight gradient of one layer.  Nothing in the backward pass consumes it, so on the device it runs on a SIDE stream,
        ordered after the kernel that produced ``dy``: its MFMA-bound tiles fill the CUs that the data-gradient's tails and the
        HBM-bound BatchNorm kernels leave idle.  All weight gradients share that stream (and the split-K workspace);
        ``_join_side_stream`` orders the main stream after them at the end of backward()."""
        need = self._lib.lp_conv_wgrad_workspace_bytes(C.byref(g), 0)
        if self._wgrad_ws is None or self._wgrad_ws.numel() < need:
            self._join_side_stream()  # the old workspace may still be in use
            self._wgrad_ws = torch.empty(max(need, 96 << 20), device=self.device, dtype=torch.uint8)
        fn = self._lib.lp_stem_wgrad if stem else self._lib.lp_conv_wgrad
        what = "lp_stem_wgrad" if stem else "lp_conv_wgrad"
        if dbias is not None:  # the layer's bias gradient (column sums of dy) rides in the same launch
            what = "lp_conv_wgrad_bias"
            fn = lambda x_, dy_, g_, dw_, split, ws, nws, st: self._lib.lp_conv_wgrad_bias(x_, dy_, g_, dw_, _p(dbias), split, ws, nws, st)  # noqa: E731
        # (while bench.py's per-launch events are on, the launch stays on the main stream: an event pair there brackets exactly this
        # kernel, as rocprofv3's serialised kernel trace does; on the side stream it would bracket nothing)
        if (self.device.type != "cuda" or not self.wgrad_side_stream or self.profile is not None
                or os.environ.get("LP_WGRAD_SIDE_STREAM", "1") == "0"):
            check(fn(_p(x), _p(dy), C.byref(g), _p(dw), 0, _p(self._wgrad_ws), self._wgrad_ws.numel(), ops._stream()), what)
            return
        if self._side is None:
            # (its priority was A/B'd in round 5 - higher than the main stream's, equal: no difference, profiles/r05f_overlap.txt)
            self._side = torch.cuda.Stream(device=self.device)
        main = torch.cuda.current_stream(self.device)
        self._side.wait_stream(main)          # dy (and the zeroed / partially accumulated G) are ready
        # keep the operands alive until the main stream has joined the side stream (_join_side_stream): their memory then returns to the
        # allocator in main-stream order.  (Not Tensor.record_stream: blocks freed that way only become reusable once an event on the
        # side stream has COMPLETED, and with the host running a step or two ahead of the device that never happens in time - the caching
        # allocator then keeps calling hipMalloc: 23 per step, 141 GB reserved for a 28 GB working set, and 2 runs in 6 five times slower.)
        self._side_keep.append((x, dy))
        with torch.cuda.stream(self._side):
            check(fn(_p(x), _p(dy), C.byref(g), _p(dw), 0, _p(self._wgrad_ws), self._wgrad_ws.numel(), ops._stream()), what)
        self._side_busy = True

    def _join_side_stream(self) -> None:
        if getattr(self, "_side_busy", False):
            torch.cuda.current_stream(self.device).wait_stream(self._side)
            self._side_busy = False
            self._side_keep.clear()  # whatever reuses this memory is enqueued on the main stream behind the wait

    @staticmethod
    def _bytes(c: "ConvP", g, wgrad: bool = False) -> float:
        """Algorithmic bytes of one contraction over this layer: the gathered tensor and the other activation-sized operand / result once
        each (bf16), the weights once (bf16 operand; fp32 for the gradient a weight-gradient launch accumulates into)"""
        kk = 8 * 8 if c.kind == "stem" else c.k * c.k
        acts = 2.0 * g.B * (g.Hi * g.Wi * g.Ci + g.Ho * g.Wo * g.Co)
        return acts + (4.0 if wgrad else 2.0) * c.Co * kk * c.Ci

    @staticmethod
    def _flops(c: "ConvP", g) -> float:
        """Algorithmic FLOPs of one contraction over this layer (2 x MACs, logical channels, no padding)."""
        kk = c.k * c.k
        if c.kind == "convT":
            return 2.0 * g.B * g.Ho * g.Wo * c.cin * kk * c.cout
        return 2.0 * g.B * g.Ho * g.Wo * c.cout * kk * c.cin

    # ------------------------------------------------------------------------------------------------ params
    def param_view(self, c: ConvP | BNP, which: str = "weight", buf: torch.Tensor | None = None) -> torch.Tensor:
        """torch-shaped (state_dict compatible) view into a flat buffer (parameters by default)."""
        buf = self.P if buf is None else buf
        if isinstance(c, BNP):
            off = c.g_off if which == "weight" else c.b_off
            return buf[off:off + c.C]
        if which == "bias":
            return buf[c.bias_off:c.bias_off + c.cout]
        flat = buf[c.w_off:c.w_off + c.numel]
        if c.kind == "stem":
            return flat.view(64, 8, 8, 4)[:, :7, :7, :3].permute(0, 3, 1, 2)
        if c.kind == "convT":  # torch ConvTranspose2d weight (cin, cout, kh, kw)
            return flat.view(c.Co, 3, 3, c.Ci)[:c.cin, :, :, :c.cout].permute(0, 3, 1, 2)
        return flat.view(c.Co, c.k, c.k, c.Ci).permute(0, 3, 1, 2)

    def running_view(self, b: BNP, which: str) -> torch.Tensor:
        off = b.r_off if which == "running_mean" else b.r_off + b.C
        return self.R[off:off + b.C]

    def state_dict(self) -> dict[str, torch.Tensor]:
        """Reference-compatible keys/shapes: backbone.{0,1,4..7}.*, head.upsampling_layers.{i}.{weight,bias}."""
        sd: dict[str, torch.Tensor] = {}
        plan = self.plan

        def put_bn(b: BNP):
            sd[f"{b.name}.weight"] = self.param_view(b, "weight")
            sd[f"{b.name}.bias"] = self.param_view(b, "bias")
            sd[f"{b.name}.running_mean"] = self.running_view(b, "running_mean")
            sd[f"{b.name}.running_var"] = self.running_view(b, "running_var")
            sd[f"{b.name}.num_batches_tracked"] = self.nbt

        sd[f"{plan.stem.name}.weight"] = self.param_view(plan.stem)
        put_bn(plan.stem_bn)
        for blk in plan.blocks:
            for c, b in ((blk.conv1, blk.bn1), (blk.conv2, blk.bn2), (blk.conv3, blk.bn3)):
                sd[f"{c.name}.weight"] = self.param_view(c)
                put_bn(b)
            if blk.down is not None:
                sd[f"{blk.down.name}.weight"] = self.param_view(blk.down)
                put_bn(blk.dbn)
        for c in plan.head:
            sd[f"{c.name}.weight"] = self.param_view(c)
            sd[f"{c.name}.bias"] = self.param_view(c, "bias")
        return sd

    @torch.no_grad()
    def load_state_dict(self, sd: dict[str, torch.Tensor], strict: bool = True) -> None:
        own = self.state_dict()
        missing = [k for k in own if k not in sd]
        unexpected = [k for k in sd if k not in own]
        if strict and (missing or unexpected):
            raise KeyError(f"state_dict mismatch: missing {missing[:5]}..., unexpected {unexpected[:5]}...")
        for k, dst in own.items():
            if k not in sd:
                continue
            if k.endswith("num_batches_tracked"):
                self.nbt.fill_(int(sd[k]))
                continue
            dst.copy_(sd[k].to(device=self.device, dtype=torch.float32))
        self.refresh_weight_copies()

    def invalidate_inference_copies(self) -> None:
        """Parameters or running statistics changed: the folded inference weights are rebuilt on the next forward_infer()."""
        self._fold = None

    def refresh_weight_copies(self, lo: int = 0, hi: int | None = None) -> None:
        """bf16 operand copy of P[lo:hi] and the transposed copies of the conv weights inside that range."""
        self._fold = None
        hi = self.plan.n_total if hi is None else hi
        check(self._lib.lp_cast_bf16(_p(self.P[lo:hi]), hi - lo, _p(self.Wb[lo:hi]), ops._stream()), "lp_cast_bf16")
        self.refresh_dgrad_copies(lo, hi)

    def refresh_dgrad_copies(self, lo: int = 0, hi: int | None = None) -> None:
        hi = self.plan.n_total if hi is None else hi
        for c in self.plan.convs:
            if c.kind == "stem" or not (lo <= c.w_off < hi):
                continue
            check(self._lib.lp_permute_cba(_p(self.Wb[c.w_off:]), c.Co, c.k * c.k, c.Ci, _p(self.Wd[c.wd_off:]), ops._stream()),
                  "lp_permute_cba")

    def zero_grad(self) -> None:
        self.G.zero_()

    # ------------------------------------------------------------------------------------------------ kernels
    def _geom(self, c: ConvP, B: int, Hi: int, Wi: int) -> _lib.ConvGeom:
        if c.kind == "convT":  # mirrored conv: input = the (2h x 2w) ConvT output, output = the ConvT input
            return _lib.ConvGeom(B, 2 * Hi, 2 * Wi, c.Ci, Hi, Wi, c.Co, 3, 3, 2, 1)
        Ho = (Hi + 2 * c.pad - c.k) // c.stride + 1
        Wo = (Wi + 2 * c.pad - c.k) // c.stride + 1
        return _lib.ConvGeom(B, Hi, Wi, c.Ci, Ho, Wo, c.Co, c.k, c.k, c.stride, c.pad)

    def _reduce_ws(self, M: int, C_: int) -> torch.Tensor:
        """Scratch rows of the stand-alone reductions (lp_bn_stats, lp_bn_bwd_reduce): one buffer, reused in stream order"""
        need = int(self._lib.lp_bn_reduce_workspace_bytes(M, C_))
        if getattr(self, "_red_ws", None) is None or self._red_ws.numel() < need:   # (ViTEngine shares the head code with its own __init__)
            self._red_ws = torch.empty(max(need, 4 << 20), device=self.device, dtype=torch.uint8)
        return self._red_ws

    def _bn_fuse(self, sums: torch.Tensor, b: BNP | None = None, z=None, mean=None, invstd=None, mask_from_z: bool = False,
                 relu_bits=None, seg: int = 0, addend_half: bool = False) -> _lib.BnFuse:
        """lp_bn_fuse for one launch: ``sums`` alone = the forward form ([sum z, sum z^2] of the output); with ``b`` / ``z`` / ``mean`` /
        ``invstd`` the backward form ([sum dz, sum dz * xhat] of the gradient the launch produces)."""
        f = _lib.BnFuse()
        f.addend_half = int(addend_half)
        f.sums = sums.data_ptr()
        f.seg_images = seg
        if b is not None:
            f.z, f.mean, f.invstd = z.data_ptr(), mean.data_ptr(), invstd.data_ptr()
            f.gamma, f.beta = self.param_view(b, "weight").data_ptr(), self.param_view(b, "bias").data_ptr()
            f.mask_from_z = int(mask_from_z)
            f.relu_bits = relu_bits.data_ptr() if relu_bits is not None else None
        return f

    def _conv_fwd(self, c: ConvP, x: torch.Tensor, B: int, Hi: int, Wi: int, sums: torch.Tensor | None = None, seg: int = 0):
        """``sums`` (segments,2,Co) fixed point: also accumulate [sum z, sum z^2] of the output there (the next BatchNorm's statistics
        pass, fused); ``seg`` > 0: images [0, seg) and [seg, B) are two BatchNorm segments with their own sums."""
        g = self._geom(c, B, Hi, Wi)
        out = torch.empty(B, g.Ho, g.Wo, c.Co, device=self.device, dtype=torch.bfloat16)
        w = self.Wb[c.w_off:]
        st = ops._stream()
        if c.kind == "stem":
            if sums is None:
                run = lambda: check(self._lib.lp_stem_fwd(_p(x), _p(w), C.byref(g), _p(out), st), "lp_stem_fwd")  # noqa: E731
            else:
                f = self._bn_fuse(sums, seg=seg)
                run = lambda: check(self._lib.lp_stem_fwd_bn(_p(x), _p(w), C.byref(g), _p(out), C.byref(f), st), "lp_stem_fwd_bn")  # noqa: E731
            self._timed("conv_igemm_kernel<64,stem>", self._flops(c, g), run, self._bytes(c, g), layer=c.name)
        else:
            if sums is None:
                run = lambda: check(self._lib.lp_conv_fwd(_p(x), _p(w), C.byref(g), None, _p(out), None, c.Co, 0, st), "lp_conv_fwd")  # noqa: E731
            else:
                f = self._bn_fuse(sums, seg=seg)
                run = lambda: check(self._lib.lp_conv_fwd_bn(_p(x), _p(w), C.byref(g), _p(out), C.byref(f), st), "lp_conv_fwd_bn")  # noqa: E731
            self._timed(f"conv_igemm_kernel<{128 if c.Co > 64 else 64},fwd>", self._flops(c, g), run, self._bytes(c, g), layer=c.name)
        return out, g

    @staticmethod
    def _segments(B: int, seg: int) -> list[tuple[int, int]]:
        """[(first image, images)] of the BatchNorm segments of a pass over B images"""
        return [(0, B)] if not seg else [(0, seg), (seg, B - seg)]

    def can_segment(self, n0: int, H: int, W: int) -> bool:
        """Can a pass over n0 + n1 images of H x W keep two BatchNorm segments in ONE launch per layer?  The segment boundary must
        fall on a 128-row tile boundary of every fused launch; the smallest per-image row count (the trunk's output map, which is also
        what a parity class of the stride-2 data gradients covers) decides, every other map is 4^k times larger."""
        return n0 > 0 and H % 32 == 0 and W % 32 == 0 and (n0 * (H // 32) * (W // 32)) % 128 == 0

    def _sync_stats(self, t: torch.Tensor) -> None:
        """SUM of one SyncBatchNorm message ((segments, 2, C) fixed-point sums = int64 words) over the ranks, in place (reference:
        ``sync_batchnorm=True``, train.py:427).  Integer sums: every rank ends with the same bits whatever order the collective adds in.
        An all-reduce, or - LP_SYNCBN_GATHER=1 - the one-shot form: all-gather the ranks' messages, add the rows locally."""
        if self.sync_bn_gather:
            if not t.is_contiguous():
                raise ValueError("a SyncBatchNorm message must be a contiguous buffer (the sum is written back through it)")
            world, n = dist.get_world_size(self.process_group), t.numel()
            if self._gather_buf is None or self._gather_buf.numel() < world * n:   # sized once for the widest layer, two segments
                widest = max([2 * b.C for b in self.plan.bns] + [2 * self.plan.stem_bn.C]) * 2 * 2
                self._gather_buf = torch.empty(world * max(n, widest), device=t.device, dtype=t.dtype)
            flat = self._gather_buf[:world * n]
            dist.all_gather_into_tensor(flat, t.reshape(-1), group=self.process_group)   # (flat output: the form gloo and RCCL both take)
            torch.sum(flat.view(world, n), dim=0, out=t.view(-1))
        else:
            dist.all_reduce(t, group=self.process_group)

    def _bn_moments(self, b: BNP, z: torch.Tensor, M: int, training: bool, sums: torch.Tensor, have_sums: bool = False, seg: int = 0):
        """-> (mean, invstd) of this pass, each (segments, C) flattened: batch statistics (running statistics updated, segment by
        segment) in training, running statistics otherwise"""
        B = z.shape[0]
        rpi = M // B
        segs = self._segments(B, seg if training else 0)
        mean = torch.empty(len(segs) * b.C, device=self.device, dtype=torch.float32)
        invstd = torch.empty_like(mean)
        if training:
            if not have_sums:
                rws = self._reduce_ws(M, b.C)
                for si, (i0, n) in enumerate(segs):
                    check(self._lib.lp_bn_stats(_p(z[i0:i0 + n]), n * rpi, b.C, _p(sums[si * 4 * b.C:]), _p(rws), rws.numel(), ops._stream()),
                          "lp_bn_stats")
            counts = [float(n * rpi) for _, n in segs]
            if self.sync_bn:  # ONE message carries every segment's [sum, sum of squares]
                self._sync_stats(sums)
                self.sync_bn_messages += 1
                counts = [c_ * dist.get_world_size(self.process_group) for c_ in counts]
            rm, rv = _p(self.running_view(b, "running_mean")), _p(self.running_view(b, "running_var"))
            if len(segs) == 1:
                check(self._lib.lp_bn_finalize(_p(sums), counts[0], b.C, BN_EPS, BN_MOMENTUM, _p(mean), _p(invstd), rm, rv, ops._stream()),
                      "lp_bn_finalize")
            else:
                check(self._lib.lp_bn_finalize2(_p(sums), counts[0], counts[1], b.C, BN_EPS, BN_MOMENTUM, _p(mean), _p(invstd), rm, rv,
                                                ops._stream()), "lp_bn_finalize2")
        else:
            mean.copy_(self.running_view(b, "running_mean"))
            invstd.copy_((self.running_view(b, "running_var") + BN_EPS).rsqrt())
        return mean, invstd

    def _bn_fwd(self, b: BNP, z: torch.Tensor, M: int, residual: torch.Tensor | None, relu: bool, training: bool, sums: torch.Tensor,
                have_sums: bool = False, want_bits: bool = False, seg: int = 0, residual_bn=None, pair=None):
        """-> (y, mean, invstd[, relu_bits]); ``want_bits``: also the 1-bit ReLU mask (M*C/8 bytes) for the backward pass.
        ``residual_bn`` = (BNP, mean_d, invstd_d): ``residual`` is the PRE-normalisation tensor of the block's projection shortcut, normalised
        inside this pass (lp_bn_apply_seg_rbn) instead of by a pass of its own that writes the normalised shortcut and this one reads back."""
        mean, invstd = self._bn_moments(b, z, M, training, sums, have_sums, seg)
        y = torch.empty_like(z)
        bits = torch.empty(M * b.C // 8, device=self.device, dtype=torch.uint8) if want_bits else None
        B = z.shape[0]
        rpi = M // B
        gam, bet = _p(self.param_view(b, "weight")), _p(self.param_view(b, "bias"))
        # both segments in ONE launch (the kernel walks each segment with that segment's terms in registers)
        if pair is not None:
            # fp32-residual policy: ``pair`` = (residual_lo or None, y_lo or None): lp_bn_apply_seg_lo adds hi + lo (or the unrounded
            # normalised shortcut) and leaves this output's own lo word for the next identity block
            res_lo, y_lo = pair
            if residual_bn is not None:
                bd, md, vd = residual_bn
                check(self._lib.lp_bn_apply_seg_lo(_p(z), _p(mean), _p(invstd), gam, bet, None, None, _p(residual), _p(md), _p(vd),
                                                   _p(self.param_view(bd, "weight")), _p(self.param_view(bd, "bias")), int(relu), M, b.C,
                                                   (seg if training else 0) * rpi, _p(y), _p(y_lo), _p(bits), ops._stream()), "lp_bn_apply_seg_lo")
            else:
                check(self._lib.lp_bn_apply_seg_lo(_p(z), _p(mean), _p(invstd), gam, bet, _p(residual), _p(res_lo), None, None, None, None, None,
                                                   int(relu), M, b.C, (seg if training else 0) * rpi, _p(y), _p(y_lo), _p(bits), ops._stream()),
                      "lp_bn_apply_seg_lo")
        elif residual_bn is not None:
            bd, md, vd = residual_bn
            check(self._lib.lp_bn_apply_seg_rbn(_p(z), _p(mean), _p(invstd), gam, bet, _p(residual), _p(md), _p(vd), _p(self.param_view(bd, "weight")),
                                                _p(self.param_view(bd, "bias")), int(relu), M, b.C, (seg if training else 0) * rpi, _p(y), _p(bits),
                                                ops._stream()), "lp_bn_apply_seg_rbn")
        else:
            check(self._lib.lp_bn_apply_seg(_p(z), _p(mean), _p(invstd), gam, bet, _p(residual), int(relu), M, b.C,
                                            (seg if training else 0) * rpi, _p(y), _p(bits), ops._stream()), "lp_bn_apply_seg")
        if want_bits:
            return y, mean, invstd, bits
        return y, mean, invstd

    # ------------------------------------------------------------------------------------------------ head
    def _head_forward(self, x: torch.Tensor, B: int, h: int, w: int, T: dict) -> torch.Tensor:
        """features (B,h,w,C) bf16 NHWC -> heat-maps (B,K,.,.) fp32: PixelShuffle(2) -> ConvTranspose2d x n -> spatial softmax
        (reference models/heads/heatmap.py:203-212).  Leaves "head.in{i}" and "heat" on the tape."""
        head = self.plan.head
        cs = head[0].cin                       # channels after the pixel shuffle
        ld = head[0].Co                        # ... stored with this pitch (zero padded up to a multiple of 64)
        alloc = torch.zeros if ld != cs else torch.empty
        ps = alloc(B, 2 * h, 2 * w, ld, device=self.device, dtype=torch.bfloat16)
        check(self._lib.lp_pixel_shuffle(_p(x), B, h, w, cs, ld, 0, _p(ps), ops._stream()), "lp_pixel_shuffle")
        h, w = 2 * h, 2 * w
        T["head.in0"] = ps
        cur = ps
        logits = None
        for li, c in enumerate(head):
            g = self._geom(c, B, h, w)
            last = li == len(head) - 1
            bias = self.P[c.bias_off:c.bias_off + c.Ci]
            wd = self.Wd[c.wd_off:]
            if last:
                logits = torch.empty(B, 2 * h, 2 * w, CPAD, device=self.device, dtype=torch.float32)
                check(self._lib.lp_conv_dgrad(_p(cur), _p(wd), C.byref(g), _p(bias), None, None, None, _p(logits), CPAD, self.K, 0, ops._stream()),
                      "lp_conv_dgrad(head)")
            else:   # (intermediate layer: c.Ci = its channels - deconv_out_channels, by default the keypoints - rounded up to 64, zeros behind them)
                nxt = torch.empty(B, 2 * h, 2 * w, c.Ci, device=self.device, dtype=torch.bfloat16)
                check(self._lib.lp_conv_dgrad(_p(cur), _p(wd), C.byref(g), _p(bias), None, None, _p(nxt), None, c.Ci, c.Ci, 0, ops._stream()),
                      "lp_conv_dgrad(head)")
                cur = nxt
                T[f"head.in{li + 1}"] = cur
            h, w = 2 * h, 2 * w
        n = h * w
        if not self.final_softmax:   # HeatmapHead(final_softmax=False), reference :209-211: the last layer's output as it is, (B, K, h, w)
            heat = logits[..., :self.K].permute(0, 3, 1, 2).contiguous()
            T["heat"] = heat
            return heat
        heat = torch.empty(B, self.K, h, w, device=self.device, dtype=torch.float32)
        check(self._lib.lp_softmax2d_fwd(_p(logits), n * CPAD, CPAD, 1, B, self.K, n, _p(heat), ops._stream()), "lp_softmax2d_fwd")
        T["heat"] = heat
        return heat

    def _head_backward(self, T: dict, B: int, g_heat: torch.Tensor) -> torch.Tensor:
        """d loss / d heat-maps -> d loss / d features (B,h,w,C) bf16; head parameter gradients accumulate into G."""
        head = self.plan.head
        heat = T["heat"]
        _, K, h, w = heat.shape
        n = h * w
        g_heat = g_heat.to(torch.float32).contiguous()
        # (lp_softmax2d_bwd's pixel-major kernel - K <= 32 maps - writes every channel of a pixel's row, the pad channels as zeros: no fill needed)
        dcur = (torch.empty if (self.final_softmax and K <= 32) else torch.zeros)(B, h, w, CPAD, device=self.device, dtype=torch.bfloat16)
        if self.final_softmax:
            check(self._lib.lp_softmax2d_bwd(_p(heat), _p(g_heat), B, K, n, _p(dcur), n * CPAD, CPAD, 1, ops._stream()), "lp_softmax2d_bwd")
        else:
            dcur[..., :K] = g_heat.permute(0, 2, 3, 1)
        # last layer first.  ConvT backward-data is the mirrored conv's forward; its wgrad is lp_conv_wgrad.
        for li in range(len(head) - 1, -1, -1):
            c = head[li]
            hs, ws = h // 2, w // 2
            g = self._geom(c, B, hs, ws)
            x_small = T[f"head.in{li}"]
            bsum = self._zeros_fx(2 * c.Ci)
            rws = self._reduce_ws(B * h * w, c.Ci)
            check(self._lib.lp_bn_stats(_p(dcur), B * h * w, c.Ci, _p(bsum), _p(rws), rws.numel(), ops._stream()), "lp_bn_stats(bias)")
            check(self._lib.lp_fxsum_accumulate(_p(bsum), c.Ci, _p(self.G[c.bias_off:]), ops._stream()), "lp_fxsum_accumulate")
            self._wgrad(dcur, x_small, g, self.G[c.w_off:])
            dx = torch.empty(B, hs, ws, c.Co, device=self.device, dtype=torch.bfloat16)
            check(self._lib.lp_conv_fwd(_p(dcur), _p(self.Wb[c.w_off:]), C.byref(g), None, _p(dx), None, c.Co, 0, ops._stream()),
                  "lp_conv_fwd(head bwd)")
            dcur, h, w = dx, hs, ws
        fh, fw = h // 2, w // 2
        cs, ld = head[0].cin, head[0].Co
        d = torch.empty(B, fh, fw, 4 * cs, device=self.device, dtype=torch.bfloat16)
        check(self._lib.lp_pixel_shuffle(_p(dcur), B, fh, fw, cs, ld, 1, _p(d), ops._stream()), "lp_pixel_shuffle(inv)")
        return d

    # ------------------------------------------------------------------------------------------------ forward
    def forward(self, images, training: bool = True) -> tuple[torch.Tensor, Tape]:
        """images (B,3,H,W) fp32 NCHW -> heat-maps (B,K,H/2^ds,W/2^ds) fp32, plus the tape for backward().

        ``images`` may be a pair of tensors (the labeled and the unlabeled frames of a semi-supervised step, reference
        models/base.py:682-695): both go through every layer in ONE launch, as two BatchNorm segments that keep their own batch
        statistics and update the running statistics in order - the result of the reference's two forward calls at the launch count
        and tile fill of one (check ``can_segment`` first)."""
        parts = list(images) if isinstance(images, (list, tuple)) else [images]
        for p_ in parts:
            ops.require_device(p_)
        parts = [p_.to(torch.float32).contiguous() for p_ in parts]
        if parts[0].dim() != 4 or parts[0].shape[1] != 3:
            raise ValueError(f"images must be (B, 3, H, W), got {tuple(parts[0].shape)}")   # (raw pointers from here on)
        _, _, H, W = parts[0].shape
        if H % 32 or W % 32:
            raise ValueError(f"image size must be a multiple of 32, got {H}x{W}")
        if len(parts) > 2 or any(p_.shape[1:] != parts[0].shape[1:] for p_ in parts):
            raise ValueError("a joint pass takes at most two batches of equally sized images")
        B = sum(p_.shape[0] for p_ in parts)
        seg = parts[0].shape[0] if (len(parts) == 2 and training) else 0
        if seg and not self.can_segment(seg, H, W):
            raise NotImplementedError(f"{seg} images of {H}x{W} do not end on a tile boundary in every layer: run the two batches separately")
        tp = Tape()
        T = tp.t
        plan = self.plan
        nseg = 2 if seg else 1
        n_bn = sum(2 * b.C for b in plan.bns) * nseg
        sums_all = torch.zeros(2 * n_bn, device=self.device, dtype=torch.int64)   # fixed-point sums: two int64 words each (lp_fxsum)
        so = [0]

        def next_sums(b: BNP) -> torch.Tensor:
            s = sums_all[so[0]:so[0] + 4 * b.C * nseg]
            so[0] += 4 * b.C * nseg
            return s

        x4 = torch.empty(B, H, W, 4, device=self.device, dtype=torch.bfloat16)
        i0 = 0
        for p_ in parts:
            check(self._lib.lp_images_to_nhwc4(_p(p_), p_.shape[0], H, W, _p(x4[i0:]), ops._stream()), "lp_images_to_nhwc4")
            i0 += p_.shape[0]
        T["x4"] = x4
        def conv_bn(c: ConvP, b: BNP, xin, hh, ww, residual, relu, bits_key=None, residual_bn=None, apply=True, pair=None):
            """conv -> BatchNorm(+residual)(+ReLU); in training the statistics come out of the convolution's store pass.
            ``bits_key``: keep the output's 1-bit ReLU mask on the tape (block outputs: their backward reads it instead of the
            activation itself).  ``apply=False``: convolution and moments only (the projection shortcut, normalised later inside the
            block output's pass: ``residual_bn``) -> (z, None, mean, invstd, geometry)"""
            sums = next_sums(b)
            zz, gg = self._conv_fwd(c, xin, B, hh, ww, sums if training else None, seg=seg)
            if not apply:
                mm, vv = self._bn_moments(b, zz, B * gg.Ho * gg.Wo, training, sums, training, seg)
                return zz, None, mm, vv, gg
            res = self._bn_fwd(b, zz, B * gg.Ho * gg.Wo, residual, relu, training, sums, have_sums=training,
                               want_bits=bits_key is not None and training, seg=seg, residual_bn=residual_bn, pair=pair)
            if len(res) == 4:
                T[bits_key] = res[3]
            aa, mm, vv = res[:3]
            return zz, aa, mm, vv, gg

        # stem: conv -> [BatchNorm -> ReLU -> max-pool] in one pass; the full-resolution activation between them is never stored
        sb = plan.stem_bn
        s_sums = next_sums(sb)
        z, g = self._conv_fwd(plan.stem, x4, B, H, W, s_sums if training else None, seg=seg)
        h, w = g.Ho, g.Wo
        mu, iv = self._bn_moments(sb, z, B * h * w, training, s_sums, have_sums=training, seg=seg)
        T["stem.z"], T["stem.mu"], T["stem.iv"] = z, mu, iv
        ph, pw = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        x = torch.empty(B, ph, pw, 64, device=self.device, dtype=torch.bfloat16)
        T["pool.arg"] = torch.empty(B, ph, pw, 64, device=self.device, dtype=torch.uint8)
        for si, (i0, n) in enumerate(self._segments(B, seg)):
            check(self._lib.lp_bn_relu_maxpool_fwd(_p(z[i0:i0 + n]), _p(mu[si * 64:]), _p(iv[si * 64:]), _p(self.param_view(sb, "weight")),
                                                   _p(self.param_view(sb, "bias")), n, h, w, 64, _p(x[i0:i0 + n]), _p(T["pool.arg"][i0:i0 + n]),
                                                   ops._stream()), "lp_bn_relu_maxpool_fwd")
        h, w = ph, pw
        tp.meta["stem_hw"] = (g.Ho, g.Wo)

        x_lo = None   # fp32-residual policy: the lo word of x (the previous block's output), when the previous block left one
        for i, blk in enumerate(plan.blocks):
            key = f"b{i}"
            T[f"{key}.x"] = x
            tp.meta[f"{key}.hw"] = (h, w)
            z1, a1, m1, v1, _ = conv_bn(blk.conv1, blk.bn1, x, h, w, None, True)
            z2, a2, m2, v2, g2 = conv_bn(blk.conv2, blk.bn2, a1, h, w, None, True)
            ho, wo = g2.Ho, g2.Wo
            rbn = None
            if blk.down is not None:
                # the shortcut's BatchNorm is applied inside the block output's pass (its normalised tensor is never stored)
                zd, idt, md, vd, _ = conv_bn(blk.down, blk.dbn, x, h, w, None, False, apply=not self.bn_apply_rbn)
                T[f"{key}.zd"], T[f"{key}.md"], T[f"{key}.vd"] = zd, md, vd
                if self.bn_apply_rbn:
                    idt, rbn = zd, (blk.dbn, md, vd)
            else:
                idt = x
            pair = None
            if self.residual_fp32 and (rbn is not None or blk.down is None):
                # this output gets a lo word iff the NEXT block adds it as an identity shortcut (a layer's last block and the trunk's: nobody does)
                nxt_identity = i + 1 < len(plan.blocks) and plan.blocks[i + 1].down is None
                y_lo = torch.empty(B, ho, wo, blk.bn3.C, device=self.device, dtype=torch.bfloat16) if nxt_identity else None
                pair = (x_lo if blk.down is None else None, y_lo)
            z3, out, m3, v3, _ = conv_bn(blk.conv3, blk.bn3, a2, ho, wo, idt, True, bits_key=f"{key}.out_bits", residual_bn=rbn, pair=pair)
            x_lo = pair[1] if pair is not None else None
            if x_lo is not None:
                T[f"{key}.out_lo"] = x_lo   # (not needed by the backward pass: kept for inspection, freed with the tape)
            for nm, val in (("z1", z1), ("a1", a1), ("m1", m1), ("v1", v1), ("z2", z2), ("a2", a2), ("m2", m2), ("v2", v2),
                            ("z3", z3), ("m3", m3), ("v3", v3), ("out", out)):
                if val is not None:
                    T[f"{key}.{nm}"] = val
            x, h, w = out, ho, wo

        heat = self._head_forward(x, B, h, w, T)
        T["heat"] = heat
        tp.meta.update(B=B, H=H, W=W, training=training, seg=seg)
        if training:
            self.nbt += nseg  # one count per forward call of the reference
            self._fold = None  # the running statistics moved
        return heat, tp

    # ------------------------------------------------------------------------------------------------ inference
    def _folded(self) -> tuple[torch.Tensor, torch.Tensor]:
        """(Wf, Bf): bf16 weights with each BatchNorm's gamma / sqrt(running_var + eps) folded in (same offsets as Wb) and the
        matching per-channel biases (offsets of the BatchNorm's bias in P).  Rebuilt only after the parameters changed."""
        if self._fold is None:
            plan = self.plan
            wf = torch.empty_like(self.Wb)
            bf = torch.zeros(plan.n_total, device=self.device, dtype=torch.float32)
            pairs = []
            for blk in plan.blocks:
                pairs += [(blk.conv1, blk.bn1), (blk.conv2, blk.bn2), (blk.conv3, blk.bn3)]
                if blk.down is not None:
                    pairs.append((blk.down, blk.dbn))
            for c, b in pairs:
                check(self._lib.lp_bn_fold(_p(self.P[c.w_off:]), _p(self.param_view(b, "weight")), _p(self.param_view(b, "bias")),
                                           _p(self.running_view(b, "running_mean")), _p(self.running_view(b, "running_var")), BN_EPS,
                                           c.Co, c.k * c.k * c.Ci, _p(wf[c.w_off:]), _p(bf[b.b_off:]), ops._stream()), "lp_bn_fold")
            self._fold = (wf, bf)
        return self._fold

    def forward_infer(self, images: torch.Tensor) -> torch.Tensor:
        """Inference forward (predict_step, reference models/heatmap_tracker.py:155-191, with eval-mode BatchNorm): every
        conv -> BatchNorm [-> + identity] [-> ReLU] of the trunk is ONE launch on folded weights (lp_conv_fwd_act); no BatchNorm
        passes, no pre-normalisation tensors, no tape.  images (B,3,H,W) fp32 -> heat-maps (B,K,H/2^ds,W/2^ds) fp32."""
        ops.require_device(images)
        images = images.to(torch.float32).contiguous()
        if images.dim() != 4 or images.shape[1] != 3:
            raise ValueError(f"images must be (B, 3, H, W), got {tuple(images.shape)}")
        B, _, H, W = images.shape
        if H % 32 or W % 32:
            raise ValueError(f"image size must be a multiple of 32, got {H}x{W}")
        plan = self.plan
        wf, bf = self._folded()
        st = ops._stream()
        x4 = torch.empty(B, H, W, 4, device=self.device, dtype=torch.bfloat16)
        check(self._lib.lp_images_to_nhwc4(_p(images), B, H, W, _p(x4), st), "lp_images_to_nhwc4")
        # stem: the 7x7 kernel has no bias / ReLU store pass; its BatchNorm -> ReLU -> max-pool is already one fused pass
        sb = plan.stem_bn
        z, g = self._conv_fwd(plan.stem, x4, B, H, W, None)
        h, w = g.Ho, g.Wo
        mu, iv = self._bn_moments(sb, z, B * h * w, False, None)
        ph, pw = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        x = torch.empty(B, ph, pw, 64, device=self.device, dtype=torch.bfloat16)
        arg = torch.empty(B, ph, pw, 64, device=self.device, dtype=torch.uint8)
        check(self._lib.lp_bn_relu_maxpool_fwd(_p(z), _p(mu), _p(iv), _p(self.param_view(sb, "weight")), _p(self.param_view(sb, "bias")), B, h, w,
                                               64, _p(x), _p(arg), st), "lp_bn_relu_maxpool_fwd")
        del z, arg
        h, w = ph, pw

        def layer(c: ConvP, b: BNP, xin, hh, ww, residual, relu):
            g_ = self._geom(c, B, hh, ww)
            out = torch.empty(B, g_.Ho, g_.Wo, c.Co, device=self.device, dtype=torch.bfloat16)
            run = lambda: check(self._lib.lp_conv_fwd_act(_p(xin), _p(wf[c.w_off:]), C.byref(g_), _p(bf[b.b_off:]), _p(residual), int(relu),  # noqa: E731
                                                          _p(out), st), "lp_conv_fwd_act")
            self._timed(f"conv_igemm_kernel<{128 if c.Co > 64 else 64},infer>", self._flops(c, g_), run, self._bytes(c, g_), layer=c.name)
            return out, g_

        for blk in plan.blocks:
            a1, _ = layer(blk.conv1, blk.bn1, x, h, w, None, True)
            a2, g2 = layer(blk.conv2, blk.bn2, a1, h, w, None, True)
            idt = x if blk.down is None else layer(blk.down, blk.dbn, x, h, w, None, False)[0]
            x, _ = layer(blk.conv3, blk.bn3, a2, g2.Ho, g2.Wo, idt, True)
            h, w = g2.Ho, g2.Wo
        return self._head_forward(x, B, h, w, {})

    # ------------------------------------------------------------------------------------------------ backward
    def _bn_bwd(self, b: BNP, dy, y_out, z, mean, invstd, M: int, want_dres: bool, sums: torch.Tensor | None = None, seg: int = 0, ds=None):
        """``sums``: the (segments,2,C) fixed-point reductions [sum dy, sum dy*xhat] when the dgrad that produced ``dy`` already made them.
        The apply launch also adds THIS rank's sums into d beta / d gamma (they are the parameter gradients).
        ``ds`` = (zd, mean_d, invstd_d) of the block's projection-shortcut BatchNorm, which the same (masked) gradient feeds: the walk then
        also leaves THAT BatchNorm's two reductions - returned as a third value, or None where the library declines (the caller then runs
        lp_bn_bwd_reduce as before) - instead of a separate pass over the gradient (lp_bn_bwd_apply_seg_ds)."""
        B = z.shape[0]
        rpi = M // B
        segs = self._segments(B, seg)
        Cn = b.C
        if sums is None:
            sums = self._zeros_fx(len(segs) * 2 * Cn)
            rws = self._reduce_ws(M, Cn)
            for si, (i0, n) in enumerate(segs):
                check(self._lib.lp_bn_bwd_reduce(_p(dy[i0:i0 + n]), _p(y_out[i0:i0 + n]) if y_out is not None else None, _p(z[i0:i0 + n]),
                                                 _p(mean[si * Cn:]), _p(invstd[si * Cn:]), n * rpi, Cn, _p(sums[si * 4 * Cn:]), _p(rws), rws.numel(),
                                                 ops._stream()), "lp_bn_bwd_reduce")
        world = 1
        local, total = sums, sums
        if not self._bwd_training:
            # eval-mode BatchNorm (running statistics) is a fixed per-channel affine map: dz = dy * gamma * invstd, without the
            # batch-statistics correction terms (d gamma / d beta are the same sums in both modes)
            total = None
        elif self.sync_bn:
            local = sums.clone()
            self._sync_stats(sums)
            self.sync_bn_messages += 1
            world = dist.get_world_size(self.process_group)
        dz = torch.empty_like(z)
        dres = torch.empty_like(z) if want_dres else None
        gam = _p(self.param_view(b, "weight"))
        counts = [float(n * rpi * world) for _, n in segs]
        # sum / count as floats (written by the call's first launch)
        terms = torch.empty(len(segs) * 2 * Cn, device=z.device, dtype=torch.float32) if self.bn_bwd_terms else None
        if ds is not None:
            dsums = None
            if self.bn_bwd_ds and terms is not None and 256 % (Cn // 8) == 0:
                zd, md, vd = ds
                dsums = self._zeros_fx(len(segs) * 2 * Cn)
                need = int(self._lib.lp_bn_bwd_ds_workspace_bytes(M, Cn))
                if getattr(self, "_ds_ws", None) is None or self._ds_ws.numel() < need:
                    self._ds_ws = torch.empty(need, device=self.device, dtype=torch.uint8)
                check(self._lib.lp_bn_bwd_apply_seg_ds(_p(dy), _p(y_out), _p(z), _p(mean), _p(invstd), gam, _p(total), counts[0], counts[-1], M, Cn,
                                                       seg * rpi, _p(dz), _p(dres), _p(local), _p(self.G[b.b_off:]), _p(self.G[b.g_off:]),
                                                       _p(terms), _p(zd), _p(md), _p(vd), _p(dsums), _p(self._ds_ws), self._ds_ws.numel(),
                                                       ops._stream()), "lp_bn_bwd_apply_seg_ds")
                return dz, dres, dsums
        check(self._lib.lp_bn_bwd_apply_seg(_p(dy), _p(y_out), _p(z), _p(mean), _p(invstd), gam, _p(total), counts[0], counts[-1], M, Cn,
                                            seg * rpi, _p(dz), _p(dres), _p(local), _p(self.G[b.b_off:]), _p(self.G[b.g_off:]), _p(terms),
                                            ops._stream()), "lp_bn_bwd_apply_seg")
        return (dz, dres, None) if ds is not None else (dz, dres)

    def _conv_bwd(self, c: ConvP, x, dz, B, Hi, Wi, need_dx: bool, addend=None, relu_mask=None, accumulate_into=None, bn=None,
                  relu_bits=None, seg: int = 0, mask_bits=None, addend_half: bool = False):
        """wgrad into G, and (optionally) dx = dgrad(dz) + addend, zeroed where relu_mask <= 0 (fused ReLU backward).
        ``accumulate_into``: add the data gradient in place into an existing gradient tensor (which must already be masked);
        for stride-2 layers only the pixels a filter tap reaches are touched.
        ``bn`` = (BNP, z, mean, invstd, sums): dx is the gradient of relu(BN(z) [+ residual]); the launch also leaves BatchNorm's
        two backward reductions in ``sums``.  Without ``relu_mask`` the ReLU mask is recomputed from z (no residual branch)."""
        g = self._geom(c, B, Hi, Wi)
        # (the weight gradient is enqueued BEFORE the layer's data gradient: the side stream starts it at once, beside the data gradient.  Enqueued
        # behind it - so that it would run beside the HBM-bound BatchNorm kernels instead - the step was 1.3 % slower, profiles/r05o_wgrad_order_ab.txt)
        self._timed(f"conv_wgrad_kernel<{128 if c.Co > 64 else 64}>", self._flops(c, g),
                    lambda: self._wgrad(x, dz, g, self.G[c.w_off:]), self._bytes(c, g, wgrad=True), layer=c.name)
        if not need_dx:
            return None
        st = ops._stream()
        if accumulate_into is not None:
            dx, addend, skip = accumulate_into, accumulate_into, 1
        else:
            dx, skip = torch.empty(B, Hi, Wi, c.Ci, device=self.device, dtype=torch.bfloat16), 0
        if bn is not None:
            assert accumulate_into is None
            b, z, mean, invstd, sums = bn
            if relu_bits is not None:
                relu_mask = None  # the 1-bit form replaces the activation tensor as the mask source
            f = self._bn_fuse(sums, b, z, mean, invstd, mask_from_z=relu_mask is None and relu_bits is None, relu_bits=relu_bits, seg=seg,
                              addend_half=addend_half)
            run = lambda: check(self._lib.lp_conv_dgrad_bn(_p(dz), _p(self.Wd[c.wd_off:]), C.byref(g), _p(addend), _p(relu_mask), _p(dx),  # noqa: E731
                                                           C.byref(f), st), "lp_conv_dgrad_bn")
        elif mask_bits is not None and self.dgrad_mask_bits:
            # the ReLU mask of x at 1 bit per element (written by lp_bn_apply beside x): 1/16 of the bytes the activation itself would cost
            relu_mask = None
            run = lambda: check(self._lib.lp_conv_dgrad_bits(_p(dz), _p(self.Wd[c.wd_off:]), C.byref(g), _p(addend), _p(mask_bits), _p(dx),  # noqa: E731
                                                             skip, st), "lp_conv_dgrad_bits")
        else:
            run = lambda: check(self._lib.lp_conv_dgrad(_p(dz), _p(self.Wd[c.wd_off:]), C.byref(g), None, _p(addend), _p(relu_mask),  # noqa: E731
                                                        _p(dx), None, c.Ci, 0, skip, st), "lp_conv_dgrad")
        # algorithmic bytes of the launch: the contraction's operands / result plus what its fused store pass reads back (the gradient arriving
        # over the residual branch, the pre-normalisation tensor of the fused BatchNorm backward, the ReLU mask as activation or 1 bit each)
        dx_bytes = 2.0 * B * Hi * Wi * c.Ci
        extra = ((dx_bytes / 4 if addend_half else dx_bytes) if addend is not None else 0.0) + (dx_bytes if bn is not None else 0.0) + \
                (dx_bytes if relu_mask is not None else 0.0) + (dx_bytes / 16 if (bn is not None and relu_bits is not None) else 0.0) + \
                (dx_bytes / 16 if (bn is None and relu_mask is None and mask_bits is not None and self.dgrad_mask_bits) else 0.0)
        self._timed(f"conv_igemm_kernel<{128 if c.Ci > 64 else 64},dgrad>", self._flops(c, g), run, self._bytes(c, g) + extra, layer=c.name)
        return dx

    def backward(self, tp: Tape, g_heat: torch.Tensor, trace: dict | None = None) -> None:
        """Accumulate d loss / d parameters into self.G given d loss / d heat-maps (B,K,h,w) fp32.

        ``trace`` (tests only): receives the gradient tensor at every block boundary."""
        T, plan = tp.t, self.plan
        B, H, W = tp.meta["B"], tp.meta["H"], tp.meta["W"]
        self._bwd_training = bool(tp.meta.get("training", True))
        seg = tp.meta.get("seg", 0)
        nseg = 2 if seg else 1
        # One zeroed arena for every reduction target of this pass: the sums of each BatchNorm backward (fused into a data gradient or
        # not), the head's bias sums, the stem's.
        self._zero_arena = torch.zeros(4 * ((sum(2 * b.C for b in plan.bns) + 2 * plan.stem_bn.C) * nseg + sum(2 * c.Ci for c in plan.head)) + 64,
                                       device=self.device, dtype=torch.float32)
        self._zero_off = 0
        try:
            self._backward(tp, g_heat, trace, seg, nseg)
        finally:
            self._zero_arena = None

    def _backward(self, tp: Tape, g_heat: torch.Tensor, trace: dict | None, seg: int, nseg: int) -> None:
        T, plan = tp.t, self.plan
        B, H, W = tp.meta["B"], tp.meta["H"], tp.meta["W"]
        d = self._head_backward(T, B, g_heat)
        progress = self.grad_progress if (self.grad_progress is not None and self.single_backward) else None
        if progress is not None:
            progress(plan.n_backbone)  # the head's gradients (the tail of the flat buffer) are complete

        def new_sums(b: BNP) -> torch.Tensor:
            return self._zeros_fx(2 * b.C * nseg)

        d_sums = None  # reductions of the current block's bn3 backward, when the dgrad that produced `d` already made them
        for i in range(len(plan.blocks) - 1, -1, -1):
            blk, key = plan.blocks[i], f"b{i}"
            if trace is not None:
                trace[f"{key}.dout"] = d
            hi, wi = tp.meta[f"{key}.hw"]
            st = blk.conv2.stride
            ho, wo = (hi - 1) // st + 1, (wi - 1) // st + 1
            Mo, Mi = B * ho * wo, B * hi * wi
            x = T[f"{key}.x"]
            last = i == len(plan.blocks) - 1
            # ReLU backward AND the two reductions of the BatchNorm backward are fused into the dgrad that PRODUCES each
            # gradient: its store pass zeroes the gradient where the activation is <= 0 (mask recomputed from the saved
            # pre-normalisation tensor, or read from the block output when there is a residual branch) and leaves
            # [sum dy, sum dy * xhat] per channel.  Only the trunk output (fed by the head) still runs lp_bn_bwd_reduce (the inputs of
            # the stride-2 blocks, with their two writers, did until round 5: see the blk.down branch below); the stem has its own fused
            # pair (lp_bn_pool_bwd_*).
            dbn_sums = None   # the projection shortcut's BatchNorm reductions, taken by bn3's backward walk over the same gradient where it can
            if last:
                dz3, dres = self._bn_bwd(blk.bn3, d, T[f"{key}.out"], T[f"{key}.z3"], T[f"{key}.m3"], T[f"{key}.v3"], Mo, True, seg=seg)
            elif blk.down is not None:
                dz3, _, dbn_sums = self._bn_bwd(blk.bn3, d, None, T[f"{key}.z3"], T[f"{key}.m3"], T[f"{key}.v3"], Mo, False, sums=d_sums, seg=seg,
                                                ds=(T[f"{key}.zd"], T[f"{key}.md"], T[f"{key}.vd"]))
                dres = d
            else:
                dz3, _ = self._bn_bwd(blk.bn3, d, None, T[f"{key}.z3"], T[f"{key}.m3"], T[f"{key}.v3"], Mo, False, sums=d_sums, seg=seg)
                dres = d
            s2 = new_sums(blk.bn2)
            da2 = self._conv_bwd(blk.conv3, T[f"{key}.a2"], dz3, B, ho, wo, True,
                                 bn=(blk.bn2, T[f"{key}.z2"], T[f"{key}.m2"], T[f"{key}.v2"], s2), seg=seg)
            dz2, _ = self._bn_bwd(blk.bn2, da2, None, T[f"{key}.z2"], T[f"{key}.m2"], T[f"{key}.v2"], Mo, False, sums=s2, seg=seg)
            s1 = new_sums(blk.bn1)
            da1 = self._conv_bwd(blk.conv2, T[f"{key}.a1"], dz2, B, hi, wi, True,
                                 bn=(blk.bn1, T[f"{key}.z1"], T[f"{key}.m1"], T[f"{key}.v1"], s1), seg=seg)
            dz1, _ = self._bn_bwd(blk.bn1, da1, None, T[f"{key}.z1"], T[f"{key}.m1"], T[f"{key}.v1"], Mi, False, sums=s1, seg=seg)
            mask_x = x if i > 0 else None  # block 0's input is the max-pool output: its ReLU is handled by the stem BN backward
            d_sums = None
            if blk.down is not None:
                dzd, _ = self._bn_bwd(blk.dbn, dres, None, T[f"{key}.zd"], T[f"{key}.md"], T[f"{key}.vd"], Mo, False, sums=dbn_sums, seg=seg)
                xbits = T.get(f"b{i - 1}.out_bits") if i > 0 else None   # x = the previous block's output: its 1-bit ReLU mask exists
                if i > 0 and xbits is not None and self.dgrad_half_addend and blk.down.stride == 2 and blk.down.k == 1 and blk.conv1.stride == 1:
                    # The shortcut's data gradient FIRST, on its own (half-resolution) grid - a plain 1x1 product, no read-modify-write of the
                    # block input's gradient - then conv1's data gradient adds it at the even pixels as the LAST writer and so can take the
                    # ReLU mask and the previous block's BatchNorm reductions in its store pass like every identity block's does: the
                    # stand-alone reduction over the two-writer gradient (lp_bn_bwd_reduce, 2 passes over it) is gone.
                    prev, pk = plan.blocks[i - 1], f"b{i - 1}"
                    gd = self._geom(blk.down, B, hi, wi)
                    self._timed(f"conv_wgrad_kernel<{128 if blk.down.Co > 64 else 64}>", self._flops(blk.down, gd),
                                lambda: self._wgrad(x, dzd, gd, self.G[blk.down.w_off:]), self._bytes(blk.down, gd, wgrad=True), layer=blk.down.name)
                    gh = _lib.ConvGeom(B, gd.Ho, gd.Wo, blk.down.Ci, gd.Ho, gd.Wo, blk.down.Co, 1, 1, 1, 0)   # the shortcut on its own grid: stride 1
                    dd = torch.empty(B, gd.Ho, gd.Wo, blk.down.Ci, device=self.device, dtype=torch.bfloat16)
                    self._timed(f"conv_igemm_kernel<{128 if blk.down.Ci > 64 else 64},dgrad>", self._flops(blk.down, gd),
                                lambda: check(self._lib.lp_conv_dgrad(_p(dzd), _p(self.Wd[blk.down.wd_off:]), C.byref(gh), None, None, None, _p(dd), None,
                                                                      blk.down.Ci, 0, 0, ops._stream()), "lp_conv_dgrad"),
                                2.0 * B * gd.Ho * gd.Wo * (blk.down.Co + blk.down.Ci), layer=blk.down.name)
                    d_sums = new_sums(prev.bn3)
                    d = self._conv_bwd(blk.conv1, x, dz1, B, hi, wi, True, addend=dd, addend_half=True,
                                       bn=(prev.bn3, T[f"{pk}.z3"], T[f"{pk}.m3"], T[f"{pk}.v3"], d_sums), relu_bits=xbits, seg=seg)
                else:
                    # main path first, then the projection shortcut accumulates in place (no dense temporary; for the stride-2
                    # shortcuts only every 4th pixel is touched); (a + b) * mask == (a * mask + b) * mask for a 0/1 mask
                    d = self._conv_bwd(blk.conv1, x, dz1, B, hi, wi, True, relu_mask=mask_x, mask_bits=xbits)
                    self._conv_bwd(blk.down, x, dzd, B, hi, wi, True, relu_mask=mask_x, accumulate_into=d, mask_bits=xbits)
            elif i > 0:
                prev, pk = plan.blocks[i - 1], f"b{i - 1}"
                d_sums = new_sums(prev.bn3)
                d = self._conv_bwd(blk.conv1, x, dz1, B, hi, wi, True, addend=dres, relu_mask=mask_x,
                                   bn=(prev.bn3, T[f"{pk}.z3"], T[f"{pk}.m3"], T[f"{pk}.v3"], d_sums),
                                   relu_bits=T.get(f"{pk}.out_bits"), seg=seg)
            else:
                d = self._conv_bwd(blk.conv1, x, dz1, B, hi, wi, True, addend=dres, relu_mask=mask_x)
            if progress is not None:
                progress(blk.conv1.w_off)  # parameters are laid out in forward order: everything from this block on is final

        if trace is not None:
            trace["stem.dpool"] = d
        sh, sw = tp.meta["stem_hw"]
        # max-pool, ReLU and BatchNorm backward of the stem in two passes over z (the reductions, then dz): the activation's gradient
        # is rebuilt on the fly from the pooled gradient and the arg-max bytes, the ReLU gate from z
        sb = plan.stem_bn
        segs = self._segments(B, seg)
        ssum = self._zeros_fx(nseg * 2 * sb.C)
        gam, bet = self.param_view(sb, "weight"), self.param_view(sb, "bias")
        arg, sz, smu, siv = T["pool.arg"], T["stem.z"], T["stem.mu"], T["stem.iv"]
        for si, (i0, n) in enumerate(segs):
            check(self._lib.lp_bn_pool_bwd_reduce(_p(arg[i0:i0 + n]), _p(d[i0:i0 + n]), _p(sz[i0:i0 + n]), _p(smu[si * sb.C:]), _p(siv[si * sb.C:]),
                                                  _p(gam), _p(bet), n, sh, sw, 64, _p(ssum[si * 4 * sb.C:]), ops._stream()), "lp_bn_pool_bwd_reduce")
        world = 1
        slocal, stotal = ssum, ssum
        if not self._bwd_training:
            stotal = None
        elif self.sync_bn:
            slocal = ssum.clone()
            self._sync_stats(ssum)
            self.sync_bn_messages += 1
            world = dist.get_world_size(self.process_group)
        dz = torch.empty(B, sh, sw, 64, device=self.device, dtype=torch.bfloat16)
        for si, (i0, n) in enumerate(segs):
            check(self._lib.lp_bn_pool_bwd_apply(_p(arg[i0:i0 + n]), _p(d[i0:i0 + n]), _p(sz[i0:i0 + n]), _p(smu[si * sb.C:]), _p(siv[si * sb.C:]),
                                                 _p(gam), _p(bet), _p(stotal[si * 4 * sb.C:]) if stotal is not None else None,
                                                 float(n * sh * sw * world), n, sh, sw, 64, _p(dz[i0:i0 + n]), _p(slocal[si * 4 * sb.C:]),
                                                 _p(self.G[sb.b_off:]), _p(self.G[sb.g_off:]), ops._stream()), "lp_bn_pool_bwd_apply")
        g = self._geom(plan.stem, B, H, W)
        self._timed("conv_wgrad_kernel<64,stem>", self._flops(plan.stem, g),
                    lambda: self._wgrad(T["x4"], dz, g, self.G[plan.stem.w_off:], stem=True), self._bytes(plan.stem, g, wgrad=True), layer=plan.stem.name)
        self._join_side_stream()


class HeadEngine(Engine):
    """The head alone, for a ``HeatmapHead`` used outside a tracker (reference models/heads/heatmap.py:147-212, as its own tests use it:
    tests/models/heads/test_heatmap.py): PixelShuffle(2) + n ConvTranspose2d + optional spatial soft-max over its own flat buffers, with
    every constructor option of the reference class (``deconv_out_channels``, ``final_softmax``).  Same kernels and tape as the head
    of the full engines; features come and go as (B, C, h, w) fp32, the arithmetic is bf16-mixed like the trunk's."""

    def __init__(self, in_channels: int, stride: int, num_keypoints: int, downsample_factor: int = 2, deconv_out_channels: int | None = None,
                 final_softmax: bool = True, device: torch.device | str = "cuda:0"):
        if num_keypoints > CPAD:
            raise NotImplementedError(f"at most {CPAD} heat-map channels are supported, got {num_keypoints}")
        if in_channels % 4:
            raise ValueError(f"PixelShuffle(2) needs a multiple of 4 input channels, got {in_channels}")
        head = build_head(in_channels, stride, num_keypoints, downsample_factor, deconv_out_channels)
        plan = Plan(None, None, [], head)
        off = wd = 0
        for c in head:
            c.w_off, c.wd_off = off, wd
            off += c.numel
            wd += c.numel
            c.bias_off = off
            off += c.Ci
            plan.convs.append(c)
        plan.n_backbone, plan.n_total, plan.n_wd, plan.n_running = 0, off, wd, 0
        super().__init__(num_keypoints, downsample_factor, device, plan=plan)
        self.in_channels = in_channels
        self.final_softmax = bool(final_softmax)

    def state_dict(self) -> dict[str, torch.Tensor]:
        sd: dict[str, torch.Tensor] = {}
        for c in self.plan.head:
            sd[f"{c.name}.weight"] = self.param_view(c)
            sd[f"{c.name}.bias"] = self.param_view(c, "bias")
        return sd

    def forward(self, features: torch.Tensor, training: bool = True) -> tuple[torch.Tensor, Tape]:   # type: ignore[override]
        ops.require_device(features)
        if features.dim() != 4 or features.shape[1] != self.in_channels:
            raise ValueError(f"features must be (B, {self.in_channels}, h, w), got {tuple(features.shape)}")
        B, _, h, w = features.shape
        x = features.detach().permute(0, 2, 3, 1).to(torch.bfloat16).contiguous()
        tp = Tape()
        tp.meta.update(B=B, h=h, w=w)
        heat = self._head_forward(x, B, h, w, tp.t)
        return heat, tp

    def backward(self, tp: Tape, g_heat: torch.Tensor, trace: dict | None = None) -> torch.Tensor:   # type: ignore[override]
        """Parameter gradients accumulate into G; returns d loss / d features (B, C, h, w) fp32."""
        d = self._head_backward(tp.t, tp.meta["B"], g_heat)
        self._join_side_stream()
        return d.permute(0, 3, 1, 2).to(torch.float32)
