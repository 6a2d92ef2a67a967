"""fp32 VALIDATION executor of the heatmap tracker's network (ResNet-50 trunk + head) on the lp_f32_* kernels (csrc/fp32.hip).

The reference trains in fp32 only (lightning_pose/train.py:411-428 passes no ``precision=``) and BASELINE.json's north_star asks for
outputs "within 1e-4 fp32 / 1e-2 bf16" of it.  The bf16-mixed ``Engine`` is the product and the measured path; this subclass runs the
SAME plan - same flat fp32 parameter / gradient / running-statistics buffers, same NHWC activations and [Co][R][S][Ci] weights, same
state_dict names, same optimiser - with every activation and every contraction in fp32 (v_mfma_f32_32x32x2_f32), nothing fused and
nothing tuned, so that a whole training step (heat-maps, keypoints, every logged scalar, every parameter gradient) can be held against
the verbatim reference at 1e-4.  Select it with ``HeatmapTracker(..., precision="fp32")`` or ``LP_PRECISION=fp32``.

Reference behaviour reproduced (paths relative to the reference tree): torchvision ResNet-50 children[:-2]
(models/backbones/factory.py:322-348), HeatmapHead (models/heads/heatmap.py:20-83,147-212), training-mode BatchNorm per forward call
with running-statistics updates (two segments when the labeled and the unlabeled batch share a pass, models/base.py:682-695).
"""

from __future__ import annotations

import ctypes as C

import torch
import torch.distributed as dist

from . import ops
from ._lib import check
from .engine import BN_EPS, BN_MOMENTUM, BNP, CPAD, ConvP, Engine, Tape
from .ops import _p


class Fp32Engine(Engine):
    wgrad_side_stream = False
    precision = "fp32"

    # ------------------------------------------------------------------------------------------------ kernels
    @staticmethod
    def _wdims(c: ConvP) -> tuple[int, int, int]:
        """storage dims (KH, KW, CiS) of the layer's weight in the flat buffer"""
        return (8, 8, 4) if c.kind == "stem" else (c.k, c.k, c.Ci)

    def can_segment(self, n0: int, H: int, W: int) -> bool:
        """nothing is fused into tiles here: two BatchNorm segments are simply two calls of the BatchNorm kernels"""
        return n0 > 0

    def _f32(self, *shape) -> torch.Tensor:
        return torch.empty(*shape, device=self.device, dtype=torch.float32)

    def _conv(self, c: ConvP, x: torch.Tensor, B: int, Hi: int, Wi: int):
        g = self._geom(c, B, Hi, Wi)
        out = self._f32(B, g.Ho, g.Wo, c.Co)
        check(self._lib.lp_f32_conv_fwd(_p(x), _p(self.P[c.w_off:]), C.byref(g), *self._wdims(c), None, None, _p(out), ops._stream()),
              "lp_f32_conv_fwd")
        return out, g

    def _dgrad(self, c: ConvP, dy: torch.Tensor, g, addend: torch.Tensor | None = None) -> torch.Tensor:
        dx = self._f32(g.B, g.Hi, g.Wi, g.Ci)
        check(self._lib.lp_f32_conv_dgrad(_p(dy), _p(self.P[c.w_off:]), C.byref(g), *self._wdims(c), None, _p(addend), _p(dx), ops._stream()),
              "lp_f32_conv_dgrad")
        return dx

    def _wg(self, c: ConvP, x: torch.Tensor, dy: torch.Tensor, g) -> None:
        check(self._lib.lp_f32_conv_wgrad(_p(x), _p(dy), C.byref(g), *self._wdims(c), _p(self.G[c.w_off:]), ops._stream()), "lp_f32_conv_wgrad")

    def _bn(self, b: BNP, z: torch.Tensor, residual, relu: bool, training: bool, seg: int):
        """-> (y, mean, invstd); mean / invstd are (segments, C) flattened"""
        B = z.shape[0]
        M = z.numel() // b.C
        rpi = M // B
        segs = self._segments(B, seg if training else 0)
        mean = self._f32(len(segs) * b.C)
        invstd = torch.empty_like(mean)
        if training:
            sums = torch.zeros(len(segs) * 2 * b.C, device=self.device, dtype=torch.float32)
            for si, (i0, n) in enumerate(segs):   # ordered partial sums (no atomics): the validation forward repeats bit for bit
                need = int(self._lib.lp_f32_bn_stats_workspace_bytes(n * rpi, b.C))
                if getattr(self, "_stats_ws", None) is None or self._stats_ws.numel() < need:
                    self._stats_ws = torch.empty(need, device=self.device, dtype=torch.uint8)
                check(self._lib.lp_f32_bn_stats_ordered(_p(z[i0:i0 + n]), n * rpi, b.C, _p(sums[si * 2 * b.C:]), _p(self._stats_ws),
                                                        self._stats_ws.numel(), ops._stream()), "lp_f32_bn_stats_ordered")
            counts = [float(n * rpi) for _, n in segs]
            if self.sync_bn:
                self._sync_stats(sums)
                self.sync_bn_messages += 1
                counts = [c_ * dist.get_world_size(self.process_group) for c_ in counts]
            rm, rv = _p(self.running_view(b, "running_mean")), _p(self.running_view(b, "running_var"))
            for si in range(len(segs)):   # segment 0's running-statistics update, then segment 1's (the reference's two forward calls)
                check(self._lib.lp_bn_finalize_f32(_p(sums[si * 2 * b.C:]), counts[si], b.C, BN_EPS, BN_MOMENTUM, _p(mean[si * b.C:]),
                                                   _p(invstd[si * b.C:]), rm, rv, ops._stream()), "lp_bn_finalize_f32")
        else:
            mean.copy_(self.running_view(b, "running_mean"))
            invstd.copy_((self.running_view(b, "running_var") + BN_EPS).rsqrt())
        y = torch.empty_like(z)
        gam, bet = _p(self.param_view(b, "weight")), _p(self.param_view(b, "bias"))
        for si, (i0, n) in enumerate(segs):
            check(self._lib.lp_f32_bn_apply(_p(z[i0:i0 + n]), _p(mean[si * b.C:]), _p(invstd[si * b.C:]), gam, bet,
                                            _p(residual[i0:i0 + n]) if residual is not None else None, int(relu), n * rpi, b.C,
                                            _p(y[i0:i0 + n]), ops._stream()), "lp_f32_bn_apply")
        return y, mean, invstd

    def _bn_back(self, b: BNP, dy, y_out, z, mean, invstd, want_dres: bool, seg: int):
        """gradient of relu?(BN(z) [+ residual]) w.r.t. z (and, with want_dres, the ReLU-masked gradient for the residual branch)"""
        B = z.shape[0]
        M = z.numel() // b.C
        rpi = M // B
        segs = self._segments(B, seg)
        Cn = b.C
        sums = torch.zeros(len(segs) * 2 * Cn, device=self.device, dtype=torch.float32)
        for si, (i0, n) in enumerate(segs):
            check(self._lib.lp_f32_bn_bwd_reduce(_p(dy[i0:i0 + n]), _p(y_out[i0:i0 + n]) if y_out is not None else None, _p(z[i0:i0 + n]),
                                                 _p(mean[si * Cn:]), _p(invstd[si * Cn:]), n * rpi, Cn, _p(sums[si * 2 * Cn:]),
                                                 _p(self.G[b.b_off:]), _p(self.G[b.g_off:]), ops._stream()), "lp_f32_bn_bwd_reduce")
        world = 1
        if not self._bwd_training:
            sums = torch.zeros_like(sums)  # eval-mode BatchNorm: a fixed affine map, no batch-statistics terms
        elif self.sync_bn:
            self._sync_stats(sums)
            self.sync_bn_messages += 1
            world = dist.get_world_size(self.process_group)
        dz = torch.empty_like(z)
        dres = torch.empty_like(z) if want_dres else None
        gam = _p(self.param_view(b, "weight"))
        for si, (i0, n) in enumerate(segs):
            check(self._lib.lp_f32_bn_bwd_apply(_p(dy[i0:i0 + n]), _p(y_out[i0:i0 + n]) if y_out is not None else None, _p(z[i0:i0 + n]),
                                                _p(mean[si * Cn:]), _p(invstd[si * Cn:]), gam, _p(sums[si * 2 * Cn:]), float(n * rpi * world),
                                                n * rpi, Cn, _p(dz[i0:i0 + n]), _p(dres[i0:i0 + n]) if want_dres else None, ops._stream()),
                  "lp_f32_bn_bwd_apply")
        return dz, dres

    # ------------------------------------------------------------------------------------------------ head
    def _head_forward(self, x: torch.Tensor, B: int, h: int, w: int, T: dict) -> torch.Tensor:
        head = self.plan.head
        cs, ld = head[0].cin, head[0].Co
        ps = (torch.zeros if ld != cs else torch.empty)(B, 2 * h, 2 * w, ld, device=self.device, dtype=torch.float32)
        check(self._lib.lp_f32_pixel_shuffle(_p(x), B, h, w, cs, ld, 0, _p(ps), ops._stream()), "lp_f32_pixel_shuffle")
        h, w = 2 * h, 2 * w
        T["head.in0"] = ps
        cur = ps
        for li, c in enumerate(head):
            g = self._geom(c, B, h, w)
            nxt = self._f32(B, 2 * h, 2 * w, CPAD)
            # ConvTranspose2d(k3, s2, p1, op1) forward = the data gradient of the mirrored convolution (+ bias)
            check(self._lib.lp_f32_conv_dgrad(_p(cur), _p(self.P[c.w_off:]), C.byref(g), *self._wdims(c), _p(self.P[c.bias_off:]), None, _p(nxt),
                                              ops._stream()), "lp_f32_conv_dgrad(head)")
            cur = nxt
            if li < len(head) - 1:
                T[f"head.in{li + 1}"] = cur
            h, w = 2 * h, 2 * w
        n = h * w
        heat = self._f32(B, self.K, h, w)
        check(self._lib.lp_softmax2d_fwd(_p(cur), n * CPAD, CPAD, 1, B, self.K, n, _p(heat), ops._stream()), "lp_softmax2d_fwd")
        T["heat"] = heat
        return heat

    def _head_backward(self, T: dict, B: int, g_heat: torch.Tensor) -> torch.Tensor:
        head = self.plan.head
        heat = T["heat"]
        _, K, h, w = heat.shape
        n = h * w
        g_heat = g_heat.to(torch.float32).contiguous()
        dcur = torch.zeros(B, h, w, CPAD, device=self.device, dtype=torch.float32)
        check(self._lib.lp_f32_softmax2d_bwd(_p(heat), _p(g_heat), B, K, n, _p(dcur), n * CPAD, CPAD, 1, ops._stream()), "lp_f32_softmax2d_bwd")
        for li in range(len(head) - 1, -1, -1):
            c = head[li]
            hs, ws = h // 2, w // 2
            g = self._geom(c, B, hs, ws)
            x_small = T[f"head.in{li}"]
            bsum = torch.zeros(2 * CPAD, device=self.device, dtype=torch.float32)
            check(self._lib.lp_f32_bn_stats(_p(dcur), B * h * w, CPAD, _p(bsum), ops._stream()), "lp_f32_bn_stats(bias)")
            self.G[c.bias_off:c.bias_off + CPAD] += bsum[:CPAD]
            self._wg(c, dcur, x_small, g)                      # the mirrored convolution: input = the ConvT output gradient
            dx = self._f32(B, hs, ws, c.Co)
            check(self._lib.lp_f32_conv_fwd(_p(dcur), _p(self.P[c.w_off:]), C.byref(g), *self._wdims(c), None, None, _p(dx), ops._stream()),
                  "lp_f32_conv_fwd(head bwd)")
            dcur, h, w = dx, hs, ws
        fh, fw = h // 2, w // 2
        cs, ld = head[0].cin, head[0].Co
        d = self._f32(B, fh, fw, 4 * cs)
        check(self._lib.lp_f32_pixel_shuffle(_p(dcur), B, fh, fw, cs, ld, 1, _p(d), ops._stream()), "lp_f32_pixel_shuffle(inv)")
        return d

    # ------------------------------------------------------------------------------------------------ forward
    def forward(self, images, training: bool = True) -> tuple[torch.Tensor, Tape]:
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
        tp = Tape()
        T, plan = tp.t, self.plan
        x4 = self._f32(B, H, W, 4)
        i0 = 0
        for p_ in parts:
            check(self._lib.lp_f32_images_to_nhwc4(_p(p_), p_.shape[0], H, W, _p(x4[i0:]), ops._stream()), "lp_f32_images_to_nhwc4")
            i0 += p_.shape[0]
        T["x4"] = x4
        z, g = self._conv(plan.stem, x4, B, H, W)
        a, mu, iv = self._bn(plan.stem_bn, z, None, True, training, seg)
        T["stem.z"], T["stem.a"], T["stem.mu"], T["stem.iv"] = z, a, mu, iv
        h, w = g.Ho, g.Wo
        tp.meta["stem_hw"] = (h, w)
        ph, pw = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        x = self._f32(B, ph, pw, 64)
        T["pool.arg"] = torch.empty(B, ph, pw, 64, device=self.device, dtype=torch.uint8)
        check(self._lib.lp_f32_maxpool_fwd(_p(a), B, h, w, 64, _p(x), _p(T["pool.arg"]), ops._stream()), "lp_f32_maxpool_fwd")
        h, w = ph, pw
        for i, blk in enumerate(plan.blocks):
            key = f"b{i}"
            T[f"{key}.x"] = x
            tp.meta[f"{key}.hw"] = (h, w)
            z1, _ = self._conv(blk.conv1, x, B, h, w)
            a1, m1, v1 = self._bn(blk.bn1, z1, None, True, training, seg)
            z2, g2 = self._conv(blk.conv2, a1, B, h, w)
            a2, m2, v2 = self._bn(blk.bn2, z2, None, True, training, seg)
            ho, wo = g2.Ho, g2.Wo
            if blk.down is not None:
                zd, _ = self._conv(blk.down, x, B, h, w)
                idt, md, vd = self._bn(blk.dbn, zd, None, False, training, seg)
                T[f"{key}.zd"], T[f"{key}.md"], T[f"{key}.vd"] = zd, md, vd
            else:
                idt = x
            z3, _ = self._conv(blk.conv3, a2, B, ho, wo)
            out, m3, v3 = self._bn(blk.bn3, z3, idt, True, training, seg)
            for nm, val in (("z1", z1), ("a1", a1), ("m1", m1), ("v1", v1), ("z2", z2), ("a2", a2), ("m2", m2), ("v2", v2),
                            ("z3", z3), ("m3", m3), ("v3", v3), ("out", out)):
                T[f"{key}.{nm}"] = val
            x, h, w = out, ho, wo
        heat = self._head_forward(x, B, h, w, T)
        tp.meta.update(B=B, H=H, W=W, training=training, seg=seg)
        if training:
            self.nbt += 2 if seg else 1
            self._fold = None
        return heat, tp

    def forward_infer(self, images: torch.Tensor) -> torch.Tensor:
        """eval-mode forward (running statistics), nothing folded: the same kernels as the training pass"""
        return self.forward(images, training=False)[0]

    # ------------------------------------------------------------------------------------------------ backward
    def backward(self, tp: Tape, g_heat: torch.Tensor, trace: dict | None = None) -> None:
        T, plan = tp.t, self.plan
        B, H, W = tp.meta["B"], tp.meta["H"], tp.meta["W"]
        seg = tp.meta.get("seg", 0)
        self._bwd_training = bool(tp.meta.get("training", True))
        d = self._head_backward(T, B, g_heat)
        for i in range(len(plan.blocks) - 1, -1, -1):
            blk, key = plan.blocks[i], f"b{i}"
            if trace is not None:
                trace[f"{key}.dout"] = d
            hi, wi = tp.meta[f"{key}.hw"]
            st = blk.conv2.stride
            ho, wo = (hi - 1) // st + 1, (wi - 1) // st + 1
            x = T[f"{key}.x"]
            dz3, dres = self._bn_back(blk.bn3, d, T[f"{key}.out"], T[f"{key}.z3"], T[f"{key}.m3"], T[f"{key}.v3"], True, seg)
            g3 = self._geom(blk.conv3, B, ho, wo)
            self._wg(blk.conv3, T[f"{key}.a2"], dz3, g3)
            da2 = self._dgrad(blk.conv3, dz3, g3)
            dz2, _ = self._bn_back(blk.bn2, da2, T[f"{key}.a2"], T[f"{key}.z2"], T[f"{key}.m2"], T[f"{key}.v2"], False, seg)
            g2 = self._geom(blk.conv2, B, hi, wi)
            self._wg(blk.conv2, T[f"{key}.a1"], dz2, g2)
            da1 = self._dgrad(blk.conv2, dz2, g2)
            dz1, _ = self._bn_back(blk.bn1, da1, T[f"{key}.a1"], T[f"{key}.z1"], T[f"{key}.m1"], T[f"{key}.v1"], False, seg)
            g1 = self._geom(blk.conv1, B, hi, wi)
            self._wg(blk.conv1, x, dz1, g1)
            if blk.down is not None:
                dzd, _ = self._bn_back(blk.dbn, dres, None, T[f"{key}.zd"], T[f"{key}.md"], T[f"{key}.vd"], False, seg)
                gd = self._geom(blk.down, B, hi, wi)
                self._wg(blk.down, x, dzd, gd)
                d = self._dgrad(blk.conv1, dz1, g1)
                d = self._dgrad(blk.down, dzd, gd, addend=d)
            else:
                d = self._dgrad(blk.conv1, dz1, g1, addend=dres)
        if trace is not None:
            trace["stem.dpool"] = d
        sh, sw = tp.meta["stem_hw"]
        da = self._f32(B, sh, sw, 64)
        check(self._lib.lp_f32_maxpool_bwd(_p(T["pool.arg"]), _p(d), B, sh, sw, 64, _p(da), ops._stream()), "lp_f32_maxpool_bwd")
        dz, _ = self._bn_back(plan.stem_bn, da, T["stem.a"], T["stem.z"], T["stem.mu"], T["stem.iv"], False, seg)
        self._wg(plan.stem, T["x4"], dz, self._geom(plan.stem, B, H, W))

