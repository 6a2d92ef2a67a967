"""fp32 VALIDATION executor of the ViT-S/16 heatmap tracker (config C4) - the ViT counterpart of engine_fp32.Fp32Engine.

The reference trains in fp32 only (lightning_pose/train.py:411-428); its ViT backbone is ``VisionEncoder`` over HuggingFace ``ViTModel``
(lightning_pose/models/backbones/vit.py:16-49).  ``ViTEngine`` (bf16-mixed) is the product and the measured path; this subclass runs the same
plan with every tensor and every contraction in fp32 so that a whole training step can be held to BASELINE.json's 1e-4.
"""

from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib, ops
from ._lib import check
from .engine import Tape
from .engine_fp32 import Fp32Engine
from .ops import _p
from .vit_engine import LN_EPS, Lin, LNP, ViTEngine


class Fp32ViTEngine(ViTEngine):
    """config C4 (ViT-S/16, reference models/backbones/vit.py:16-49 over HuggingFace ``ViTModel``) in the reference's own precision:
    the plan, flat buffers, state_dict names and optimiser of :class:`ViTEngine`, every tensor and contraction fp32
    (csrc/vit_f32.hip for LayerNorm / GELU / tokens / attention, lp_f32_conv_* with a 1x1 geometry for the Linear layers, the head of
    :class:`Fp32Engine`).  VALIDATION path: untuned, selected with ``precision="fp32"`` / ``LP_PRECISION=fp32``."""

    precision = "fp32"
    _f32 = Fp32Engine._f32
    _wdims = staticmethod(Fp32Engine._wdims)
    _wg = Fp32Engine._wg
    _head_forward = Fp32Engine._head_forward
    _head_backward = Fp32Engine._head_backward

    @staticmethod
    def _lin_geom(l: Lin, M: int):
        return _lib.ConvGeom(1, 1, M, l.K, 1, M, l.N, 1, 1, 1, 0)

    def _linear(self, x: torch.Tensor, l: Lin, M: int) -> torch.Tensor:
        out = self._f32(M, l.N)
        check(self._lib.lp_f32_conv_fwd(_p(x), _p(self.P[l.w_off:]), C.byref(self._lin_geom(l, M)), 1, 1, l.K, _p(self.P[l.b_off:]), None,
                                        _p(out), ops._stream()), "lp_f32_conv_fwd(linear)")
        return out

    def _linear_bwd(self, l: Lin, x: torch.Tensor, dy: torch.Tensor, M: int, need_dx: bool = True):
        g = self._lin_geom(l, M)
        bsum = torch.zeros(2 * l.N, device=self.device, dtype=torch.float32)
        check(self._lib.lp_f32_bn_stats(_p(dy), M, l.N, _p(bsum), ops._stream()), "lp_f32_bn_stats(bias)")
        self.G[l.b_off:l.b_off + l.N] += bsum[:l.N]
        check(self._lib.lp_f32_conv_wgrad(_p(x), _p(dy), C.byref(g), 1, 1, l.K, _p(self.G[l.w_off:]), ops._stream()), "lp_f32_conv_wgrad(linear)")
        if not need_dx:
            return None
        dx = self._f32(M, l.K)
        check(self._lib.lp_f32_conv_dgrad(_p(dy), _p(self.P[l.w_off:]), C.byref(g), 1, 1, l.K, None, None, _p(dx), ops._stream()),
              "lp_f32_conv_dgrad(linear)")
        return dx

    def _ln(self, x, delta, l: LNP, M: int, drop_T: int = 0):
        D = self.plan.D
        xo = torch.empty_like(x) if delta is not None else None
        rows = M - M // drop_T if drop_T else M
        y = self._f32(rows, D)
        mean, rstd = self._f32(M), self._f32(M)
        check(self._lib.lp_f32_layernorm_fwd(_p(x), _p(delta), _p(xo), _p(self.P[l.g_off:]), _p(self.P[l.b_off:]), LN_EPS, M, D, drop_T, _p(y),
                                             _p(mean), _p(rstd), ops._stream()), "lp_f32_layernorm_fwd")
        return y, mean, rstd, (xo if xo is not None else x)

    def _ln_bwd(self, dy, x, mean, rstd, l: LNP, M: int, dx, drop_T: int = 0, want_bf16: bool = False):
        check(self._lib.lp_f32_layernorm_bwd(_p(dy), _p(x), _p(mean), _p(rstd), _p(self.P[l.g_off:]), M, self.plan.D, drop_T, _p(dx),
                                             _p(self.G[l.g_off:]), _p(self.G[l.b_off:]), ops._stream()), "lp_f32_layernorm_bwd")
        return dx.clone() if want_bf16 else None   # (the operand of the next Linear backward: dx itself moves on in place)

    def _forward(self, images, training: bool, keep: bool):
        parts = list(images) if isinstance(images, (list, tuple)) else [images]
        for p_ in parts:
            ops.require_device(p_)
        parts = [p_.to(torch.float32).contiguous() for p_ in parts]
        if any(p_.shape[1:] != parts[0].shape[1:] for p_ in parts):
            raise ValueError("a joint pass takes batches of equally sized images")
        if parts[0].dim() != 4 or parts[0].shape[1] != 3:
            raise ValueError(f"images must be (B, 3, H, W), got {tuple(parts[0].shape)}")
        _, _, H, W = parts[0].shape
        B = sum(p_.shape[0] for p_ in parts)
        pl = self.plan
        D, nh, pt = pl.D, pl.heads, pl.patch
        if H % pt or W % pt:
            raise ValueError(f"image size must be a multiple of the patch size {pt}, got {H}x{W}")
        gh, gw = H // pt, W // pt
        Np, Tn = gh * gw, gh * gw + 1
        M = B * Tn
        tp = Tape()
        T = tp.t
        patches = self._f32(B * Np, 3 * pt * pt)
        i0 = 0
        for p_ in parts:
            check(self._lib.lp_f32_vit_patchify(_p(p_), p_.shape[0], H, W, pt, _p(patches[i0 * Np:]), ops._stream()), "lp_f32_vit_patchify")
            i0 += p_.shape[0]
        pe = self._linear(patches, pl.patch_lin, B * Np)
        x = self._f32(M, D)
        pos = self._pos(gh, gw)
        check(self._lib.lp_f32_vit_tokens_fwd(_p(pe), _p(self.P[pl.cls_off:]), _p(pos), B, Np, D, _p(x), ops._stream()), "lp_f32_vit_tokens_fwd")
        T["patches"] = patches
        delta = None
        scale = 1.0 / math.sqrt(D // nh)
        qs = 3 * D
        for i, L in enumerate(pl.layers):
            y1, m1, r1, x = self._ln(x, delta, L["ln1"], M)
            qkv = self._linear(y1, L["qkv"], M)
            Pm = self._f32(B * nh * Tn, Tn)
            attn = self._f32(M, D)
            check(self._lib.lp_f32_attn_fwd(_p(qkv), qs, D, 2 * D, B, nh, Tn, scale, _p(Pm), _p(attn), D, ops._stream()), "lp_f32_attn_fwd")
            proj = self._linear(attn, L["proj"], M)
            x_in = x
            y2, m2, r2, x = self._ln(x, proj, L["ln2"], M)
            h1 = self._linear(y2, L["fc1"], M)
            a1 = torch.empty_like(h1)
            check(self._lib.lp_f32_gelu_fwd(_p(h1), h1.numel(), _p(a1), ops._stream()), "lp_f32_gelu_fwd")
            delta = self._linear(a1, L["fc2"], M)
            if keep:
                for nm, v in (("x_in", x_in), ("m1", m1), ("r1", r1), ("y1", y1), ("qkv", qkv), ("P", Pm), ("attn", attn), ("x_mid", x),
                              ("m2", m2), ("r2", r2), ("y2", y2), ("h1", h1), ("a1", a1)):
                    T[f"l{i}.{nm}"] = v
        feat, mf, rf, x = self._ln(x, delta, pl.lnf, M, drop_T=Tn)
        T["x_last"], T["mf"], T["rf"] = x, mf, rf
        heat = self._head_forward(feat.view(B, gh, gw, D), B, gh, gw, T)
        tp.meta.update(B=B, H=H, W=W, gh=gh, gw=gw, training=training)
        return heat, tp

    def backward(self, tp: Tape, g_heat: torch.Tensor, trace: dict | None = None) -> None:
        T, pl = tp.t, self.plan
        B, gh, gw = tp.meta["B"], tp.meta["gh"], tp.meta["gw"]
        D, nh = pl.D, pl.heads
        Np, Tn = gh * gw, gh * gw + 1
        M = B * Tn
        scale = 1.0 / math.sqrt(D // nh)
        qs = 3 * D
        d_feat = self._head_backward(T, B, g_heat).contiguous()            # (B, gh, gw, D) fp32
        dx = torch.zeros(M, D, device=self.device, dtype=torch.float32)   # gradient of the residual stream
        dcur = self._ln_bwd(d_feat, T["x_last"], T["mf"], T["rf"], pl.lnf, M, dx, drop_T=Tn, want_bf16=True)
        for i in range(pl.depth - 1, -1, -1):
            L = pl.layers[i]
            t = lambda nm: T[f"l{i}.{nm}"]  # noqa: E731
            if trace is not None:
                trace[f"l{i}.dout"] = dx.clone()
            d_a1 = self._linear_bwd(L["fc2"], t("a1"), dcur, M)
            d_h1 = torch.empty_like(d_a1)
            check(self._lib.lp_f32_gelu_bwd(_p(t("h1")), _p(d_a1), d_a1.numel(), _p(d_h1), ops._stream()), "lp_f32_gelu_bwd")
            d_y2 = self._linear_bwd(L["fc1"], t("y2"), d_h1, M)
            dcur = self._ln_bwd(d_y2, t("x_mid"), t("m2"), t("r2"), L["ln2"], M, dx, want_bf16=True)
            d_attn = self._linear_bwd(L["proj"], t("attn"), dcur, M)
            dqkv = self._f32(M, qs)
            dS = self._f32(B * nh * Tn, Tn)
            check(self._lib.lp_f32_attn_bwd(_p(t("qkv")), qs, D, 2 * D, _p(d_attn), D, _p(t("P")), B, nh, Tn, scale, _p(dS), _p(dqkv), qs,
                                            ops._stream()), "lp_f32_attn_bwd")
            if trace is not None:
                trace[f"l{i}.dqkv"] = dqkv
            d_y1 = self._linear_bwd(L["qkv"], t("y1"), dqkv, M)
            dcur = self._ln_bwd(d_y1, t("x_in"), t("m1"), t("r1"), L["ln1"], M, dx, want_bf16=i > 0)
        if trace is not None:
            trace["tokens.dx"] = dx
        dpatch = self._f32(B * Np, D)
        dpos = self._f32(Tn, D)
        check(self._lib.lp_f32_vit_tokens_bwd(_p(dx), B, Np, D, _p(dpatch), _p(dpos), ops._stream()), "lp_f32_vit_tokens_bwd")
        self.G[pl.cls_off:pl.cls_off + D] += dpos[0]
        gpos = self.G[pl.pos_off:pl.pos_off + pl.n_pos * D].view(pl.n_pos, D)
        gpos[0] += dpos[0]
        if gh == self.grid0 and gw == self.grid0:
            gpos[1:] += dpos[1:]
        else:
            check(self._lib.lp_small_matmul(_p(self._interp[(gh, gw)]), _p(dpos[1:]), Np, pl.n_pos - 1, D, 1, 1, _p(gpos[1:]), ops._stream()),
                  "lp_small_matmul(adjoint)")
        self._linear_bwd(pl.patch_lin, T["patches"], dpatch, B * Np, need_dx=False)
