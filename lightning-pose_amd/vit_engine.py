"""ViT-S/16 ("vits_dino") backbone + heat-map head on the MI355X kernels: the drop-in for
``VisionEncoder`` -> HuggingFace ``ViTModel`` (reference models/backbones/vit.py:16-49, factory.py:188-190).

Same contract as :class:`lightning_pose_amd.engine.Engine` (flat fp32 parameter / gradient buffers, bf16 operand copies,
``forward`` -> (heat-maps, tape), hand-written ``backward``), parameters exposed under the reference's ``state_dict`` names
(``backbone.vision_encoder.embeddings.*``, ``...layers.{i}.attention.{q,k,v,o}_proj.*``, ``...layers.{i}.mlp.fc{1,2}.*``,
``...layernorm*``, ``head.upsampling_layers.*`` - the names of the installed transformers 5.x ``ViTModel``; the 4.x names
``encoder.layer.{i}.attention.attention.query`` ... are accepted by ``load_state_dict``).

Every Linear layer runs on the MFMA GEMM (``lp_gemm_nt``: forward and data-gradient; ``lp_conv_wgrad_bias``: weight + bias gradient).
Attention is ``lp_attn_fwd`` - one fused kernel per layer working straight on the (image, head) slices of the fused QKV tensor: the
scores never leave the chip, the probabilities are written once in bf16 (row pitch padded to a multiple of 64) for the backward
pass - and, backward, ``lp_attn_rowdot`` + ``lp_attn_dscores`` (soft-max backward in the store pass of dO V^T), ``lp_gemm_tn`` (dV, dK:
both operands contracted over their rows, no transposed copies) and one more ``lp_gemm_nt`` (dQ).  The glue (LayerNorm, GELU, token
assembly, the small head-slice transpose) is csrc/vit.hip.  The residual stream, LayerNorm statistics and all reductions are fp32;
GEMM operands bf16.
"""

from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib, ops
from ._lib import check
from .engine import ConvP, Engine, Tape, build_head
from .ops import _p

LN_EPS = 1e-12  # ViTConfig.layer_norm_eps


def bicubic_matrix(n_in: int, n_out: int) -> np.ndarray:
    """(n_out, n_in) weights of torch's 1-D bicubic resize, align_corners=False (A = -0.75, border taps clamped)."""
    A = -0.75
    w = np.zeros((n_out, n_in), np.float64)
    scale = n_in / n_out

    def k1(t):
        return ((A + 2) * t - (A + 3)) * t * t + 1

    def k2(t):
        return ((A * t - 5 * A) * t + 8 * A) * t - 4 * A

    for i in range(n_out):
        src = (i + 0.5) * scale - 0.5
        i0 = math.floor(src)
        t = src - i0
        for tap, wt in zip((i0 - 1, i0, i0 + 1, i0 + 2), (k2(t + 1), k1(t), k1(1 - t), k2(2 - t))):
            w[i, min(max(tap, 0), n_in - 1)] += wt
    return w


@dataclass
class Lin:
    name: str      # state_dict prefix (weight / bias), or a list of three for the fused q, k, v
    N: int
    K: int
    w_off: int = 0
    b_off: int = 0
    wd_off: int = 0


@dataclass
class LNP:
    name: str
    g_off: int = 0
    b_off: int = 0


class ViTPlan:
    def __init__(self, num_keypoints: int, downsample_factor: int, D: int, depth: int, heads: int, mlp: int, patch: int, n_pos: int):
        self.D, self.depth, self.heads, self.mlp, self.patch, self.n_pos = D, depth, heads, mlp, patch, n_pos
        pre = "backbone.vision_encoder"
        off = wd = 0

        def lin(name, N, K):
            nonlocal off, wd
            l = Lin(name, N, K, off, off + N * K, wd)
            off += N * K + N
            wd += N * K
            return l

        def ln(name):
            nonlocal off
            l = LNP(name, off, off + D)
            off += 2 * D
            return l

        self.cls_off = off
        off += D
        self.pos_off = off
        off += n_pos * D
        self.patch_lin = lin(f"{pre}.embeddings.patch_embeddings.projection", D, 3 * patch * patch)
        self.layers = []
        for i in range(depth):
            p = f"{pre}.layers.{i}"
            self.layers.append(dict(
                ln1=ln(f"{p}.layernorm_before"), qkv=lin(f"{p}.attention.qkv", 3 * D, D), proj=lin(f"{p}.attention.o_proj", D, D),
                ln2=ln(f"{p}.layernorm_after"), fc1=lin(f"{p}.mlp.fc1", mlp, D), fc2=lin(f"{p}.mlp.fc2", D, mlp)))
        self.lnf = ln(f"{pre}.layernorm")
        self.n_backbone = off
        self.head: list[ConvP] = build_head(D, patch, num_keypoints, downsample_factor)
        self.convs: list[ConvP] = []
        self.bns: list = []
        for c in self.head:
            c.w_off = off
            off += c.numel
            c.wd_off = wd
            wd += c.numel
            c.bias_off = off
            off += c.Ci
            self.convs.append(c)
        self.n_total, self.n_wd = off, wd
        self.n_running = 0

    def linears(self) -> list[Lin]:
        out = [self.patch_lin]
        for L in self.layers:
            out += [L["qkv"], L["proj"], L["fc1"], L["fc2"]]
        return out

    def norms(self) -> list[LNP]:
        out = []
        for L in self.layers:
            out += [L["ln1"], L["ln2"]]
        return out + [self.lnf]


class ViTEngine(Engine):
    """ViT-S/16 defaults = facebook/dino-vits16 (hidden 384, 12 layers, 6 heads, MLP 1536, patch 16, 224-px position table)."""

    wgrad_side_stream = False

    def __init__(self, num_keypoints: int, downsample_factor: int = 2, device: torch.device | str = "cuda:0", hidden: int = 384,
                 depth: int = 12, heads: int = 6, mlp: int = 1536, patch: int = 16, pretrain_grid: int = 14):
        self.device = torch.device(device)
        ops.require_device_type(self.device)
        self._lib = _lib.lib()
        self.K, self.ds = num_keypoints, downsample_factor
        if hidden % 64 or (hidden // heads) != 64 or mlp % 64:
            raise NotImplementedError("ViT widths must be multiples of 64 with 64-wide heads (ViT-S/B)")
        self.grid0 = pretrain_grid
        self.plan = ViTPlan(num_keypoints, downsample_factor, hidden, depth, heads, mlp, patch, 1 + pretrain_grid * pretrain_grid)
        n, dev = self.plan.n_total, self.device
        self.P = torch.zeros(n, device=dev, dtype=torch.float32)
        self.G = torch.zeros(n, device=dev, dtype=torch.float32)
        self.Wb = torch.zeros(n, device=dev, dtype=torch.bfloat16)
        self.Wd = torch.zeros(self.plan.n_wd, device=dev, dtype=torch.bfloat16)
        self.R = torch.zeros(0, device=dev, dtype=torch.float32)
        for l in self.plan.norms():
            self.P[l.g_off:l.g_off + hidden] = 1.0
        self.nbt = torch.zeros((), dtype=torch.long)
        self.sync_bn, self.process_group = False, None   # no BatchNorm here; kept for the DataParallel wrapper
        self.sync_bn_messages = 0
        self.grad_progress, self.single_backward = None, False   # gradient buckets leave during backward (Engine.backward, distributed.py)
        self._bwd_training = True
        self.profile = None
        self._wgrad_ws = None
        self._side, self._side_busy, self._side_keep = None, False, []
        self._fold = None
        self._interp: dict[tuple[int, int], torch.Tensor] = {}

    # ------------------------------------------------------------------------------------------------ params
    def _views(self, buf: torch.Tensor) -> dict[str, torch.Tensor]:
        pl, D = self.plan, self.plan.D
        pre = "backbone.vision_encoder"
        sd = {f"{pre}.embeddings.cls_token": buf[pl.cls_off:pl.cls_off + D].view(1, 1, D),
              f"{pre}.embeddings.position_embeddings": buf[pl.pos_off:pl.pos_off + pl.n_pos * D].view(1, pl.n_pos, D)}
        for l in pl.linears():
            w, b = buf[l.w_off:l.w_off + l.N * l.K], buf[l.b_off:l.b_off + l.N]
            if l.name.endswith("attention.qkv"):  # fused [q; k; v]: three reference parameters share one GEMM weight
                stem = l.name[:-len("qkv")]
                for j, nm in enumerate(("q_proj", "k_proj", "v_proj")):
                    sd[f"{stem}{nm}.weight"] = w[j * D * D:(j + 1) * D * D].view(D, D)
                    sd[f"{stem}{nm}.bias"] = b[j * D:(j + 1) * D]
            elif l is pl.patch_lin:
                sd[f"{l.name}.weight"] = w.view(D, 3, pl.patch, pl.patch)
                sd[f"{l.name}.bias"] = b
            else:
                sd[f"{l.name}.weight"] = w.view(l.N, l.K)
                sd[f"{l.name}.bias"] = b
        for l in pl.norms():
            sd[f"{l.name}.weight"] = buf[l.g_off:l.g_off + D]
            sd[f"{l.name}.bias"] = buf[l.b_off:l.b_off + D]
        for c in pl.head:
            sd[f"{c.name}.weight"] = self.param_view(c, "weight", buf=buf)
            sd[f"{c.name}.bias"] = self.param_view(c, "bias", buf=buf)
        return sd

    def state_dict(self) -> dict[str, torch.Tensor]:
        return self._views(self.P)

    def grad_views(self) -> dict[str, torch.Tensor]:
        return self._views(self.G)

    _LEGACY = (("encoder.layer.", "layers."), ("attention.attention.query", "attention.q_proj"),
               ("attention.attention.key", "attention.k_proj"), ("attention.attention.value", "attention.v_proj"),
               ("attention.output.dense", "attention.o_proj"), ("intermediate.dense", "mlp.fc1"), ("output.dense", "mlp.fc2"))

    @classmethod
    def canonical_key(cls, k: str) -> str:
        """transformers 4.x ViT parameter name -> the 5.x name used here (checkpoints saved by older reference installs)"""
        if ".encoder.layer." in k:
            for a, b in cls._LEGACY:
                k = k.replace(a, b)
        return k

    @torch.no_grad()
    def load_state_dict(self, sd: dict[str, torch.Tensor], strict: bool = True) -> None:
        sd = {self.canonical_key(k): v for k, v in sd.items()}
        own = self.state_dict()
        missing = [k for k in own if k not in sd]
        unexpected = [k for k in sd if k not in own]
        if strict and (missing or unexpected):
            raise KeyError(f"state_dict mismatch: missing {missing[:5]}..., unexpected {unexpected[:5]}...")
        for k, dst in own.items():
            if k in sd:
                dst.copy_(sd[k].to(device=self.device, dtype=torch.float32).reshape(dst.shape))
        self.refresh_weight_copies()

    def refresh_dgrad_copies(self, lo: int = 0, hi: int | None = None) -> None:
        hi = self.plan.n_total if hi is None else hi
        for l in self.plan.linears():
            if lo <= l.w_off < hi:  # [N][K] -> [K][N]
                check(self._lib.lp_permute_cba(_p(self.Wb[l.w_off:]), l.N, 1, l.K, _p(self.Wd[l.wd_off:]), ops._stream()), "lp_permute_cba")
        for c in self.plan.convs:
            if lo <= c.w_off < hi:
                check(self._lib.lp_permute_cba(_p(self.Wb[c.w_off:]), c.Co, c.k * c.k, c.Ci, _p(self.Wd[c.wd_off:]), ops._stream()),
                      "lp_permute_cba")

    # ------------------------------------------------------------------------------------------------ kernels
    def _gemm(self, a, lda, b, ldb, M, N, K, out, ldc, n_store=0, bias=None, batch=None):
        gb = _lib.GemmBatch(*batch) if batch is not None else None
        check(self._lib.lp_gemm_nt(a, lda, b, ldb, _p(out), None, ldc, M, N, K, n_store, _p(bias), C.byref(gb) if gb else None,
                                   ops._stream()), "lp_gemm_nt")

    def _gemm_tn(self, x, ldx, y, ldy, M, J, N, out, ldo, batch):
        """out[z][j][n] = sum_m x[z][m][j] y[z][m][n] (both operands contracted over their rows; no transposed copies)"""
        gb = _lib.GemmBatch(*batch)
        check(self._lib.lp_gemm_tn(x, ldx, y, ldy, _p(out), ldo, M, J, N, C.byref(gb), ops._stream()), "lp_gemm_tn")

    def _linear(self, x: torch.Tensor, l: Lin, M: int) -> torch.Tensor:
        out = torch.empty(M, l.N, device=self.device, dtype=torch.bfloat16)
        self._timed("lp_gemm_nt<linear fwd>", 2.0 * M * l.N * l.K, lambda: self._gemm(
            _p(x), l.K, _p(self.Wb[l.w_off:]), l.K, M, l.N, l.K, out, l.N, bias=self.P[l.b_off:l.b_off + l.N]),
            nbytes=2.0 * (M * l.K + M * l.N + l.N * l.K))
        return out

    def _linear_gelu(self, x: torch.Tensor, l: Lin, M: int) -> tuple[torch.Tensor, torch.Tensor]:
        """(u, GELU(u)) for u = x W^T + b: one launch whose store pass writes both (lp_gemm_nt_gelu_fwd, round 6); LP_VIT_GELU_FUSED=0 or a
        shape the pipelined kernel does not tile: the Linear layer, then lp_gelu_fwd"""
        u = torch.empty(M, l.N, device=self.device, dtype=torch.bfloat16)
        act = torch.empty_like(u)
        if os.environ.get("LP_VIT_GELU_FUSED", "1") not in ("0", "bwd"):
            rc = self._timed("lp_gemm_nt<linear fwd+gelu>", 2.0 * M * l.N * l.K, lambda: self._lib.lp_gemm_nt_gelu_fwd(
                _p(x), _p(self.Wb[l.w_off:]), _p(self.P[l.b_off:l.b_off + l.N]), _p(u), _p(act), M, l.N, l.K, ops._stream()),
                nbytes=2.0 * (M * l.K + 2 * M * l.N + l.N * l.K))
            if rc == 0:
                return u, act
            if rc != -2:   # (LP_ERR_UNSUPPORTED)
                check(rc, "lp_gemm_nt_gelu_fwd")
        u = self._linear(x, l, M)
        check(self._lib.lp_gelu_fwd(_p(u), u.numel(), _p(act), ops._stream()), "lp_gelu_fwd")
        return u, act

    def _linear_bwd(self, l: Lin, x: torch.Tensor, dy: torch.Tensor, M: int, need_dx: bool = True, bias_done: bool = False,
                    gelu_of: tuple[torch.Tensor, torch.Tensor] | None = None):
        """bias / weight gradients into G; returns dX = dY W (bf16) if wanted.  ``bias_done``: the kernel that produced ``dy`` already left its
        column sums in G (lp_gelu_bwd_colsum / lp_layernorm_bwd_bf16_colsum), so the weight gradient runs without the bias pass - which lets the
        pipelined weight-gradient kernel take it."""
        rows, cols = self._wg_shape
        g = _lib.ConvGeom(1, rows, cols, l.K, rows, cols, l.N, 1, 1, 1, 0) if rows * cols == M else _lib.ConvGeom(1, 1, M, l.K, 1, M, l.N, 1, 1, 1, 0)
        # bias gradient = column sums of dy, same pass
        self._timed("conv_wgrad_kernel<linear wgrad" + ("" if bias_done else "+bias") + ">", 2.0 * M * l.N * l.K,
                    lambda: self._wgrad(x, dy, g, self.G[l.w_off:], dbias=None if bias_done else self.G[l.b_off:]),
                    nbytes=2.0 * (M * l.K + M * l.N) + 4.0 * l.N * l.K)
        if not need_dx:
            return None
        dx = torch.empty(M, l.K, device=self.device, dtype=torch.bfloat16)
        if gelu_of is not None:
            # this layer's input is GELU(u): the data gradient leaves the store pass as the gradient of u (lp_gemm_nt_gelu_bwd), its column sums
            # - the bias gradient of the layer that produced u - as fixed-point totals; a shape the pipelined kernel does not tile: two passes
            u, dbias_u = gelu_of
            sums = torch.zeros(4 * l.K, device=self.device, dtype=torch.int64)
            rc = self._timed("lp_gemm_nt<linear dgrad+gelu>", 2.0 * M * l.N * l.K, lambda: self._lib.lp_gemm_nt_gelu_bwd(
                _p(dy), _p(self.Wd[l.wd_off:]), _p(u), _p(dx), M, l.K, l.N, _p(sums), ops._stream()),
                nbytes=2.0 * (2 * M * l.K + M * l.N + l.N * l.K))
            if rc == 0:
                check(self._lib.lp_fxsum_accumulate(_p(sums), l.K, _p(dbias_u), ops._stream()), "lp_fxsum_accumulate")
                return dx
            if rc != -2:   # (LP_ERR_UNSUPPORTED, include/lp_hip.h: a shape outside the pipelined tiles)
                check(rc, "lp_gemm_nt_gelu_bwd")
            d_a = torch.empty_like(dx)
            self._gemm(_p(dy), l.N, _p(self.Wd[l.wd_off:]), l.N, M, l.K, l.N, d_a, l.K)
            check(self._lib.lp_gelu_bwd_colsum(_p(u), _p(d_a), M, l.K, _p(dx), _p(dbias_u), ops._stream()), "lp_gelu_bwd_colsum")
            return dx
        self._timed("lp_gemm_nt<linear dgrad>", 2.0 * M * l.N * l.K, lambda: self._gemm(_p(dy), l.N, _p(self.Wd[l.wd_off:]), l.N, M, l.K, l.N, dx, l.K),
                    nbytes=2.0 * (M * l.K + M * l.N + l.N * l.K))
        return dx

    def _ln(self, x, delta, l: LNP, M: int, drop_T: int = 0):
        D = self.plan.D
        xo = torch.empty_like(x) if delta is not None else None
        rows = M - M // drop_T if drop_T else M
        y = torch.empty(rows, D, device=self.device, dtype=torch.bfloat16)
        mean = torch.empty(M, device=self.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        check(self._lib.lp_layernorm_fwd(_p(x), _p(delta), _p(xo), _p(self.P[l.g_off:]), _p(self.P[l.b_off:]), LN_EPS, M, D, drop_T, _p(y),
                                         _p(mean), _p(rstd), ops._stream()), "lp_layernorm_fwd")
        return y, mean, rstd, (xo if xo is not None else x)

    def _ln_bwd(self, dy, x, mean, rstd, l: LNP, M: int, dx, drop_T: int = 0, want_bf16: bool = False, colsum: torch.Tensor | None = None):
        """dx (fp32, the residual stream's gradient) += LayerNorm backward of dy; ``want_bf16``: also return the updated dx in bf16;
        ``colsum``: accumulate that bf16 tensor's column sums there (the bias gradient of the Linear layer it is the dy of)"""
        if want_bf16 and colsum is not None:
            out = torch.empty(M, self.plan.D, device=self.device, dtype=torch.bfloat16)
            check(self._lib.lp_layernorm_bwd_bf16_colsum(_p(dy), _p(x), _p(mean), _p(rstd), _p(self.P[l.g_off:]), M, self.plan.D, drop_T, _p(dx),
                                                         _p(out), _p(self.G[l.g_off:]), _p(self.G[l.b_off:]), _p(colsum), ops._stream()),
                  "lp_layernorm_bwd_bf16_colsum")
            return out
        if not want_bf16:
            check(self._lib.lp_layernorm_bwd(_p(dy), _p(x), _p(mean), _p(rstd), _p(self.P[l.g_off:]), M, self.plan.D, drop_T, _p(dx),
                                             _p(self.G[l.g_off:]), _p(self.G[l.b_off:]), ops._stream()), "lp_layernorm_bwd")
            return None
        out = torch.empty(M, self.plan.D, device=self.device, dtype=torch.bfloat16)
        check(self._lib.lp_layernorm_bwd_bf16(_p(dy), _p(x), _p(mean), _p(rstd), _p(self.P[l.g_off:]), M, self.plan.D, drop_T, _p(dx), _p(out),
                                              _p(self.G[l.g_off:]), _p(self.G[l.b_off:]), ops._stream()), "lp_layernorm_bwd_bf16")
        return out

    def _transpose(self, src_ptr, R, Cc, ldi, in_b, in_h, out, ldo, out_b, out_h, nb, nh):
        check(self._lib.lp_transpose_batched(src_ptr, R, Cc, ldi, in_b, in_h, _p(out), ldo, out_b, out_h, nb, nh, ops._stream()),
              "lp_transpose_batched")

    def _pos(self, gh: int, gw: int) -> torch.Tensor:
        """(1 + gh*gw, D) position embeddings: the [CLS] row + the pretraining grid resized bicubically (HF interpolate_pos_encoding)"""
        pl, D = self.plan, self.plan.D
        pos = self.P[pl.pos_off:pl.pos_off + pl.n_pos * D].view(pl.n_pos, D)
        if gh == self.grid0 and gw == self.grid0:
            return pos
        key = (gh, gw)
        if key not in self._interp:
            w = np.kron(bicubic_matrix(self.grid0, gh), bicubic_matrix(self.grid0, gw)).astype(np.float32)
            self._interp[key] = torch.from_numpy(w).to(self.device).contiguous()
        out = torch.empty(1 + gh * gw, D, device=self.device, dtype=torch.float32)
        out[0].copy_(pos[0])
        check(self._lib.lp_small_matmul(_p(self._interp[key]), _p(pos[1:]), gh * gw, pl.n_pos - 1, D, 0, 0, _p(out[1:]), ops._stream()),
              "lp_small_matmul")
        return out

    # ------------------------------------------------------------------------------------------------ forward
    def can_segment(self, n0: int, H: int, W: int) -> bool:
        """no batch statistics anywhere (LayerNorm is per token): two batches of equally sized images always share one pass"""
        return n0 > 0

    def forward(self, images, training: bool = True) -> tuple[torch.Tensor, Tape]:
        """``images``: a tensor, or a pair (labeled, unlabeled frames) that runs as one batch"""
        return self._forward(images, training, keep=True)

    def forward_infer(self, images: torch.Tensor) -> torch.Tensor:
        """Inference forward (predict_step under eval + no_grad): the attention probabilities are never written (lp_attn_fwd with
        p = NULL drops the 2 x 0.85 GB per layer the training pass stores for its backward) and no tape is kept."""
        return self._forward(images, False, keep=False)[0]

    def _forward(self, images, training: bool, keep: bool) -> tuple[torch.Tensor, Tape]:
        parts = list(images) if isinstance(images, (list, tuple)) else [images]
        for p_ in parts:
            ops.require_device(p_)
        parts = [p_.to(torch.float32).contiguous() for p_ in parts]
        if any(p_.shape[1:] != parts[0].shape[1:] for p_ in parts):
            raise ValueError("a joint pass takes batches of equally sized images")
        if parts[0].dim() != 4 or parts[0].shape[1] != 3:
            raise ValueError(f"images must be (B, 3, H, W), got {tuple(parts[0].shape)}")   # (raw pointers from here on)
        _, _, H, W = parts[0].shape
        B = sum(p_.shape[0] for p_ in parts)
        pl = self.plan
        D, nh, pt = pl.D, pl.heads, pl.patch
        if H % pt or W % pt:
            raise ValueError(f"image size must be a multiple of the patch size {pt}, got {H}x{W}")
        gh, gw = H // pt, W // pt
        Np, Tn = gh * gw, gh * gw + 1
        M, Tp = B * Tn, -(-Tn // 64) * 64
        tp = Tape()
        T = tp.t
        dev = self.device
        self._wg_shape = (B, Tn)

        patches = torch.empty(B * Np, 3 * pt * pt, device=dev, dtype=torch.bfloat16)
        i0 = 0
        for p_ in parts:
            check(self._lib.lp_vit_patchify(_p(p_), p_.shape[0], H, W, pt, _p(patches[i0 * Np:]), ops._stream()), "lp_vit_patchify")
            i0 += p_.shape[0]
        pe = self._linear(patches, pl.patch_lin, B * Np)
        x = torch.empty(M, D, device=dev, dtype=torch.float32)
        pos = self._pos(gh, gw)  # (kept alive across the launch)
        check(self._lib.lp_vit_tokens_fwd(_p(pe), _p(self.P[pl.cls_off:]), _p(pos), B, Np, D, _p(x), ops._stream()), "lp_vit_tokens_fwd")
        if keep:
            T["patches"] = patches
        delta = None
        scale = 1.0 / math.sqrt(D // nh)
        qs = 3 * D  # row pitch of the fused qkv tensor
        for i, L in enumerate(pl.layers):
            y1, m1, r1, x = self._ln(x, delta, L["ln1"], M)
            qkv = self._linear(y1, L["qkv"], M)
            # P = softmax(Q K^T / 8) (bf16, row pitch Tp, pad columns zero; kept for the backward pass) and attn = P V in one kernel:
            # the scores themselves never reach memory
            S = torch.empty(B * nh * Tn, Tp, device=dev, dtype=torch.bfloat16) if keep else None
            attn = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
            # scores + probabilities x values: 2 products of B * nh * Tn * Tn * (D / nh) MACs; the training pass computes the scores twice
            self._timed("attn_fwd_kernel", 4.0 * B * Tn * Tn * D, lambda: check(self._lib.lp_attn_fwd(
                _p(qkv), qs, D, 2 * D, B, nh, Tn, scale, _p(S), Tp, _p(attn), D, ops._stream()), "lp_attn_fwd"),
                nbytes=2.0 * (3 * B * Tn * D + B * Tn * D) + (2.0 * B * nh * Tn * Tp if S is not None else 0.0))
            proj = self._linear(attn, L["proj"], M)
            x_in = x
            y2, m2, r2, x = self._ln(x, proj, L["ln2"], M)
            h1, a1 = self._linear_gelu(y2, L["fc1"], M)
            delta = self._linear(a1, L["fc2"], M)
            if keep:
                for nm, v in (("x_in", x_in), ("m1", m1), ("r1", r1), ("y1", y1), ("qkv", qkv), ("P", S), ("attn", attn), ("x_mid", x),
                              ("m2", m2), ("r2", r2), ("y2", y2), ("h1", h1), ("a1", a1)):
                    T[f"l{i}.{nm}"] = v
        feat, mf, rf, x = self._ln(x, delta, pl.lnf, M, drop_T=Tn)
        if keep:
            T["x_last"], T["mf"], T["rf"] = x, mf, rf
        heat = self._head_forward(feat.view(B, gh, gw, D), B, gh, gw, T if keep else {})
        tp.meta.update(B=B, H=H, W=W, gh=gh, gw=gw, training=training)
        return heat, tp

    # ------------------------------------------------------------------------------------------------ backward
    def backward(self, tp: Tape, g_heat: torch.Tensor, trace: dict | None = None) -> None:
        T, pl = tp.t, self.plan
        B, gh, gw = tp.meta["B"], tp.meta["gh"], tp.meta["gw"]
        D, nh = pl.D, pl.heads
        Np, Tn = gh * gw, gh * gw + 1
        M, Tp = B * Tn, -(-Tn // 64) * 64
        dev = self.device
        self._wg_shape = (B, Tn)
        scale = 1.0 / math.sqrt(D // nh)
        qs = 3 * D

        d_feat = self._head_backward(T, B, g_heat)                      # (B, gh, gw, D) bf16
        progress = self.grad_progress if (self.grad_progress is not None and self.single_backward) else None
        if progress is not None:
            progress(pl.n_backbone)  # the head's gradients (the tail of the flat buffer) are complete
        dx = torch.zeros(M, D, device=dev, dtype=torch.float32)         # gradient of the residual stream
        # every LayerNorm backward also leaves the updated stream gradient in bf16: it is the operand of the next Linear backward
        # ... and its column sums are the bias gradient of that layer (fc2 of the last block here), so no weight gradient needs a bias pass
        fuse_bias = os.environ.get("LP_VIT_BIAS_FUSED", "1") != "0"
        fuse_gelu = os.environ.get("LP_VIT_GELU_FUSED", "1") != "0"   # (0: A/B runs - lp_gemm_nt, then lp_gelu_bwd_colsum)
        bsum = (lambda lin: self.G[lin.b_off:lin.b_off + lin.N]) if fuse_bias else (lambda lin: None)
        dx16 = self._ln_bwd(d_feat, T["x_last"], T["mf"], T["rf"], pl.lnf, M, dx, drop_T=Tn, want_bf16=True, colsum=bsum(pl.layers[-1]["fc2"]))

        for i in range(pl.depth - 1, -1, -1):
            L = pl.layers[i]
            t = lambda nm: T[f"l{i}.{nm}"]  # noqa: E731
            if trace is not None:
                trace[f"l{i}.dout"] = dx.clone()
            # ---- MLP branch: x_out = x_mid + fc2(gelu(fc1(LN2(x_mid))))
            dmlp = dx16
            if fuse_bias and fuse_gelu:   # GELU's backward inside fc2's data gradient (round 6): d_a1 is never written
                d_h1 = self._linear_bwd(L["fc2"], t("a1"), dmlp, M, bias_done=True, gelu_of=(t("h1"), bsum(L["fc1"])))
                d_a1 = None
            else:
                d_a1 = self._linear_bwd(L["fc2"], t("a1"), dmlp, M, bias_done=fuse_bias)
                d_h1 = torch.empty_like(d_a1)
            if d_a1 is None:
                pass
            elif fuse_bias:
                check(self._lib.lp_gelu_bwd_colsum(_p(t("h1")), _p(d_a1), M, pl.mlp, _p(d_h1), _p(bsum(L["fc1"])), ops._stream()), "lp_gelu_bwd_colsum")
            else:
                check(self._lib.lp_gelu_bwd(_p(t("h1")), _p(d_a1), d_a1.numel(), _p(d_h1), ops._stream()), "lp_gelu_bwd")
            d_y2 = self._linear_bwd(L["fc1"], t("y2"), d_h1, M, bias_done=fuse_bias)
            # (proj keeps the bias sums inside its weight-gradient launch: a 384 x 384 layer gains 5 us from the pipelined kernel, the
            #  column sums cost the LayerNorm backward 40 - profiles/archive/r03_final_vit_kernel_stats.txt)
            dx16 = self._ln_bwd(d_y2, t("x_mid"), t("m2"), t("r2"), L["ln2"], M, dx, want_bf16=True)
            # ---- attention branch: x_mid = x_in + proj(softmax(Q K^T / 8) V)
            dproj = dx16
            d_attn = self._linear_bwd(L["proj"], t("attn"), dproj, M)
            qkv, Pm = t("qkv"), t("P")
            dqkv = torch.empty(M, qs, device=dev, dtype=torch.bfloat16)
            zP = (nh * Tn * Tp, Tn * Tp)       # batch strides of a [B][nh][Tn][Tp] tensor
            zT = (nh * 64 * Tp, 64 * Tp)       # ... of a [B][nh][64][Tp] transposed head slice
            # D = rowsum(dO o O) (== rowsum(dP o P) because O = P V), then ONE pass over the stored probabilities leaves
            # dS = scale * P o (dO V^T - D), dV = P^T dO and dK = dS^T Q (dP never exists; P and dS are touched once each here)
            drow = torch.empty(M, nh, device=dev, dtype=torch.float32)
            check(self._lib.lp_attn_rowdot(_p(d_attn), _p(t("attn")), M, nh, D, _p(drow), ops._stream()), "lp_attn_rowdot")
            dS = torch.empty(B * nh * Tn, Tp, device=dev, dtype=torch.bfloat16)
            # dP = dO V^T, dV = P^T dO, dK = dS^T Q: 3 products of B * Tn * Tn * D MACs
            self._timed("attn_bwd_kv_kernel", 6.0 * B * Tn * Tn * D, lambda: check(self._lib.lp_attn_bwd_kv(
                _p(qkv), qs, 2 * D, _p(d_attn), D, _p(Pm), Tp, _p(drow), B, nh, Tn, scale, _p(dS), _p(dqkv), qs, D, 2 * D, ops._stream()),
                "lp_attn_bwd_kv"), nbytes=2.0 * (2 * B * nh * Tn * Tp + 6 * B * Tn * D))
            # dQ = dS K
            tmpT = torch.empty(B * nh * 64, Tp, device=dev, dtype=torch.bfloat16)      # a transposed [64][Tp] head slice
            self._transpose(qkv[:, D:].data_ptr(), Tn, 64, qs, Tn * qs, 64, tmpT, Tp, *zT, B, nh)
            self._gemm(_p(dS), Tp, _p(tmpT), Tp, Tn, 64, Tp, dqkv, qs, batch=(B, nh, *zP, *zT, Tn * qs, 64))
            if trace is not None:
                trace[f"l{i}.dqkv"] = dqkv
            d_y1 = self._linear_bwd(L["qkv"], t("y1"), dqkv, M)
            dx16 = self._ln_bwd(d_y1, t("x_in"), t("m1"), t("r1"), L["ln1"], M, dx, want_bf16=i > 0,
                                colsum=bsum(pl.layers[i - 1]["fc2"]) if i > 0 else None)
            if progress is not None:
                progress(L["ln1"].g_off)  # everything from this layer's first parameter to the end of G is final
        if trace is not None:
            trace["tokens.dx"] = dx
        # ---- embeddings
        dpatch = torch.empty(B * Np, D, device=dev, dtype=torch.bfloat16)
        dpos = torch.empty(Tn, D, device=dev, dtype=torch.float32)
        check(self._lib.lp_vit_tokens_bwd(_p(dx), B, Np, D, _p(dpatch), _p(dpos), ops._stream()), "lp_vit_tokens_bwd")
        self.G[pl.cls_off:pl.cls_off + D] += dpos[0]
        gpos = self.G[pl.pos_off:pl.pos_off + pl.n_pos * D].view(pl.n_pos, D)
        gpos[0] += dpos[0]
        if gh == self.grid0 and gw == self.grid0:
            gpos[1:] += dpos[1:]
        else:
            check(self._lib.lp_small_matmul(_p(self._interp[(gh, gw)]), _p(dpos[1:]), Np, pl.n_pos - 1, D, 1, 1, _p(gpos[1:]), ops._stream()),
                  "lp_small_matmul(adjoint)")
        self._wg_shape = (B, Np)
        self._linear_bwd(pl.patch_lin, T["patches"], dpatch, B * Np, need_dx=False)
        self._join_side_stream()
