"""Load the CPU-emulated build of the kernel sources (tests/hipemu/liblp_emu.so) and call its C ABI with numpy arrays.

TEST INFRASTRUCTURE ONLY: checks kernel logic in the GPU-less container.  The product never loads this library.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

import _lp_bootstrap  # noqa: F401
from lightning_pose_amd import _lib, _tables

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
BUILD = os.path.join(ROOT, "lightning-pose_amd", "csrc", "build.sh")

_emu = None
BACKEND = "emu"  # "emu": CPU-emulated build, numpy buffers;  "gpu": the product liblp_hip.so, torch ROCm buffers


def emu_lib() -> C.CDLL:
    global _emu
    if _emu is None:
        subprocess.run(["bash", BUILD, "emu"], check=True, capture_output=True)
        _emu = _lib.declare(C.CDLL(os.path.join(HERE, "liblp_emu.so")))
    return _emu


def lib() -> C.CDLL:
    return _lib.lib() if BACKEND == "gpu" else emu_lib()


def stream():
    if BACKEND == "gpu":
        import torch
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)
    return None


class Buf:
    """A kernel argument buffer: numpy memory on the emulator, a torch ROCm tensor on the GPU."""

    def __init__(self, a: np.ndarray):
        a = np.ascontiguousarray(a)
        self.dtype, self.shape = a.dtype, a.shape
        if BACKEND == "gpu":
            import torch
            raw = a.view(np.int16) if a.dtype == np.uint16 else a
            self.t = torch.from_numpy(raw.copy()).cuda()
            self.p = C.c_void_p(self.t.data_ptr())
        else:
            self.a = a.copy()
            self.p = self.a.ctypes.data_as(C.c_void_p)

    def np(self) -> np.ndarray:
        if BACKEND == "gpu":
            import torch
            torch.cuda.synchronize()
            out = self.t.cpu().numpy()
            return out.view(np.uint16) if self.dtype == np.uint16 else out
        return self.a


def ptr(a):
    """Device pointer of a Buf (or None)."""
    if a is None:
        return None
    return a.p


def f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def i32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def B(a, dtype=None):
    """Stage an input array (None passes through)."""
    if a is None:
        return None
    a = np.asarray(a) if dtype is None else np.asarray(a, dtype=dtype)
    return Buf(a)


def Z(shape, dtype=np.float32):
    return Buf(np.zeros(shape, dtype))


def ok(rc):
    assert rc == 0, rc


class Tables:
    """Keeps the staged tables alive next to the ctypes struct."""

    def __init__(self, h: int, w: int, ds: int):
        ty, tx = _tables.axis_tables(h, ds), _tables.axis_tables(w, ds)
        self.keep = [Buf(a) for a in (ty["row_base"], ty["row_taps"], tx["col_start"], tx["col_taps"], tx["colT_start"], tx["colT_taps"])]
        self.struct = _lib.DecodeTables(*[b.p for b in self.keep], ty["ty"], tx["tx"], tx["tc"])


def frame_map(transforms=None, tf_mode=_lib.TF_NONE, bbox=None, views=1, K=1, model_h=1.0, model_w=1.0):
    keep = [B(transforms, np.float32), B(bbox, np.float32)]
    fm = _lib.FrameMap(ptr(keep[0]), tf_mode, ptr(keep[1]), 4 * views, max(1, K // views), float(model_h), float(model_w))
    return fm, keep


def decode_fwd(heat, ds, temperature=1000.0, fm=None, prune=0):
    heat = f32(heat)
    b, k, h, w = heat.shape
    tb = Tables(h, w, ds)
    keep = None
    if fm is None:
        fm, keep = frame_map(K=k)
    hb = Buf(heat)
    kp_aug, kp_frame, conf, stats = Z((b, k, 2)), Z((b, k, 2)), Z((b, k)), Z((b, k, 4))
    ok(lib().lp_decode_fwd(hb.p, b, k, h, w, ds, temperature, C.byref(tb.struct), C.byref(fm), kp_aug.p, kp_frame.p, conf.p, stats.p,
                           int(prune), stream()))
    return kp_aug.np(), kp_frame.np(), conf.np(), stats.np()


def decode_bwd(heat, ds, stats, g_aug=None, g_frame=None, temperature=1000.0, fm=None, prune=0):
    heat = f32(heat)
    b, k, h, w = heat.shape
    tb = Tables(h, w, ds)
    keep = None
    if fm is None:
        fm, keep = frame_map(K=k)
    hb, sb, ga, gf = Buf(heat), Buf(f32(stats)), B(g_aug, np.float32), B(g_frame, np.float32)
    g_heat = Z(heat.shape)
    ok(lib().lp_decode_bwd(hb.p, b, k, h, w, ds, temperature, C.byref(tb.struct), C.byref(fm), sb.p, ptr(ga), ptr(gf), g_heat.p, 0,
                           int(prune), stream()))
    return g_heat.np()


def heatmap_gen(kp, vis, img_h, img_w, h, w, sigma=1.25):
    kp = f32(kp)
    b, k, _ = kp.shape
    kb, vb, out = Buf(kp), B(vis, np.int32), Z((b, k, h, w))
    ok(lib().lp_heatmap_gen(kb.p, ptr(vb), b, k, img_h, img_w, h, w, sigma, out.p, stream()))
    return out.np()


def heatmap_mse(targ, pred, gout=1.0):
    targ, pred = f32(targ), f32(pred)
    b, k, h, w = pred.shape
    tb, pb = Buf(targ), Buf(pred)
    ws = Z(lib().lp_heatmap_mse_workspace_bytes(b, k), np.uint8)
    loss, g, go = Z(1), Z(pred.shape), Buf(f32([gout]))
    ok(lib().lp_heatmap_mse_fwd(tb.p, pb.p, b, k, h, w, loss.p, ws.p, stream()))
    ok(lib().lp_heatmap_mse_bwd(tb.p, pb.p, b, k, h, w, ws.p, go.p, g.p, 0, stream()))
    return loss.np()[0], g.np()


def heatmap_div(kind, targ, pred, gout=1.0):
    """kind: _lib.HM_KL / _lib.HM_JS (or HM_MSE) -> (loss, grad wrt pred)"""
    targ, pred = f32(targ), f32(pred)
    b, k, h, w = pred.shape
    tb, pb = Buf(targ), Buf(pred)
    ws = Z(lib().lp_heatmap_mse_workspace_bytes(b, k), np.uint8)
    loss, g, go = Z(1), Z(pred.shape), Buf(f32([gout]))
    ok(lib().lp_heatmap_loss_fwd(kind, tb.p, pb.p, b, k, h, w, loss.p, ws.p, stream()))
    ok(lib().lp_heatmap_loss_bwd(kind, tb.p, pb.p, b, k, h, w, ws.p, go.p, g.p, 0, stream()))
    return loss.np()[0], g.np()


def unimodal_mse(kp_aug, pred, conf, img_h, img_w, thr, sigma=1.25, gout=1.0):
    kp_aug, pred, conf = f32(kp_aug), f32(pred), f32(conf)
    s, k, h, w = pred.shape
    kb, pb, cb = Buf(kp_aug), Buf(pred), Buf(conf)
    ws = Z(lib().lp_heatmap_mse_workspace_bytes(s, k), np.uint8)
    loss, g, go = Z(1), Z(pred.shape), Buf(f32([gout]))
    ok(lib().lp_unimodal_mse_fwd(kb.p, pb.p, cb.p, s, k, img_h, img_w, h, w, sigma, thr, loss.p, ws.p, stream()))
    ok(lib().lp_unimodal_mse_bwd(kb.p, pb.p, s, k, img_h, img_w, h, w, sigma, ws.p, go.p, g.p, 0, stream()))
    return loss.np()[0], g.np()


def softmax2d(logits_nhwc, K):
    """logits (B, n, C) channel-padded -> prob (B, K, n)"""
    x = f32(logits_nhwc)
    b, n, c = x.shape
    xb, out = Buf(x), Z((b, K, n))
    ok(lib().lp_softmax2d_fwd(xb.p, n * c, c, 1, b, K, n, out.p, stream()))
    return out.np()


def softmax2d_bwd(prob, gprob, Cn):
    prob, gprob = f32(prob), f32(gprob)
    b, k, n = prob.shape
    fill = 0x7fc0 if Cn % 8 == 0 else 0  # (bf16 NaN bits where the pixel-major kernel owns the whole row)
    pb, gb, gin = Buf(prob), Buf(gprob), Buf(np.full((b, n, Cn), fill, np.uint16))
    ok(lib().lp_softmax2d_bwd(pb.p, gb.p, b, k, n, gin.p, n * Cn, Cn, 1, stream()))
    return from_bf16_bits(gin.np()).numpy()


def temporal(kp, conf, eps, thr):
    kp = f32(kp)
    s, k, _ = kp.shape
    kb, cb = Buf(kp), B(conf, np.float32)
    eb = Buf(f32(np.broadcast_to(np.asarray(eps, np.float32), (k,))))
    loss, g = Z(1), Z(kp.shape)
    ok(lib().lp_temporal_fwd_bwd(kb.p, ptr(cb), s, k, eb.p, thr, loss.p, g.p, stream()))
    return loss.np()[0], g.np()


def pca(kp, index, mean, kept, eps):
    kp = f32(kp)
    s, k, _ = kp.shape
    idx = i32(index)
    rows, pts = idx.shape
    kept = f32(kept)
    kb, ib, mb, vb = Buf(kp), Buf(idx), Buf(f32(mean)), Buf(kept)
    loss, g = Z(1), Z(kp.shape)
    ok(lib().lp_pca_fwd_bwd(kb.p, s, k, ib.p, rows, pts, mb.p, vb.p, kept.shape[0], float(eps), loss.p, g.p, stream()))
    return loss.np()[0], g.np()


def rmse(targ, pred):
    targ, pred = f32(targ), f32(pred)
    tb, pb, loss = Buf(targ), Buf(pred), Z(1)
    ok(lib().lp_rmse_fwd(tb.p, pb.p, targ.size // 2, loss.p, stream()))
    return loss.np()[0]


# ---- bf16 helpers / conv wrappers -------------------------------------------------------------------
import torch  # noqa: E402


def to_bf16_bits(t: "torch.Tensor") -> np.ndarray:
    return np.ascontiguousarray(t.contiguous().to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16))


def from_bf16_bits(a: np.ndarray) -> "torch.Tensor":
    return torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16).float()


def geom(Bn, Hi, Wi, Ci, Co, R, S, stride, pad, Ho=None, Wo=None):
    if Ho is None:
        Ho = (Hi + 2 * pad - R) // stride + 1
        Wo = (Wi + 2 * pad - S) // stride + 1
    return _lib.ConvGeom(Bn, Hi, Wi, Ci, Ho, Wo, Co, R, S, stride, pad)


def conv_fwd(x_nhwc_bits, w_bits, g, bias=None, f32_out=False, ldo=None, n_store=0):
    M = g.B * g.Ho * g.Wo
    ldo = ldo or g.Co
    xb, wb, bb = Buf(x_nhwc_bits), Buf(w_bits), B(bias, np.float32)
    ob = Z((M, ldo), np.uint16)
    of = Z((M, ldo)) if f32_out else None
    ok(lib().lp_conv_fwd(xb.p, wb.p, C.byref(g), ptr(bb), ob.p, ptr(of), ldo, n_store, stream()))
    return ob.np(), (of.np() if of is not None else None)


def conv_dgrad(dy_bits, wd_bits, g, addend_bits=None, bias=None, f32_out=False, ldo=None, n_store=0, mask_bits=None, into=None):
    """``into`` (bf16 bits of an existing gradient): accumulate in place (addend = output, skip_empty_classes = 1)"""
    M = g.B * g.Hi * g.Wi
    ldo = ldo or g.Ci
    db, wb, ab, bb, mb = Buf(dy_bits), Buf(wd_bits), B(addend_bits), B(bias, np.float32), B(mask_bits)
    if into is not None:
        ob = Buf(into)
        ok(lib().lp_conv_dgrad(db.p, wb.p, C.byref(g), None, ob.p, ptr(mb), ob.p, None, ldo, n_store, 1, stream()))
        return ob.np(), None
    ob = Z((M, ldo), np.uint16)
    of = Z((M, ldo)) if f32_out else None
    ok(lib().lp_conv_dgrad(db.p, wb.p, C.byref(g), ptr(bb), ptr(ab), ptr(mb), ob.p, ptr(of), ldo, n_store, 0, stream()))
    return ob.np(), (of.np() if of is not None else None)


def conv_dgrad_bits(dy_bits, wd_bits, g, relu_bits, addend_bits=None, into=None, skip=0):
    """lp_conv_dgrad_bits: the ReLU mask at 1 bit per element; ``into`` (bf16 bits of an existing gradient) = accumulate in place, as the
    projection shortcut's data gradient does (addend = output, skip_empty_classes)"""
    M = g.B * g.Hi * g.Wi
    db, wb, rb = Buf(dy_bits), Buf(wd_bits), Buf(relu_bits)
    if into is not None:
        ob = Buf(into)
        ok(lib().lp_conv_dgrad_bits(db.p, wb.p, C.byref(g), ob.p, rb.p, ob.p, skip, stream()))
    else:
        ab, ob = B(addend_bits), Z((M, g.Ci), np.uint16)
        ok(lib().lp_conv_dgrad_bits(db.p, wb.p, C.byref(g), ptr(ab), rb.p, ob.p, skip, stream()))
    return ob.np()


def ZX(shape):
    """zeroed fixed-point sums (include/lp_hip.h: lp_fxsum = {int64 hi, int64 lo}) of the given value shape"""
    return Z(tuple(shape) + (2,), np.int64)


def fx(raw: np.ndarray) -> np.ndarray:
    """the values of raw lp_fxsum words (..., 2) as fp32: hi * 2^-12 + lo * 2^-60 (lp_common.h: fx_value)"""
    return (raw[..., 0].astype(np.float64) * 2.0 ** -12 + raw[..., 1].astype(np.float64) * 2.0 ** -60).astype(np.float32)


def to_fx(v) -> np.ndarray:
    """fp32 values -> lp_fxsum words (..., 2), the split fx_add makes: hi = round(v * 2^12), lo = round((v - hi * 2^-12) * 2^60)"""
    v = np.asarray(v, np.float64)
    hi = np.rint(v * 2.0 ** 12)
    lo = np.rint((v - hi * 2.0 ** -12) * 2.0 ** 60)
    return np.stack([hi, lo], -1).astype(np.int64)


def _bn_fuse(Cn, z=None, mean=None, invstd=None, gamma=None, beta=None, mask_from_z=0, relu_bits=None, seg=0):
    """lp_bn_fuse + the buffers it points at (kept alive on the returned object).  seg > 0: two BatchNorm segments (images [0, seg)
    and the rest): sums is (2, 2, Cn), mean / invstd are (2, Cn)."""
    f = _lib.BnFuse()
    f.seg_images = seg
    f.keep = dict(z=B(z), mean=B(mean, np.float32), invstd=B(invstd, np.float32), gamma=B(gamma, np.float32), beta=B(beta, np.float32),
                  sums=ZX((2, 2, Cn) if seg else (2, Cn)))
    k = f.keep
    f.z, f.mean, f.invstd, f.gamma, f.beta = (ptr(k[n]).value if k[n] is not None else None for n in ("z", "mean", "invstd", "gamma", "beta"))
    f.mask_from_z = int(mask_from_z)
    k["bits"] = B(relu_bits)
    f.relu_bits = k["bits"].p.value if relu_bits is not None else None
    f.sums = k["sums"].p.value
    return f


def conv_fwd_bn(x_nhwc_bits, w_bits, g, seg=0, rc=False, raw=False):
    """-> (z bits, sums (2,Co) or (2,2,Co) with seg); rc=True: return the status code instead of asserting it; raw=True: the sums as
    their fixed-point words (..., 2) int64 instead of fp32 values"""
    xb, wb, ob = Buf(x_nhwc_bits), Buf(w_bits), Z((g.B * g.Ho * g.Wo, g.Co), np.uint16)
    f = _bn_fuse(g.Co, seg=seg)
    code = lib().lp_conv_fwd_bn(xb.p, wb.p, C.byref(g), ob.p, C.byref(f), stream())
    if rc:
        return code
    ok(code)
    words = f.keep["sums"].np()
    return ob.np(), (words if raw else fx(words))


def stem_fwd_bn(x4_bits, w_bits, g, seg=0):
    xb, wb, ob = Buf(x4_bits), Buf(w_bits), Z((g.B * g.Ho * g.Wo, 64), np.uint16)
    f = _bn_fuse(64, seg=seg)
    ok(lib().lp_stem_fwd_bn(xb.p, wb.p, C.byref(g), ob.p, C.byref(f), stream()))
    return ob.np(), fx(f.keep["sums"].np())


def conv_dgrad_bn(dy_bits, wd_bits, g, z_bits, mean, invstd, gamma=None, beta=None, addend_bits=None, mask_bits=None, relu_bits=None, seg=0,
                  rc=False, raw=False, addend_half=False):
    """-> (dx bits, sums (2,Ci), dbeta, dgamma); the ReLU mask comes from mask_bits (bf16 activation) or relu_bits (1 bit per
    element) if given, else is recomputed from z.  seg > 0: mean / invstd are (2, Ci), sums comes back (2, 2, Ci).  d beta / d gamma are
    the sums themselves added over the segments (the library adds them into the gradient buffer in lp_bn_bwd_apply: bn_backward)."""
    db, wb, ab, mb = Buf(dy_bits), Buf(wd_bits), B(addend_bits), B(mask_bits)
    ob = Z((g.B * g.Hi * g.Wi, g.Ci), np.uint16)
    f = _bn_fuse(g.Ci, z_bits, mean, invstd, gamma, beta, mask_from_z=mask_bits is None and relu_bits is None, relu_bits=relu_bits, seg=seg)
    f.addend_half = int(addend_half)   # (the addend on the half-resolution grid, added at the even pixels: lp_bn_fuse.addend_half)
    code = lib().lp_conv_dgrad_bn(db.p, wb.p, C.byref(g), ptr(ab), ptr(mb), ob.p, C.byref(f), stream())
    if rc:
        return code
    ok(code)
    words = f.keep["sums"].np()
    vals = fx(words)
    tot = vals.reshape(-1, 2, g.Ci).sum(0)
    return ob.np(), (words if raw else vals), tot[0], tot[1]


def gemm_nt(a_bits, lda, b_bits, ldb, M, N, K, ldc, c_rows, n_store=0, bias=None, batch=None, f32_out=False):
    """C[z][m][n] = sum_k A[z][m][k] B[z][n][k]; a_bits / b_bits flat uint16 buffers; batch = (nb, nh, a_b, a_h, b_b, b_h, c_b, c_h).
    Returns the flat C buffer of c_rows * ldc elements (bf16 bits, or fp32)."""
    ab, bb, bi = Buf(a_bits), Buf(b_bits), B(bias, np.float32)
    cb = Z(c_rows * ldc, np.uint16) if not f32_out else None
    cf = Z(c_rows * ldc) if f32_out else None
    gb = None
    if batch is not None:
        gb = _lib.GemmBatch(*batch)
    ok(lib().lp_gemm_nt(ab.p, lda, bb.p, ldb, ptr(cb), ptr(cf), ldc, M, N, K, n_store, ptr(bi), C.byref(gb) if gb else None, stream()))
    return cf.np() if f32_out else cb.np()


def gemm_nt_gelu_fwd(a_bits, b_bits, bias, M, N, K):
    """lp_gemm_nt_gelu_fwd: (c bits, GELU(c) bits), both (M, N)"""
    ab, bb, bi, cb, gb = Buf(a_bits), Buf(b_bits), B(bias, np.float32), Z((M, N), np.uint16), Z((M, N), np.uint16)
    ok(lib().lp_gemm_nt_gelu_fwd(ab.p, bb.p, ptr(bi), cb.p, gb.p, M, N, K, stream()))
    return cb.np(), gb.np()


def gemm_nt_gelu_bwd(a_bits, b_bits, u_bits, M, N, K, colsum=True):
    """lp_gemm_nt_gelu_bwd: (c bits (M, N), column sums of c as fp32 (N,) or None); raises Unsupported shapes as the library reports them"""
    ab, bb, ub, cb = Buf(a_bits), Buf(b_bits), Buf(u_bits), Z((M, N), np.uint16)
    sums = ZX((2, N)) if colsum else None
    ok(lib().lp_gemm_nt_gelu_bwd(ab.p, bb.p, ub.p, cb.p, M, N, K, ptr(sums), stream()))
    return cb.np(), (fx(sums.np())[0] if colsum else None)


def gemm_tn(x_bits, ldx, y_bits, ldy, M, J, N, ldo, o_elems, batch=None):
    """out[z][j][n] = sum_m x[z][m][j] y[z][m][n]; batch = (nb, nh, x_b, x_h, y_b, y_h, o_b, o_h). Returns the flat bf16 output."""
    xb, yb, ob = Buf(x_bits), Buf(y_bits), Z(o_elems, np.uint16)
    gb = _lib.GemmBatch(*batch) if batch is not None else None
    ok(lib().lp_gemm_tn(xb.p, ldx, yb.p, ldy, ob.p, ldo, M, J, N, C.byref(gb) if gb else None, stream()))
    return ob.np()


def attn_fwd(qkv_bits, ld, k_off, v_off, nb, nh, T, scale, ldp, ldo, write_p=True):
    """-> (P bits [nb*nh*T][ldp] (None with write_p=False: the inference form), O bits [nb*T][ldo])"""
    qb, ob = Buf(qkv_bits), Z((nb * T, ldo), np.uint16)
    pb = Z((nb * nh * T, ldp), np.uint16) if write_p else None
    ok(lib().lp_attn_fwd(qb.p, ld, k_off, v_off, nb, nh, T, scale, ptr(pb), ldp, ob.p, ldo, stream()))
    return (pb.np() if write_p else None), ob.np()


def attn_bwd_kv(qkv_bits, ld, v_off, do_bits, ld_do, p_bits, ldp, d_rows, nb, nh, T, scale, ld_dqkv, dk_off, dv_off):
    """-> (dS bits [nb*nh*T][ldp], dqkv bits [nb*T][ld_dqkv] with dK / dV filled in)"""
    qb, db, pb, dr = Buf(qkv_bits), Buf(do_bits), Buf(p_bits), B(d_rows, np.float32)
    ds, dq = Z((nb * nh * T, ldp), np.uint16), Z((nb * T, ld_dqkv), np.uint16)
    ok(lib().lp_attn_bwd_kv(qb.p, ld, v_off, db.p, ld_do, pb.p, ldp, dr.p, nb, nh, T, scale, ds.p, dq.p, ld_dqkv, dk_off, dv_off, stream()))
    return ds.np(), dq.np()


def attn_rowdot(a_bits, b_bits, rows, nh, ld):
    ab, bb, o = Buf(a_bits), Buf(b_bits), Z((rows, nh))
    ok(lib().lp_attn_rowdot(ab.p, bb.p, rows, nh, ld, o.p, stream()))
    return o.np()


def attn_dscores(do_bits, ld_do, v_bits, ldv, p_bits, d_rows, d_row_stride, d_b, d_h, scale, ldc, o_elems, M, N, K, batch):
    """dS = scale * P o (dO V^T - D); batch = (nb, nh, do_b, do_h, v_b, v_h, c_b, c_h). Returns the flat bf16 dS (o_elems elements)."""
    db, vb, pb, dr, ob = Buf(do_bits), Buf(v_bits), Buf(p_bits), B(d_rows, np.float32), Z(o_elems, np.uint16)
    gb = _lib.GemmBatch(*batch)
    ok(lib().lp_attn_dscores(db.p, ld_do, vb.p, ldv, pb.p, dr.p, d_row_stride, d_b, d_h, scale, ob.p, ldc, M, N, K, C.byref(gb), stream()))
    return ob.np()


def conv_wgrad(x_bits, dy_bits, g, split=0):
    xb, db, dw = Buf(x_bits), Buf(dy_bits), Z((g.Co, g.R * g.S * g.Ci))
    nws = lib().lp_conv_wgrad_workspace_bytes(C.byref(g), split)
    ws = Z(nws, np.uint8)
    ok(lib().lp_conv_wgrad(xb.p, db.p, C.byref(g), dw.p, split, ws.p, nws, stream()))
    return dw.np()


def conv_wgrad_bias(x_bits, dy_bits, g, split=0):
    """-> (dw, dbias): the weight gradient and the column sums of dy from the same launch"""
    xb, db, dw, dbias = Buf(x_bits), Buf(dy_bits), Z((g.Co, g.R * g.S * g.Ci)), Z(g.Co)
    nws = lib().lp_conv_wgrad_workspace_bytes(C.byref(g), split)
    ws = Z(nws, np.uint8)
    ok(lib().lp_conv_wgrad_bias(xb.p, db.p, C.byref(g), dw.p, dbias.p, split, ws.p, nws, stream()))
    return dw.np(), dbias.np()


def stem_fwd(x4_bits, w_bits, g):
    xb, wb, ob = Buf(x4_bits), Buf(w_bits), Z((g.B * g.Ho * g.Wo, 64), np.uint16)
    ok(lib().lp_stem_fwd(xb.p, wb.p, C.byref(g), ob.p, stream()))
    return ob.np()


def stem_wgrad(x4_bits, dy_bits, g, split=0):
    xb, db, dw = Buf(x4_bits), Buf(dy_bits), Z((64, 256))
    nws = lib().lp_conv_wgrad_workspace_bytes(C.byref(g), split)
    ws = Z(nws, np.uint8)
    ok(lib().lp_stem_wgrad(xb.p, db.p, C.byref(g), dw.p, split, ws.p, nws, stream()))
    return dw.np()


def _reduce_ws(M, Cn):
    """workspace of lp_bn_stats / lp_bn_bwd_reduce, poisoned: every partial that is read must have been written"""
    nws = lib().lp_bn_reduce_workspace_bytes(M, Cn)
    ws = Buf(np.full(nws // 4, np.nan, np.float32))
    ws.nbytes = nws
    return ws


def bn_stats(x_bits, M, Cn, raw=False):
    """lp_bn_stats -> [sum x, sum x^2] (2, Cn)"""
    xb, sums = Buf(x_bits), ZX((2, Cn))
    ws = _reduce_ws(M, Cn)
    ok(lib().lp_bn_stats(xb.p, M, Cn, sums.p, ws.p, ws.nbytes, stream()))
    return sums.np() if raw else fx(sums.np())


def bn_finalize_words(words: np.ndarray, count: float, eps=1e-5):
    """lp_bn_finalize on given lp_fxsum words (2, Cn, 2) -> (mean, invstd): what a rank computes from a SyncBatchNorm message after the exchange"""
    Cn = words.shape[1]
    sums, mean, invstd = Buf(np.ascontiguousarray(words, np.int64)), Z(Cn), Z(Cn)
    ok(lib().lp_bn_finalize(sums.p, float(count), Cn, eps, 0.1, mean.p, invstd.p, None, None, stream()))
    return mean.np(), invstd.np()


def bn_forward(x_bits, M, Cn, gamma, beta, residual_bits=None, relu=True, eps=1e-5, momentum=0.1, running=None, want_bits=False):
    xb, rb = Buf(x_bits), B(residual_bits)
    sums, mean, invstd = ZX((2, Cn)), Z(Cn), Z(Cn)
    ws = _reduce_ws(M, Cn)
    ok(lib().lp_bn_stats(xb.p, M, Cn, sums.p, ws.p, ws.nbytes, stream()))
    rm = Buf(running[0]) if running is not None else None
    rv = Buf(running[1]) if running is not None else None
    ok(lib().lp_bn_finalize(sums.p, float(M), Cn, eps, momentum, mean.p, invstd.p, ptr(rm), ptr(rv), stream()))
    y, gb, bb = Z((M, Cn), np.uint16), Buf(f32(gamma)), Buf(f32(beta))
    bits = Z(M * Cn // 8, np.uint8) if want_bits else None
    ok(lib().lp_bn_apply(xb.p, mean.p, invstd.p, gb.p, bb.p, ptr(rb), int(relu), M, Cn, y.p, ptr(bits), stream()))
    if running is not None:
        running[0][:] = rm.np()
        running[1][:] = rv.np()
    if want_bits:
        return y.np(), mean.np(), invstd.np(), bits.np()
    return y.np(), mean.np(), invstd.np()


def bn_apply_lo(x_bits, mean, invstd, gamma, beta, M, Cn, residual_bits=None, residual_lo_bits=None, zd_bits=None, dbn=None, relu=True, want_lo=True,
                seg_rows=0):
    """lp_bn_apply_seg_lo -> (y bits, y_lo bits or None, relu bits); ``dbn`` = (mean_d, invstd_d, gamma_d, beta_d) with ``zd_bits``"""
    xb, rb, rlb, zb = Buf(x_bits), B(residual_bits), B(residual_lo_bits), B(zd_bits)
    mb, vb, gb, bb = Buf(f32(mean)), Buf(f32(invstd)), Buf(f32(gamma)), Buf(f32(beta))
    d = [Buf(f32(a)) for a in dbn] if dbn is not None else [None] * 4
    y, ylo = Z((M, Cn), np.uint16), (Z((M, Cn), np.uint16) if want_lo else None)
    bits = Z(M * Cn // 8, np.uint8)
    ok(lib().lp_bn_apply_seg_lo(xb.p, mb.p, vb.p, gb.p, bb.p, ptr(rb), ptr(rlb), ptr(zb), ptr(d[0]), ptr(d[1]), ptr(d[2]), ptr(d[3]), int(relu), M, Cn,
                                seg_rows, y.p, ptr(ylo), bits.p, stream()))
    return y.np(), (ylo.np() if ylo is not None else None), bits.np()


def bn_backward(dy_bits, y_bits, x_bits, mean, invstd, gamma, M, Cn, want_dres=False, eval_mode=False, acc0=None, terms_ws=True):
    """-> (dx bits, dres bits, dgamma, dbeta): lp_bn_bwd_reduce, then lp_bn_bwd_apply (which also adds the sums into d beta / d gamma,
    starting from ``acc0`` = (dbeta0, dgamma0) if given).  eval_mode: no batch-statistics terms (sums = NULL)."""
    db, yb, xb, mb, vb, gb = Buf(dy_bits), B(y_bits), Buf(x_bits), Buf(f32(mean)), Buf(f32(invstd)), Buf(f32(gamma))
    sums = ZX((2, Cn))
    dbeta, dgamma = (Buf(f32(acc0[0])), Buf(f32(acc0[1]))) if acc0 is not None else (Z(Cn), Z(Cn))
    ws = _reduce_ws(M, Cn)
    ok(lib().lp_bn_bwd_reduce(db.p, ptr(yb), xb.p, mb.p, vb.p, M, Cn, sums.p, ws.p, ws.nbytes, stream()))
    dx = Z((M, Cn), np.uint16)
    dres = Z((M, Cn), np.uint16) if want_dres else None
    tws = Z(2 * Cn) if terms_ws else None   # (the two-launch form the engine uses; None: the self-contained kernel)
    ok(lib().lp_bn_bwd_apply(db.p, ptr(yb), xb.p, mb.p, vb.p, gb.p, None if eval_mode else sums.p, float(M), M, Cn, dx.p, ptr(dres), sums.p,
                             dbeta.p, dgamma.p, ptr(tws), stream()))
    return dx.np(), (dres.np() if dres is not None else None), dgamma.np(), dbeta.np()


def maxpool(x_bits, Bn, Hi, Wi, Cn):
    Ho, Wo = (Hi - 1) // 2 + 1, (Wi - 1) // 2 + 1
    xb, y, arg = Buf(x_bits), Z((Bn, Ho, Wo, Cn), np.uint16), Z((Bn, Ho, Wo, Cn), np.uint8)
    ok(lib().lp_maxpool_fwd(xb.p, Bn, Hi, Wi, Cn, y.p, arg.p, stream()))
    return y.np(), arg.np()


def maxpool_bwd(arg_u8, dy_bits, Bn, Hi, Wi, Cn):
    ab, db, dx = Buf(arg_u8), Buf(dy_bits), Z((Bn, Hi, Wi, Cn), np.uint16)
    ok(lib().lp_maxpool_bwd(ab.p, db.p, Bn, Hi, Wi, Cn, dx.p, stream()))
    return dx.np()


def bn_relu_maxpool(z_bits, mean, invstd, gamma, beta, Bn, Hi, Wi, Cn):
    Ho, Wo = (Hi - 1) // 2 + 1, (Wi - 1) // 2 + 1
    zb, y, arg = Buf(z_bits), Z((Bn, Ho, Wo, Cn), np.uint16), Z((Bn, Ho, Wo, Cn), np.uint8)
    mb, vb, gb, bb = Buf(f32(mean)), Buf(f32(invstd)), Buf(f32(gamma)), Buf(f32(beta))
    ok(lib().lp_bn_relu_maxpool_fwd(zb.p, mb.p, vb.p, gb.p, bb.p, Bn, Hi, Wi, Cn, y.p, arg.p, stream()))
    return y.np(), arg.np()


def bn_pool_backward(arg_u8, dy_bits, z_bits, mean, invstd, gamma, beta, Bn, Hi, Wi, Cn, raw=False):
    """-> (dz bits, dgamma, dbeta, sums) of the fused stem backward"""
    ab, db, zb = Buf(arg_u8), Buf(dy_bits), Buf(z_bits)
    mb, vb, gb, bb = Buf(f32(mean)), Buf(f32(invstd)), Buf(f32(gamma)), Buf(f32(beta))
    sums, dbeta, dgamma = ZX((2, Cn)), Z(Cn), Z(Cn)
    ok(lib().lp_bn_pool_bwd_reduce(ab.p, db.p, zb.p, mb.p, vb.p, gb.p, bb.p, Bn, Hi, Wi, Cn, sums.p, stream()))
    dz = Z((Bn * Hi * Wi, Cn), np.uint16)
    ok(lib().lp_bn_pool_bwd_apply(ab.p, db.p, zb.p, mb.p, vb.p, gb.p, bb.p, sums.p, float(Bn * Hi * Wi), Bn, Hi, Wi, Cn, dz.p, sums.p, dbeta.p,
                                  dgamma.p, stream()))
    return dz.np(), dgamma.np(), dbeta.np(), (sums.np() if raw else fx(sums.np()))


def images_to_nhwc4(img):
    img = f32(img)
    b, _, h, w = img.shape
    ib, out = Buf(img), Z((b, h, w, 4), np.uint16)
    ok(lib().lp_images_to_nhwc4(ib.p, b, h, w, out.p, stream()))
    return out.np()


def pixel_shuffle(x_bits, Bn, h, w, c_out, inverse=False, ld=None):
    ld = ld or c_out
    xb = Buf(x_bits)
    out = Z((Bn, h, w, 4 * c_out) if inverse else (Bn, 2 * h, 2 * w, ld), np.uint16)
    ok(lib().lp_pixel_shuffle(xb.p, Bn, h, w, c_out, ld, int(inverse), out.p, stream()))
    return out.np()


def adam(p, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, wd=0.0, decoupled=False):
    pb, gb, mb, vb = Buf(f32(p)), Buf(f32(g)), Buf(f32(m)), Buf(f32(v))
    wb = Z(np.shape(p), np.uint16)
    ok(lib().lp_adam_step(pb.p, gb.p, mb.p, vb.p, int(np.size(p)), lr, beta1, beta2, eps, wd, int(decoupled), step, 1.0, wb.p, stream()))
    return pb.np(), mb.np(), vb.np(), wb.np()


def permute_cba(src_bits, A, Bm, Cn):
    sb, dst = Buf(src_bits), Z((Cn, Bm, A), np.uint16)
    ok(lib().lp_permute_cba(sb.p, A, Bm, Cn, dst.p, stream()))
    return dst.np()


# ---- ViT glue (csrc/vit.hip) ------------------------------------------------------------------------------------------
def vit_patchify(img, patch):
    img = f32(img)
    b, _, h, w = img.shape
    ib, ob = Buf(img), Z((b * (h // patch) * (w // patch), 3 * patch * patch), np.uint16)
    ok(lib().lp_vit_patchify(ib.p, b, h, w, patch, ob.p, stream()))
    return ob.np()


def vit_tokens_fwd(patch_bits, cls, pos, Bn, Np, D):
    pb, cb, qb, xb = Buf(patch_bits), Buf(f32(cls)), Buf(f32(pos)), Z((Bn, Np + 1, D))
    ok(lib().lp_vit_tokens_fwd(pb.p, cb.p, qb.p, Bn, Np, D, xb.p, stream()))
    return xb.np()


def vit_tokens_bwd(dx, Bn, Np, D):
    db, pb, qb = Buf(f32(dx)), Z((Bn * Np, D), np.uint16), Z((Np + 1, D))
    ok(lib().lp_vit_tokens_bwd(db.p, Bn, Np, D, pb.p, qb.p, stream()))
    return pb.np(), qb.np()


def small_matmul(w, x, transpose_w=False, y0=None):
    w, x = f32(w), f32(x)
    r, q = w.shape
    d = x.shape[1]
    wb, xb = Buf(w), Buf(x)
    yb = Buf(f32(y0)) if y0 is not None else Z((q if transpose_w else r, d))
    ok(lib().lp_small_matmul(wb.p, xb.p, r, q, d, int(transpose_w), int(y0 is not None), yb.p, stream()))
    return yb.np()


def layernorm_fwd(x, gamma, beta, eps, delta_bits=None, drop_T=0):
    x = f32(x)
    m, d = x.shape
    rows_y = m - m // drop_T if drop_T else m
    xb, db, gb, bb = Buf(x), B(delta_bits), Buf(f32(gamma)), Buf(f32(beta))
    xo = Z((m, d)) if delta_bits is not None else None
    y, mean, rstd = Z((rows_y, d), np.uint16), Z(m), Z(m)
    ok(lib().lp_layernorm_fwd(xb.p, ptr(db), ptr(xo), gb.p, bb.p, eps, m, d, drop_T, y.p, mean.p, rstd.p, stream()))
    return y.np(), mean.np(), rstd.np(), (xo.np() if xo is not None else None)


def layernorm_bwd(dy_bits, x, mean, rstd, gamma, dx0, drop_T=0, want_bf16=False):
    x = f32(x)
    m, d = x.shape
    db, xb, mb, rb, gb = Buf(dy_bits), Buf(x), Buf(f32(mean)), Buf(f32(rstd)), Buf(f32(gamma))
    dx, dg, dbeta = Buf(f32(dx0)), Z(d), Z(d)
    if want_bf16:  # -> additionally the updated dx rounded to bf16
        d16 = Z((m, d), np.uint16)
        ok(lib().lp_layernorm_bwd_bf16(db.p, xb.p, mb.p, rb.p, gb.p, m, d, drop_T, dx.p, d16.p, dg.p, dbeta.p, stream()))
        return dx.np(), dg.np(), dbeta.np(), d16.np()
    ok(lib().lp_layernorm_bwd(db.p, xb.p, mb.p, rb.p, gb.p, m, d, drop_T, dx.p, dg.p, dbeta.p, stream()))
    return dx.np(), dg.np(), dbeta.np()


def gelu(x_bits, dy_bits=None):
    xb = Buf(x_bits)
    n = int(np.prod(xb.shape))
    if dy_bits is None:
        y = Z(xb.shape, np.uint16)
        ok(lib().lp_gelu_fwd(xb.p, n, y.p, stream()))
        return y.np()
    db, dx = Buf(dy_bits), Z(xb.shape, np.uint16)
    ok(lib().lp_gelu_bwd(xb.p, db.p, n, dx.p, stream()))
    return dx.np()


def softmax_rows(s_bits, n, scale, p_bits=None):
    """forward (p_bits None): softmax in place of a copy; backward: s_bits = dp, returns ds"""
    sb = Buf(s_bits)
    rows, ld = sb.shape
    if p_bits is None:
        ok(lib().lp_softmax_rows_fwd(sb.p, rows, n, ld, scale, stream()))
    else:
        pb = Buf(p_bits)
        ok(lib().lp_softmax_rows_bwd(pb.p, sb.p, rows, n, ld, scale, stream()))
    return sb.np()


def transpose_batched(in_bits, R, Cc, ldi, in_b, in_h, ldo, out_b, out_h, nb, nh, out_elems):
    ib, ob = Buf(in_bits), Buf(np.full(out_elems, 0x7fc0, np.uint16))  # NaN-poisoned: pads must be written
    ok(lib().lp_transpose_batched(ib.p, R, Cc, ldi, in_b, in_h, ob.p, ldo, out_b, out_h, nb, nh, stream()))
    return ob.np()


# ---- batch producers (csrc/frames.hip) ----------------------------------------------------------------------------------
def frame_norm(mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    return _lib.FrameNorm((C.c_float * 3)(*mean), (C.c_float * 3)(*std))


def frames_resize(src_u8, H, W, border=_lib.BORDER_RENORM, norm=None):
    src = np.ascontiguousarray(src_u8, dtype=np.uint8)
    S, Hs, Ws, _ = src.shape
    sb = Buf(src)
    out = Z((S, 3, H, W) if norm is not None else (S, H, W, 3))
    ok(lib().lp_frames_resize(sb.p, S, Hs, Ws, Hs * Ws * 3, Ws * 3, H, W, border, C.byref(norm) if norm is not None else None, out.p, stream()))
    return out.np()


def frames_resize_cubic(src_u8, H, W, round_u8=1, norm=None):
    src = np.ascontiguousarray(src_u8, dtype=np.uint8)
    S, Hs, Ws, _ = src.shape
    sb = Buf(src)
    out = Z((S, 3, H, W) if norm is not None else (S, H, W, 3))
    ok(lib().lp_frames_resize_cubic(sb.p, S, Hs, Ws, Hs * Ws * 3, Ws * 3, H, W, round_u8, C.byref(norm) if norm is not None else None, out.p,
                                    stream()))
    return out.np()


def frames_augment(src_hwc, matrix=None, brightness=1.0, contrast=1.0, contrast_center=0.5, shot_factor=0.0, seed=0, norm=None):
    src = f32(src_hwc)
    S, H, W, _ = src.shape
    aug = _lib.FrameAugment(int(matrix is not None), (C.c_float * 6)(*(np.asarray(matrix, np.float32).reshape(-1) if matrix is not None else [0] * 6)),
                            brightness, contrast, contrast_center, shot_factor, seed)
    norm = norm if norm is not None else frame_norm()
    sb, out = Buf(src), Z((S, 3, H, W))
    ok(lib().lp_frames_augment(sb.p, S, H, W, C.byref(aug), C.byref(norm), out.p, stream()))
    return out.np()


def labeled_keypoints(kp, src_hw, H, W, affine=None, hflip=None, swap=None, vis=None, uniform=False):
    kp = f32(kp)
    b, k, _ = kp.shape
    kb, hb = Buf(kp), Buf(f32(src_hw))
    ab, fb, sb, vb = B(affine, np.float32), B(hflip, np.int32), B(swap, np.int32), B(vis, np.int32)
    out, vout = Z((b, k, 2)), Z((b, k), np.int32)
    ok(lib().lp_labeled_keypoints(kb.p, hb.p, ptr(ab), ptr(fb), ptr(sb), ptr(vb), int(uniform), b, k, H, W, out.p, vout.p, stream()))
    return out.np(), vout.np()


def temporal_heatmap(kind, pred, conf, eps, thr, gout=1.0):
    pred, conf = f32(pred), f32(conf)
    s, k, h, w = pred.shape
    eps = np.broadcast_to(f32(eps).reshape(-1), (k,)) if np.size(eps) == 1 else f32(eps)
    pb, cb, eb = Buf(pred), Buf(conf), Buf(f32(eps))
    ws = Z((lib().lp_temporal_heatmap_workspace_bytes(s, k),), np.uint8)
    loss, grad, go = Z((1,)), Z(pred.shape), Buf(f32([gout]))
    ok(lib().lp_temporal_heatmap_fwd(kind, pb.p, cb.p, s, k, h, w, eb.p, thr, loss.p, ws.p, stream()))
    ok(lib().lp_temporal_heatmap_bwd(kind, pb.p, s, k, h, w, ws.p, go.p, grad.p, 0, stream()))
    return float(loss.np()[0]), grad.np()


# ---- inference: folded BatchNorm + conv with residual / ReLU in the store pass -------------------------------------------------
def bn_fold(w, gamma, beta, rmean, rvar, eps=1e-5):
    w = f32(w)
    Co, per = w.shape[0], int(np.prod(w.shape[1:]))
    wb, gb, bb, mb, vb = Buf(w), Buf(f32(gamma)), Buf(f32(beta)), Buf(f32(rmean)), Buf(f32(rvar))
    wo, bo = Z((Co, per), np.uint16), Z((Co,))
    ok(lib().lp_bn_fold(wb.p, gb.p, bb.p, mb.p, vb.p, eps, Co, per, wo.p, bo.p, stream()))
    return wo.np(), bo.np()


def conv_fwd_act(x_nhwc_bits, w_bits, g, bias=None, residual_bits=None, relu=False):
    M = g.B * g.Ho * g.Wo
    xb, wb, bb, rb = Buf(x_nhwc_bits), Buf(w_bits), B(bias, np.float32), B(residual_bits)
    ob = Z((M, g.Co), np.uint16)
    ok(lib().lp_conv_fwd_act(xb.p, wb.p, C.byref(g), ptr(bb), ptr(rb), int(relu), ob.p, stream()))
    return ob.np()
