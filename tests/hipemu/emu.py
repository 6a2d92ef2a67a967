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


def lib() -> C.CDLL:
    global _emu
    if _emu is None:
        subprocess.run(["bash", BUILD, "emu"], check=True, capture_output=True)
        _emu = _lib.declare(C.CDLL(os.path.join(HERE, "liblp_emu.so")))
    return _emu


def ptr(a: np.ndarray | None):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def i32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


class Tables:
    """Keeps the numpy tables alive next to the ctypes struct."""

    def __init__(self, h: int, w: int, ds: int):
        ty, tx = _tables.axis_tables(h, ds), _tables.axis_tables(w, ds)
        self.keep = [ty["row_base"], ty["row_taps"], tx["col_start"], tx["col_taps"], tx["colT_start"], tx["colT_taps"]]
        self.keep = [np.ascontiguousarray(a) for a in self.keep]
        self.struct = _lib.DecodeTables(*[ptr(a) for a in self.keep], ty["ty"], tx["tx"], tx["tc"])


def frame_map(transforms=None, tf_mode=_lib.TF_NONE, bbox=None, views=1, K=1, model_h=1.0, model_w=1.0):
    keep = [f32(transforms) if transforms is not None else None, f32(bbox) if bbox is not None else None]
    fm = _lib.FrameMap(ptr(keep[0]), tf_mode, ptr(keep[1]), 4 * views, max(1, K // views), float(model_h), float(model_w))
    return fm, keep


def decode_fwd(heat, ds, temperature=1000.0, fm=None):
    heat = f32(heat)
    b, k, h, w = heat.shape
    tb = Tables(h, w, ds)
    keep = None
    if fm is None:
        fm, keep = frame_map(K=k)
    kp_aug, kp_frame = np.zeros((b, k, 2), np.float32), np.zeros((b, k, 2), np.float32)
    conf, stats = np.zeros((b, k), np.float32), np.zeros((b, k, 4), np.float32)
    rc = lib().lp_decode_fwd(ptr(heat), b, k, h, w, ds, temperature, C.byref(tb.struct), C.byref(fm), ptr(kp_aug),
                             ptr(kp_frame), ptr(conf), ptr(stats), None)
    assert rc == 0, rc
    return kp_aug, kp_frame, conf, stats


def decode_bwd(heat, ds, stats, g_aug=None, g_frame=None, temperature=1000.0, fm=None):
    heat = f32(heat)
    b, k, h, w = heat.shape
    tb = Tables(h, w, ds)
    keep = None
    if fm is None:
        fm, keep = frame_map(K=k)
    g_heat = np.zeros_like(heat)
    ga = f32(g_aug) if g_aug is not None else None
    gf = f32(g_frame) if g_frame is not None else None
    rc = lib().lp_decode_bwd(ptr(heat), b, k, h, w, ds, temperature, C.byref(tb.struct), C.byref(fm), ptr(f32(stats)),
                             ptr(ga), ptr(gf), ptr(g_heat), 0, None)
    assert rc == 0, rc
    return g_heat


def heatmap_gen(kp, vis, img_h, img_w, h, w, sigma=1.25):
    kp = f32(kp)
    b, k, _ = kp.shape
    out = np.zeros((b, k, h, w), np.float32)
    v = i32(vis) if vis is not None else None
    rc = lib().lp_heatmap_gen(ptr(kp), ptr(v), b, k, img_h, img_w, h, w, sigma, ptr(out), None)
    assert rc == 0, rc
    return out


def heatmap_mse(targ, pred, gout=1.0):
    targ, pred = f32(targ), f32(pred)
    b, k, h, w = pred.shape
    ws = np.zeros(lib().lp_heatmap_mse_workspace_bytes(b, k), np.uint8)
    loss = np.zeros(1, np.float32)
    assert lib().lp_heatmap_mse_fwd(ptr(targ), ptr(pred), b, k, h, w, ptr(loss), ptr(ws), None) == 0
    g = np.zeros_like(pred)
    go = f32([gout])
    assert lib().lp_heatmap_mse_bwd(ptr(targ), ptr(pred), b, k, h, w, ptr(ws), ptr(go), ptr(g), 0, None) == 0
    return loss[0], g


def unimodal_mse(kp_aug, pred, conf, img_h, img_w, thr, sigma=1.25, gout=1.0):
    kp_aug, pred, conf = f32(kp_aug), f32(pred), f32(conf)
    s, k, h, w = pred.shape
    ws = np.zeros(lib().lp_heatmap_mse_workspace_bytes(s, k), np.uint8)
    loss = np.zeros(1, np.float32)
    assert lib().lp_unimodal_mse_fwd(ptr(kp_aug), ptr(pred), ptr(conf), s, k, img_h, img_w, h, w, sigma, thr, ptr(loss),
                                     ptr(ws), None) == 0
    g = np.zeros_like(pred)
    go = f32([gout])
    assert lib().lp_unimodal_mse_bwd(ptr(kp_aug), ptr(pred), s, k, img_h, img_w, h, w, sigma, ptr(ws), ptr(go), ptr(g), 0,
                                     None) == 0
    return loss[0], g


def softmax2d(logits_nhwc, K):
    """logits (B, n, C) channel-padded -> prob (B, K, n)"""
    x = f32(logits_nhwc)
    b, n, c = x.shape
    out = np.zeros((b, K, n), np.float32)
    assert lib().lp_softmax2d_fwd(ptr(x), n * c, c, 1, b, K, n, ptr(out), None) == 0
    return out


def softmax2d_bwd(prob, gprob, C):
    prob, gprob = f32(prob), f32(gprob)
    b, k, n = prob.shape
    gin = np.zeros((b, n, C), np.uint16)
    assert lib().lp_softmax2d_bwd(ptr(prob), ptr(gprob), b, k, n, ptr(gin), n * C, C, 1, None) == 0
    return from_bf16_bits(gin).numpy()


def temporal(kp, conf, eps, thr):
    kp = f32(kp)
    s, k, _ = kp.shape
    c = f32(conf) if conf is not None else None
    e = f32(np.broadcast_to(np.asarray(eps, np.float32), (k,)))
    loss, g = np.zeros(1, np.float32), np.zeros_like(kp)
    assert lib().lp_temporal_fwd_bwd(ptr(kp), ptr(c), s, k, ptr(e), thr, ptr(loss), ptr(g), None) == 0
    return loss[0], g


def pca(kp, index, mean, kept, eps):
    kp = f32(kp)
    s, k, _ = kp.shape
    idx = i32(index)
    rows, pts = idx.shape
    mean, kept = f32(mean), f32(kept)
    loss, g = np.zeros(1, np.float32), np.zeros_like(kp)
    assert lib().lp_pca_fwd_bwd(ptr(kp), s, k, ptr(idx), rows, pts, ptr(mean), ptr(kept), kept.shape[0], float(eps), ptr(loss),
                                ptr(g), None) == 0
    return loss[0], g


def rmse(targ, pred):
    targ, pred = f32(targ), f32(pred)
    loss = np.zeros(1, np.float32)
    assert lib().lp_rmse_fwd(ptr(targ), ptr(pred), targ.size // 2, ptr(loss), None) == 0
    return loss[0]


# ---- bf16 helpers / conv wrappers -------------------------------------------------------------------
import torch  # noqa: E402


def to_bf16_bits(t: "torch.Tensor") -> np.ndarray:
    return np.ascontiguousarray(t.contiguous().to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16))


def from_bf16_bits(a: np.ndarray) -> "torch.Tensor":
    return torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16).float()


def geom(B, Hi, Wi, Ci, Co, R, S, stride, pad, Ho=None, Wo=None):
    if Ho is None:
        Ho = (Hi + 2 * pad - R) // stride + 1
        Wo = (Wi + 2 * pad - S) // stride + 1
    return _lib.ConvGeom(B, Hi, Wi, Ci, Ho, Wo, Co, R, S, stride, pad)


def conv_fwd(x_nhwc_bits, w_bits, g, bias=None, f32_out=False, ldo=None, n_store=0):
    M = g.B * g.Ho * g.Wo
    ldo = ldo or g.Co
    ob = np.zeros((M, ldo), np.uint16)
    of = np.zeros((M, ldo), np.float32) if f32_out else None
    b = f32(bias) if bias is not None else None
    rc = lib().lp_conv_fwd(ptr(x_nhwc_bits), ptr(w_bits), C.byref(g), ptr(b), ptr(ob), ptr(of), ldo, n_store, None)
    assert rc == 0, rc
    return ob, of


def conv_dgrad(dy_bits, wd_bits, g, addend_bits=None, bias=None, f32_out=False, ldo=None, n_store=0):
    M = g.B * g.Hi * g.Wi
    ldo = ldo or g.Ci
    ob = np.zeros((M, ldo), np.uint16)
    of = np.zeros((M, ldo), np.float32) if f32_out else None
    b = f32(bias) if bias is not None else None
    rc = lib().lp_conv_dgrad(ptr(dy_bits), ptr(wd_bits), C.byref(g), ptr(b), ptr(addend_bits), ptr(ob), ptr(of), ldo, n_store, None)
    assert rc == 0, rc
    return ob, of


def conv_wgrad(x_bits, dy_bits, g, split=0):
    dw = np.zeros((g.Co, g.R * g.S * g.Ci), np.float32)
    rc = lib().lp_conv_wgrad(ptr(x_bits), ptr(dy_bits), C.byref(g), ptr(dw), split, None)
    assert rc == 0, rc
    return dw


def stem_fwd(x4_bits, w_bits, g):
    ob = np.zeros((g.B * g.Ho * g.Wo, 64), np.uint16)
    assert lib().lp_stem_fwd(ptr(x4_bits), ptr(w_bits), C.byref(g), ptr(ob), None) == 0
    return ob


def stem_wgrad(x4_bits, dy_bits, g, split=0):
    dw = np.zeros((64, 256), np.float32)
    assert lib().lp_stem_wgrad(ptr(x4_bits), ptr(dy_bits), C.byref(g), ptr(dw), split, None) == 0
    return dw


def bn_forward(x_bits, M, C, gamma, beta, residual_bits=None, relu=True, eps=1e-5, momentum=0.1, running=None):
    sums = np.zeros((2, C), np.float32)
    assert lib().lp_bn_stats(ptr(x_bits), M, C, ptr(sums), None) == 0
    mean, invstd = np.zeros(C, np.float32), np.zeros(C, np.float32)
    rm, rv = (running if running is not None else (None, None))
    assert lib().lp_bn_finalize(ptr(sums), float(M), C, eps, momentum, ptr(mean), ptr(invstd), ptr(rm), ptr(rv), None) == 0
    y = np.zeros((M, C), np.uint16)
    g, b = f32(gamma), f32(beta)
    assert lib().lp_bn_apply(ptr(x_bits), ptr(mean), ptr(invstd), ptr(g), ptr(b), ptr(residual_bits), int(relu), M, C, ptr(y), None) == 0
    return y, mean, invstd


def bn_backward(dy_bits, y_bits, x_bits, mean, invstd, gamma, M, C, want_dres=False):
    sums = np.zeros((2, C), np.float32)
    dbeta, dgamma = np.zeros(C, np.float32), np.zeros(C, np.float32)
    assert lib().lp_bn_bwd_reduce(ptr(dy_bits), ptr(y_bits), ptr(x_bits), ptr(mean), ptr(invstd), M, C, ptr(sums), ptr(dbeta),
                                  ptr(dgamma), None) == 0
    dx = np.zeros((M, C), np.uint16)
    dres = np.zeros((M, C), np.uint16) if want_dres else None
    g = f32(gamma)
    assert lib().lp_bn_bwd_apply(ptr(dy_bits), ptr(y_bits), ptr(x_bits), ptr(mean), ptr(invstd), ptr(g), ptr(sums), float(M), M, C,
                                 ptr(dx), ptr(dres), None) == 0
    return dx, dres, dgamma, dbeta


def maxpool(x_bits, B, Hi, Wi, C):
    Ho, Wo = (Hi - 1) // 2 + 1, (Wi - 1) // 2 + 1
    y = np.zeros((B, Ho, Wo, C), np.uint16)
    assert lib().lp_maxpool_fwd(ptr(x_bits), B, Hi, Wi, C, ptr(y), None) == 0
    return y


def maxpool_bwd(x_bits, dy_bits, B, Hi, Wi, C):
    dx = np.zeros((B, Hi, Wi, C), np.uint16)
    assert lib().lp_maxpool_bwd(ptr(x_bits), ptr(dy_bits), B, Hi, Wi, C, ptr(dx), None) == 0
    return dx


def images_to_nhwc4(img):
    img = f32(img)
    b, _, h, w = img.shape
    out = np.zeros((b, h, w, 4), np.uint16)
    assert lib().lp_images_to_nhwc4(ptr(img), b, h, w, ptr(out), None) == 0
    return out


def pixel_shuffle(x_bits, B, h, w, c_out, inverse=False):
    out = np.zeros((B, h, w, 4 * c_out) if inverse else (B, 2 * h, 2 * w, c_out), np.uint16)
    assert lib().lp_pixel_shuffle(ptr(x_bits), B, h, w, c_out, int(inverse), ptr(out), None) == 0
    return out


def adam(p, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, wd=0.0, decoupled=False):
    p, g, m, v = f32(p).copy(), f32(g), f32(m).copy(), f32(v).copy()
    pb = np.zeros(p.shape, np.uint16)
    assert lib().lp_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), p.size, lr, beta1, beta2, eps, wd, int(decoupled), step, 1.0, ptr(pb),
                              None) == 0
    return p, m, v, pb


def permute_cba(src_bits, A, B, Cn):
    dst = np.zeros((Cn, B, A), np.uint16)
    assert lib().lp_permute_cba(ptr(src_bits), A, B, Cn, ptr(dst), None) == 0
    return dst
