"""Torch-tensor front end of the lp_hip C ABI: device checks, stream plumbing, autograd glue.

Every function here enqueues hand-written HIP kernels from ``liblp_hip.so`` on torch's current stream.  PyTorch is
used for device memory, streams and autograd bookkeeping only.  There is no CPU path: tensors must live on a ROCm
device, otherwise ``LpHipUnavailable`` is raised.
"""

from __future__ import annotations

import ctypes as C
import os
import functools

import numpy as np
import torch

from . import _lib, _tables
from ._lib import LpHipUnavailable, check

__all__ = [
    "decode", "DecodeFrameMap", "generate_heatmaps", "heatmap_mse", "unimodal_mse", "temporal_loss", "pca_loss",
    "rmse", "require_device", "frames_resize", "frames_augment", "labeled_keypoints",
]


def require_device(*tensors: torch.Tensor) -> torch.device:
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise LpHipUnavailable(
                "lightning_pose_amd ops run only on a ROCm device (got a CPU tensor); there is no CPU fallback")
        if t.device.index is not None and t.device.index != torch.cuda.current_device():
            raise LpHipUnavailable(f"tensor on {t.device} but the current device is cuda:{torch.cuda.current_device()}: kernels launch on the "
                                   "current device's stream - call torch.cuda.set_device (one process per GPU)")
        dev = t.device
    return dev


def require_device_type(device: torch.device) -> None:
    if device.type != "cuda":
        raise LpHipUnavailable(f"lightning_pose_amd needs a ROCm device (got {device}); there is no CPU fallback")
    if device.index is not None and device.index != torch.cuda.current_device():
        torch.cuda.set_device(device)  # the engine's device becomes the process's current device (one process per GPU)


def _stream() -> C.c_void_p:
    """torch's current stream on the CURRENT device.  Kernels are launched on the current device, so a model's device must be the
    current one: Engine / ViTEngine make it so when they are built (one process per GPU, as the reference's DDP), and require_device
    refuses tensors of another device instead of launching device-0 kernels on device-1 pointers."""
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: torch.Tensor | None) -> C.c_void_p | None:
    return None if t is None else C.c_void_p(t.data_ptr())


def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous()


# --------------------------------------------------------------------------------------------------------
# decode
# --------------------------------------------------------------------------------------------------------

@functools.lru_cache(maxsize=16)
def _device_tables(h: int, w: int, ds: int, dev: torch.device):
    ty, tx = _tables.axis_tables(h, ds), _tables.axis_tables(w, ds)
    keep = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in
            (ty["row_base"], ty["row_taps"], tx["col_start"], tx["col_taps"], tx["colT_start"], tx["colT_taps"])]
    struct = _lib.DecodeTables(*[t.data_ptr() for t in keep], ty["ty"], tx["tx"], tx["tc"])
    return struct, keep


class DecodeFrameMap:
    """undo-affine + model->frame epilogue parameters (reference: data/utils.py:191-234, data/bboxes.py:222-288)."""

    def __init__(self, transforms: torch.Tensor | None, is_multiview: bool, bbox: torch.Tensor | None, num_views: int,
                 model_h: int, model_w: int, num_keypoints: int):
        self.keep = []
        self.rows_needed: int | None = None
        self.bbox_rows: int | None = None
        self._tf = self._bbox = None
        tf_mode, tf = _lib.TF_NONE, None
        if transforms is not None and transforms.shape[-1] == 3:
            tf = _f32c(transforms)
            if is_multiview:
                tf_mode = _lib.TF_PER_VIEW
            elif tf.dim() == 2 or tf.shape[0] == 1:
                # one matrix for the whole batch: (2, 3), or (1, 2, 3) as the DALI pipeline hands it over - the reference replicates a
                # single inverse over the batch (data/utils.py:176-180)
                tf_mode = _lib.TF_SINGLE
            else:
                tf_mode = _lib.TF_PER_FRAME
                self.rows_needed = int(tf.shape[0])   # one matrix per frame: checked against the batch where it is known (check_batch)
            self._tf = tf
            self.keep.append(tf)
        bb = None
        if bbox is not None:
            bb = _f32c(bbox)
            self.bbox_rows = int(bb.shape[0]) if bb.dim() == 2 else None
            self._bbox = bb
            self.keep.append(bb)
        self.struct = _lib.FrameMap(_p(tf), tf_mode, _p(bb), 4 * num_views, max(1, num_keypoints // num_views),
                                    float(model_h), float(model_w))

    def check_batch(self, b: int, device: torch.device | None = None) -> None:
        """Called where the batch (and its device) is known.  The kernels index the per-frame tables by frame: a table shorter than the batch
        would be read out of bounds (the reference fails in torch.bmm / broadcasting on the same inputs); and they dereference raw pointers:
        tables the caller left on another device (the reference moves its inverse matrices to the keypoints' device, data/utils.py:170-172)
        are moved here."""
        if self.rows_needed is not None and self.rows_needed != b:
            raise ValueError(f"{self.rows_needed} affine transforms for a batch of {b} frames (one per frame, or a single (2, 3) / (1, 2, 3))")
        if self.bbox_rows is not None and self.bbox_rows != b:
            if self.bbox_rows != 1:
                raise ValueError(f"{self.bbox_rows} bounding boxes for a batch of {b} frames")
            self._bbox = self._bbox.expand(b, -1).contiguous()   # one box for every frame: what broadcasting gives the reference (data/bboxes.py:222-288)
            self.keep.append(self._bbox)
            self.struct.bbox = self._bbox.data_ptr()
            self.bbox_rows = b
        if device is not None:
            if self._tf is not None and self._tf.device != device:
                self._tf = self._tf.to(device)
                self.keep.append(self._tf)
                self.struct.transforms = self._tf.data_ptr()
            if self._bbox is not None and self._bbox.device != device:
                self._bbox = self._bbox.to(device)
                self.keep.append(self._bbox)
                self.struct.bbox = self._bbox.data_ptr()


def _prune_env() -> str:
    """LP_DECODE_PRUNE: "auto" (default / unset / empty), "0" or "1" - anything else is a configuration error, not a silent default"""
    v = os.environ.get("LP_DECODE_PRUNE", "").strip().lower()
    if v in ("", "auto"):
        return "auto"
    if v in ("0", "1"):
        return v
    raise ValueError(f"LP_DECODE_PRUNE must be auto, 0 or 1 (got {v!r})")


class _DecodePruneAuto:
    """Chooses between the plain and the exactly-pruned decode kernels (the `prune` argument of lp_decode_fwd / lp_decode_bwd) from what the
    decode itself reports.

    stats[..., 1] of lp_decode_fwd is sum exp(T (y - max y)) over the up-sampled map = the number of pixels that carry weight in the
    soft-argmax: ~2 - 10 on the peaked maps of a trained head (T = 1000), ~all 147 456 on the flat maps of an untrained one.  Pruning pays
    (forward 1.4x, backward 1.8x, profiles/archive/r02k_decode_microbench.jsonl) when most maps are peaked and costs when they are flat, and which
    regime a run is in changes once, early in training.  So every PERIOD-th call (and the FIRST-th) the fraction of peaked maps is reduced
    on the device and copied to pinned host memory WITHOUT a synchronisation; a later call picks the value up once its event has
    completed and flips the choice if needed.  LP_DECODE_PRUNE=0 / 1 pins the choice instead (read per call, here on the host: the
    library itself holds no switch since round 5).

    One instance per OWNER (a tracker: HeatmapTracker._decode passes its own; stand-alone ops.decode calls share `_decode_prune_default`),
    so one model's maps never steer another's kernels and a fresh model starts from the plain kernels.  While the current stream is being
    captured into a HIP graph nothing here runs - no event query, no pinned allocation, no copy, no counter: a replayed graph keeps the
    kernels it was captured with, and the eager steps around captures are where the choice is re-evaluated."""

    PERIOD, FIRST, PEAKED_FRACTION_OF_PIXELS, PEAKED_MAPS = 32, 2, 0.01, 0.5

    def __init__(self) -> None:
        self.calls, self.pending, self.want = 0, None, 0   # want: this owner's current choice (0 plain, 1 pruned)
        self.last = -2                                      # what the most recent decode call was given (-2: none yet)

    @property
    def state(self) -> int:
        """the `prune` argument of this owner's most recent decode call (tests, bench): 0 plain, 1 pruned, -2 no call yet"""
        return self.last

    @staticmethod
    def _capturing() -> bool:
        return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()

    def before(self) -> int:
        """the `prune` argument for the call about to be made"""
        env = _prune_env()
        if env != "auto":          # pinned by the environment
            self.pending = None
            self.last = int(env)
            return self.last
        if not self._capturing() and self.pending is not None:
            host, ev = self.pending
            if ev is None or ev.query():
                self.want = 1 if float(host[0]) >= self.PEAKED_MAPS else 0
                self.pending = None
        self.last = self.want
        return self.last

    def after(self, stats: torch.Tensor, n_up: int) -> None:
        if _prune_env() != "auto" or self._capturing():
            return
        self.calls += 1
        if self.pending is not None or not (self.calls == self.FIRST or self.calls % self.PERIOD == 0):
            return
        frac = (stats[..., 1] < self.PEAKED_FRACTION_OF_PIXELS * n_up).float().mean().reshape(1)
        if stats.device.type == "cuda":
            host = torch.empty(1, dtype=torch.float32, pin_memory=True)
            host.copy_(frac, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self.pending = (host, ev)
        else:
            self.pending = (frac.detach().clone(), None)


_decode_prune_default = _DecodePruneAuto()
_decode_prune_auto = _decode_prune_default   # (the name round 3's tests and bench read)


class _DecodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, heat, ds, temperature, frame_map, prune):
        require_device(heat)
        ctx.in_dtype = heat.dtype
        heat = heat.to(torch.float32).contiguous()
        b, k, h, w = heat.shape
        frame_map.check_batch(b, heat.device)
        tables, keep = _device_tables(h, w, ds, heat.device)
        kp_aug = torch.empty(b, k, 2, device=heat.device, dtype=torch.float32)
        kp_frame = torch.empty_like(kp_aug)
        conf = torch.empty(b, k, device=heat.device, dtype=torch.float32)
        stats = torch.empty(b, k, 4, device=heat.device, dtype=torch.float32)
        mode = prune.before()
        check(_lib.lib().lp_decode_fwd(_p(heat), b, k, h, w, ds, float(temperature), C.byref(tables),
                                       C.byref(frame_map.struct), _p(kp_aug), _p(kp_frame), _p(conf), _p(stats), mode, _stream()),
              "lp_decode_fwd")
        prune.after(stats, h * w * (4 ** int(ds)))
        ctx.save_for_backward(heat, stats)
        ctx.args = (ds, float(temperature), frame_map, tables, keep, mode)
        ctx.mark_non_differentiable(conf)
        return kp_aug.reshape(b, 2 * k), kp_frame.reshape(b, 2 * k), conf

    @staticmethod
    def backward(ctx, g_aug, g_frame, _g_conf):
        heat, stats = ctx.saved_tensors
        ds, temperature, frame_map, tables, _keep, mode = ctx.args   # (the backward runs the kernel family its forward ran)
        b, k, h, w = heat.shape
        ga = _f32c(g_aug) if g_aug is not None else None
        gf = _f32c(g_frame) if g_frame is not None else None
        g_heat = torch.empty_like(heat)
        check(_lib.lib().lp_decode_bwd(_p(heat), b, k, h, w, ds, temperature, C.byref(tables), C.byref(frame_map.struct),
                                       _p(stats), _p(ga), _p(gf), _p(g_heat), 0, mode, _stream()), "lp_decode_bwd")
        return g_heat.to(ctx.in_dtype), None, None, None, None


class _FrameMapFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kp, frame_map):
        require_device(kp)
        x = _f32c(kp)
        b, k = x.shape[0], x.shape[1] // 2
        frame_map.check_batch(b, x.device)
        out = torch.empty_like(x)
        check(_lib.lib().lp_frame_map_apply(_p(x), b, k, C.byref(frame_map.struct), 0, _p(out), _stream()), "lp_frame_map_apply")
        ctx.frame_map = frame_map
        return out

    @staticmethod
    def backward(ctx, g):
        g = _f32c(g)
        b, k = g.shape[0], g.shape[1] // 2
        out = torch.empty_like(g)
        check(_lib.lib().lp_frame_map_apply(_p(g), b, k, C.byref(ctx.frame_map.struct), 1, _p(out), _stream()), "lp_frame_map_apply")
        return out, None


def frame_map_apply(keypoints: torch.Tensor, frame_map: DecodeFrameMap) -> torch.Tensor:
    """keypoints (B, 2K) in model px -> frame px through the undo-affine / bbox map (differentiable)."""
    return _FrameMapFn.apply(keypoints, frame_map)


def decode(heatmaps: torch.Tensor, downsample_factor: int, temperature: float, frame_map: DecodeFrameMap,
           prune: "_DecodePruneAuto | None" = None):
    """heatmaps (B,K,h,w) -> keypoints in model px (B,2K), keypoints in frame px (B,2K), confidences (B,K).
    ``prune``: the caller's own plain / pruned-kernel chooser (a tracker passes its own, see _DecodePruneAuto); default: a shared one."""
    return _DecodeFn.apply(heatmaps, int(downsample_factor), float(temperature), frame_map, prune or _decode_prune_default)


# --------------------------------------------------------------------------------------------------------
# heat-map targets / losses
# --------------------------------------------------------------------------------------------------------

class _HeatmapGenFn(torch.autograd.Function):
    """generate_heatmaps with the keypoints attached (keep_gradients=True): lp_heatmap_gen forward, lp_heatmap_gen_bwd backward"""

    @staticmethod
    def forward(ctx, keypoints, height, width, output_shape, sigma, visibility):
        out = generate_heatmaps(keypoints.detach(), height, width, output_shape, sigma, visibility)
        ctx.save_for_backward(_f32c(keypoints))
        ctx.args = (int(height), int(width), tuple(output_shape), float(sigma),
                    None if visibility is None else visibility.to(device=keypoints.device, dtype=torch.int32).contiguous())
        ctx.in_dtype = keypoints.dtype
        return out

    @staticmethod
    def backward(ctx, g):
        (kp,) = ctx.saved_tensors
        height, width, (h, w), sigma, vis = ctx.args
        b, k, _ = kp.shape
        gkp = torch.empty_like(kp)
        check(_lib.lib().lp_heatmap_gen_bwd(_p(kp), _p(vis), b, k, height, width, h, w, sigma, _p(_f32c(g)), _p(gkp), _stream()),
              "lp_heatmap_gen_bwd")
        return gkp.to(ctx.in_dtype), None, None, None, None, None


def generate_heatmaps_with_grad(keypoints: torch.Tensor, height: int, width: int, output_shape: tuple[int, int], sigma: float = 1.25,
                                visibility: torch.Tensor | None = None) -> torch.Tensor:
    return _HeatmapGenFn.apply(keypoints, height, width, tuple(output_shape), sigma, visibility)


def generate_heatmaps(keypoints: torch.Tensor, height: int, width: int, output_shape: tuple[int, int], sigma: float = 1.25,
                      visibility: torch.Tensor | None = None) -> torch.Tensor:
    require_device(keypoints)
    kp = _f32c(keypoints)
    if kp.dim() != 3 or kp.shape[2] != 2:
        raise ValueError(f"keypoints must be (B, K, 2), got {tuple(kp.shape)}")
    b, k, _ = kp.shape
    h, w = output_shape
    vis = None
    if visibility is not None:
        if tuple(visibility.shape) != (b, k):
            raise ValueError(f"visibility must be {(b, k)}, got {tuple(visibility.shape)}")
        vis = visibility.to(device=kp.device, dtype=torch.int32).contiguous()
    out = torch.empty(b, k, h, w, device=kp.device, dtype=torch.float32)
    check(_lib.lib().lp_heatmap_gen(_p(kp), _p(vis), b, k, int(height), int(width), h, w, float(sigma), _p(out), _stream()),
          "lp_heatmap_gen")
    return out


def heatmap_confidence(heatmaps: torch.Tensor, locs: torch.Tensor, radius: int) -> torch.Tensor:
    require_device(heatmaps, locs)
    heat, lc = _f32c(heatmaps), _f32c(locs)
    b, k, h, w = heat.shape
    if lc.shape != (b, k, 2):
        raise ValueError(f"locs must be {(b, k, 2)}, got {tuple(lc.shape)}")
    out = torch.empty(b, k, device=heat.device, dtype=torch.float32)
    check(_lib.lib().lp_heatmap_confidence(_p(heat), _p(lc), b, k, h, w, int(radius), _p(out), _stream()), "lp_heatmap_confidence")
    return out


class _HeatmapLossFn(torch.autograd.Function):
    """masked mean over the labelled maps of a per-map sum: kind = HM_MSE / HM_KL / HM_JS"""

    @staticmethod
    def forward(ctx, targ, pred, kind):
        require_device(targ, pred)
        ctx.in_dtype = pred.dtype
        targ, pred = _f32c(targ), _f32c(pred)
        if pred.dim() != 4 or targ.shape != pred.shape:   # raw pointers from here on: a shape the reference's mse_loss would refuse
            raise ValueError(f"heat-map targets {tuple(targ.shape)} and predictions {tuple(pred.shape)} must both be (B, K, h, w)")
        b, k, h, w = pred.shape
        ws = torch.empty(_lib.lib().lp_heatmap_mse_workspace_bytes(b, k), device=pred.device, dtype=torch.uint8)
        loss = torch.empty(1, device=pred.device, dtype=torch.float32)
        check(_lib.lib().lp_heatmap_loss_fwd(kind, _p(targ), _p(pred), b, k, h, w, _p(loss), _p(ws), _stream()), "lp_heatmap_loss_fwd")
        ctx.save_for_backward(targ, pred, ws)
        ctx.kind = kind
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        targ, pred, ws = ctx.saved_tensors
        b, k, h, w = pred.shape
        g = torch.empty_like(pred)
        go = _f32c(gout).reshape(1)
        check(_lib.lib().lp_heatmap_loss_bwd(ctx.kind, _p(targ), _p(pred), b, k, h, w, _p(ws), _p(go), _p(g), 0, _stream()),
              "lp_heatmap_loss_bwd")
        return None, g.to(ctx.in_dtype), None


def heatmap_mse(targets: torch.Tensor, predictions: torch.Tensor) -> torch.Tensor:
    return _HeatmapLossFn.apply(targets, predictions, _lib.HM_MSE)


def heatmap_kl(targets: torch.Tensor, predictions: torch.Tensor) -> torch.Tensor:
    return _HeatmapLossFn.apply(targets, predictions, _lib.HM_KL)


def heatmap_js(targets: torch.Tensor, predictions: torch.Tensor) -> torch.Tensor:
    return _HeatmapLossFn.apply(targets, predictions, _lib.HM_JS)


class _UnimodalFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kp_aug, pred, conf, img_h, img_w, sigma, thr):
        require_device(kp_aug, pred, conf)
        ctx.in_dtype = pred.dtype
        kp, pred, conf = _f32c(kp_aug), _f32c(pred), _f32c(conf)
        s, k, h, w = pred.shape
        if kp.numel() != s * k * 2 or tuple(conf.shape) != (s, k):
            raise ValueError(f"unimodal_mse: keypoints {tuple(kp.shape)} / confidences {tuple(conf.shape)} do not match heat-maps {tuple(pred.shape)}")
        ws = torch.empty(_lib.lib().lp_heatmap_mse_workspace_bytes(s, k), device=pred.device, dtype=torch.uint8)
        loss = torch.empty(1, device=pred.device, dtype=torch.float32)
        check(_lib.lib().lp_unimodal_mse_fwd(_p(kp), _p(pred), _p(conf), s, k, img_h, img_w, h, w, sigma, thr, _p(loss), _p(ws),
                                             _stream()), "lp_unimodal_mse_fwd")
        ctx.save_for_backward(kp, pred, ws)
        ctx.args = (img_h, img_w, sigma)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        kp, pred, ws = ctx.saved_tensors
        img_h, img_w, sigma = ctx.args
        s, k, h, w = pred.shape
        g = torch.empty_like(pred)
        go = _f32c(gout).reshape(1)
        check(_lib.lib().lp_unimodal_mse_bwd(_p(kp), _p(pred), s, k, img_h, img_w, h, w, sigma, _p(ws), _p(go), _p(g), 0,
                                             _stream()), "lp_unimodal_mse_bwd")
        return None, g.to(ctx.in_dtype), None, None, None, None, None


def unimodal_mse(keypoints_pred_augmented: torch.Tensor, heatmaps_pred: torch.Tensor, confidences: torch.Tensor,
                 image_height: int, image_width: int, sigma: float = 1.25, prob_threshold: float = 0.0) -> torch.Tensor:
    return _UnimodalFn.apply(keypoints_pred_augmented, heatmaps_pred, confidences, int(image_height), int(image_width),
                             float(sigma), float(prob_threshold))


# --------------------------------------------------------------------------------------------------------
# keypoint-space losses
# --------------------------------------------------------------------------------------------------------

class _UnitGradFn(torch.autograd.Function):
    """loss whose gradient for unit upstream was produced together with the value (one launch)."""

    @staticmethod
    def forward(ctx, kp, loss, grad_unit):
        ctx.save_for_backward(grad_unit)
        ctx.shape = kp.shape
        return loss

    @staticmethod
    def backward(ctx, gout):
        (grad_unit,) = ctx.saved_tensors
        return (grad_unit * gout).reshape(ctx.shape), None, None


_EPS_CACHE: dict = {}


def _device_epsilon(epsilon: torch.Tensor, k: int, dev: torch.device) -> torch.Tensor:
    """per-keypoint epsilon vector on the device; uploaded once per (tensor, k, device) - a host-to-device copy every step would
    serialise the host with the launch queue and cannot be captured into a HIP graph"""
    key = (id(epsilon), k, str(dev))
    hit = _EPS_CACHE.get(key)
    if hit is not None and hit[0] is epsilon and hit[1] == epsilon._version:
        return hit[2]
    eps = epsilon.to(device=dev, dtype=torch.float32).reshape(-1)
    eps = eps.expand(k).contiguous() if eps.numel() == 1 else eps.contiguous()
    if eps.numel() != k:
        raise ValueError(f"temporal epsilon must be a scalar or have one entry per keypoint ({k}), got {eps.numel()}")
    if len(_EPS_CACHE) > 64:
        _EPS_CACHE.clear()
    _EPS_CACHE[key] = (epsilon, epsilon._version, eps)
    return eps


def temporal_loss(keypoints: torch.Tensor, confidences: torch.Tensor | None, epsilon: torch.Tensor, prob_threshold: float):
    """keypoints (S, 2K) -> scalar (reference losses/losses.py:674-703)."""
    require_device(keypoints)
    kp = _f32c(keypoints)
    s = kp.shape[0]
    k = kp.shape[1] // 2
    conf = _f32c(confidences) if confidences is not None else None
    if kp.dim() != 2 or kp.shape[1] % 2 or (conf is not None and tuple(conf.shape) != (s, k)):
        raise ValueError(f"temporal loss: keypoints {tuple(kp.shape)} must be (S, 2K) and confidences (S, K)")
    eps = _device_epsilon(epsilon, k, kp.device)
    loss = torch.empty(1, device=kp.device, dtype=torch.float32)
    grad = torch.empty_like(kp)
    check(_lib.lib().lp_temporal_fwd_bwd(_p(kp), _p(conf), s, k, _p(eps), float(prob_threshold), _p(loss), _p(grad), _stream()),
          "lp_temporal_fwd_bwd")
    return _UnitGradFn.apply(keypoints, loss.reshape(()), grad)


class _TemporalHeatmapFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, conf, eps, thr, kind):
        require_device(pred, conf)
        ctx.in_dtype = pred.dtype
        pred, conf = _f32c(pred), _f32c(conf)
        s, k, h, w = pred.shape
        ws = torch.empty(_lib.lib().lp_temporal_heatmap_workspace_bytes(s, k), device=pred.device, dtype=torch.uint8)
        loss = torch.empty(1, device=pred.device, dtype=torch.float32)
        check(_lib.lib().lp_temporal_heatmap_fwd(kind, _p(pred), _p(conf), s, k, h, w, _p(eps), float(thr), _p(loss), _p(ws), _stream()),
              "lp_temporal_heatmap_fwd")
        ctx.save_for_backward(pred, ws)
        ctx.kind = kind
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        pred, ws = ctx.saved_tensors
        s, k, h, w = pred.shape
        g = torch.empty_like(pred)
        go = _f32c(gout).reshape(1)
        check(_lib.lib().lp_temporal_heatmap_bwd(ctx.kind, _p(pred), s, k, h, w, _p(ws), _p(go), _p(g), 0, _stream()),
              "lp_temporal_heatmap_bwd")
        return g.to(ctx.in_dtype), None, None, None, None


def temporal_heatmap_distances(heatmaps_pred: torch.Tensor, kind: int) -> torch.Tensor:
    """(S, K, h, w) -> (S-1, K) distances between consecutive heat-maps (TemporalHeatmapLoss.compute_loss, reference :793-829);
    no autograd (the differentiable form is ``temporal_heatmap_loss``)."""
    require_device(heatmaps_pred)
    pred = _f32c(heatmaps_pred)
    s, k, h, w = pred.shape
    if s < 2:
        return torch.zeros(0, k, device=pred.device, dtype=torch.float32)
    ws = torch.empty(_lib.lib().lp_temporal_heatmap_workspace_bytes(s, k), device=pred.device, dtype=torch.uint8)
    conf = torch.ones(s, k, device=pred.device, dtype=torch.float32)
    eps = torch.zeros(k, device=pred.device, dtype=torch.float32)
    loss = torch.empty(1, device=pred.device, dtype=torch.float32)
    check(_lib.lib().lp_temporal_heatmap_fwd(kind, _p(pred), _p(conf), s, k, h, w, _p(eps), 0.0, _p(loss), _p(ws), _stream()),
          "lp_temporal_heatmap_fwd")
    return ws[: (s - 1) * k * 4].view(torch.float32).reshape(s - 1, k).clone()


def temporal_heatmap_loss(heatmaps_pred: torch.Tensor, confidences: torch.Tensor, epsilon: torch.Tensor, prob_threshold: float,
                          kind: int) -> torch.Tensor:
    """(S, K, h, w) heat-maps, (S, K) confidences -> scalar (reference losses/losses.py:841-869); kind = HM_MSE | HM_KL."""
    k = heatmaps_pred.shape[1]
    eps = _device_epsilon(epsilon, k, heatmaps_pred.device)
    if tuple(confidences.shape) != tuple(heatmaps_pred.shape[:2]):
        raise ValueError(f"confidences must be {tuple(heatmaps_pred.shape[:2])}, got {tuple(confidences.shape)}")
    return _TemporalHeatmapFn.apply(heatmaps_pred, confidences, eps, float(prob_threshold), kind)


def pca_loss(keypoints: torch.Tensor, index: torch.Tensor, mean: torch.Tensor, kept_eigenvectors: torch.Tensor, epsilon: float):
    """keypoints (S, 2K); index (rows, points) int32 keypoint ids per PCA sample -> scalar (losses/losses.py:548-573)."""
    require_device(keypoints, index, mean, kept_eigenvectors)
    kp = _f32c(keypoints)
    # raw pointers go to the kernel: every operand must be dense row-major fp32 / int32 (a transposed view of the eigenvector matrix,
    # for instance, keeps its strides through .to(device))
    mean, kept_eigenvectors = _f32c(mean), _f32c(kept_eigenvectors)
    index = index.to(torch.int32).contiguous()
    s, k = kp.shape[0], kp.shape[1] // 2
    rows, pts = index.shape
    if mean.numel() != 2 * pts or kept_eigenvectors.dim() != 2 or kept_eigenvectors.shape[1] != 2 * pts:
        raise ValueError(f"pca loss: mean {tuple(mean.shape)} / eigenvectors {tuple(kept_eigenvectors.shape)} do not span the "
                         f"{2 * pts} coordinates of a sample ({pts} points)")
    loss = torch.empty(1, device=kp.device, dtype=torch.float32)
    grad = torch.empty_like(kp)
    check(_lib.lib().lp_pca_fwd_bwd(_p(kp), s, k, _p(index), rows, pts, _p(mean), _p(kept_eigenvectors),
                                    kept_eigenvectors.shape[0], float(epsilon), _p(loss), _p(grad), _stream()), "lp_pca_fwd_bwd")
    return _UnitGradFn.apply(keypoints, loss.reshape(()), grad)


def rmse(keypoints_targ: torch.Tensor, keypoints_pred: torch.Tensor) -> torch.Tensor:
    require_device(keypoints_targ, keypoints_pred)
    t, p = _f32c(keypoints_targ), _f32c(keypoints_pred)
    if t.shape != p.shape or t.numel() % 2:
        raise ValueError(f"rmse: target {tuple(t.shape)} and predicted {tuple(p.shape)} keypoints must both be (B, 2K)")
    loss = torch.empty(1, device=p.device, dtype=torch.float32)
    check(_lib.lib().lp_rmse_fwd(_p(t), _p(p), t.numel() // 2, _p(loss), _stream()), "lp_rmse_fwd")
    return loss.reshape(())


class _LossCombineFn(torch.autograd.Function):
    """weighted[i] = w[i] * x[i];  total = sum_i a[i] * weighted[i]  (csrc/kploss.hip: lp_loss_combine, one launch each way)"""

    @staticmethod
    def forward(ctx, w, a, *xs):
        n = len(xs)
        dev = xs[0].device
        weighted = torch.empty(n, device=dev, dtype=torch.float32)
        total = torch.empty((), device=dev, dtype=torch.float32)
        keep = [x.detach().reshape(1) for x in xs]   # (views of the callers' scalars: alive for the launch)
        ptrs = (C.c_void_p * n)(*[_p(x) for x in keep])
        wa, aa = (C.c_float * n)(*w), (C.c_float * n)(*a)
        check(_lib.lib().lp_loss_combine(ptrs, wa, aa, n, _p(weighted), _p(total), _stream()), "lp_loss_combine")
        ctx.w, ctx.a, ctx.n = tuple(w), tuple(a), n
        return weighted, total

    @staticmethod
    def backward(ctx, g_weighted, g_total):
        n = ctx.n
        dev = g_total.device if g_total is not None else g_weighted.device
        gx = torch.empty(n, device=dev, dtype=torch.float32)
        gw = None if g_weighted is None else _f32c(g_weighted)
        gt = None if g_total is None else _f32c(g_total)
        check(_lib.lib().lp_loss_combine_bwd((C.c_float * n)(*ctx.w), (C.c_float * n)(*ctx.a), n, _p(gw), _p(gt), _p(gx), _stream()),
              "lp_loss_combine_bwd")
        return (None, None) + tuple(gx[i] for i in range(n))


def loss_combine(losses: list[torch.Tensor], weights: list[float], anneal: list[float]) -> tuple[torch.Tensor, torch.Tensor]:
    """LossFactory's weighted sum of device scalars -> ((n,) weighted values a_l w_l x_l, 0-dim total); differentiable.  Up to 8 losses are
    one launch each way (lp_loss_combine); more - the reference's registry holds more than 8 loss classes and its factory sums any number,
    losses/factory.py:229-285 - go through groups of 8 whose totals are combined the same way."""
    require_device(*losses)
    if len(losses) == 0:
        raise ValueError("a LossFactory needs at least one loss")
    if any(x.numel() != 1 for x in losses):
        raise ValueError("every loss must be a scalar")
    xs = [x if x.dtype == torch.float32 else x.to(torch.float32) for x in losses]
    w, a = [float(v) for v in weights], [float(v) for v in anneal]
    if len(xs) <= 8:
        return _LossCombineFn.apply(w, a, *xs)
    parts = [_LossCombineFn.apply(w[i:i + 8], a[i:i + 8], *xs[i:i + 8]) for i in range(0, len(xs), 8)]
    _, total = loss_combine([p[1] for p in parts], [1.0] * len(parts), [1.0] * len(parts))
    return torch.cat([p[0] for p in parts]), total


# --------------------------------------------------------------------------------------------------------
# batch producers (csrc/frames.hip)
# --------------------------------------------------------------------------------------------------------

def _frame_norm(mean, std) -> _lib.FrameNorm:
    return _lib.FrameNorm((C.c_float * 3)(*[float(m) for m in mean]), (C.c_float * 3)(*[float(v) for v in std]))


def frames_resize(frames_u8: torch.Tensor, height: int, width: int, border: str = "clamp", mean=None, std=None,
                  interpolation: str = "linear") -> torch.Tensor:
    """(S, Hs, Ws, 3) uint8 device frames -> resized frames.  Without mean/std: fp32 (S, height, width, 3) in [0, 255] (input of
    ``frames_augment``); with them: fp32 (S, 3, height, width) normalised planes in one launch.
    ``interpolation``: "linear" = antialiased triangle filter (DALI ``fn.resize``, the video pipeline); "cubic" = bicubic without
    antialiasing, rounded to uint8 levels (imgaug ``iaa.Resize``'s default = OpenCV INTER_CUBIC, the labeled images)."""
    require_device(frames_u8)
    if interpolation not in ("linear", "cubic"):
        raise ValueError(f"interpolation must be 'linear' or 'cubic', got {interpolation!r}")
    if frames_u8.dtype != torch.uint8 or frames_u8.dim() != 4 or frames_u8.shape[-1] != 3:
        raise ValueError(f"frames must be uint8 (S, H, W, 3), got {frames_u8.dtype} {tuple(frames_u8.shape)}")
    if border not in ("clamp", "renorm"):
        raise ValueError(f"border must be 'clamp' or 'renorm', got {border!r}")
    if frames_u8.stride(3) != 1 or frames_u8.stride(2) != 3:
        frames_u8 = frames_u8.contiguous()
    s, hs, ws, _ = frames_u8.shape
    finish = mean is not None
    out = torch.empty((s, 3, height, width) if finish else (s, height, width, 3), device=frames_u8.device, dtype=torch.float32)
    norm = _frame_norm(mean, std) if finish else None
    if interpolation == "cubic":
        check(_lib.lib().lp_frames_resize_cubic(_p(frames_u8), s, hs, ws, frames_u8.stride(0), frames_u8.stride(1), int(height), int(width), 1,
                                                C.byref(norm) if finish else None, _p(out), _stream()), "lp_frames_resize_cubic")
        return out
    check(_lib.lib().lp_frames_resize(_p(frames_u8), s, hs, ws, frames_u8.stride(0), frames_u8.stride(1), int(height), int(width),
                                      _lib.BORDER_CLAMP if border == "clamp" else _lib.BORDER_RENORM,
                                      C.byref(norm) if finish else None, _p(out), _stream()), "lp_frames_resize")
    return out


def frames_augment(frames_hwc: torch.Tensor, mean, std, matrix=None, brightness: float = 1.0, contrast: float = 1.0,
                   contrast_center: float = 0.5, shot_factor: float = 0.0, seed: int = 0) -> torch.Tensor:
    """fp32 (S, H, W, 3) in [0, 255] -> warp_affine(matrix: source -> destination, fill 0) -> brightness / contrast -> shot
    noise -> /255 -> normalise -> fp32 (S, 3, H, W).  ``matrix`` is a host (2, 3) array-like (one DALI sample = one sequence)."""
    require_device(frames_hwc)
    x = _f32c(frames_hwc)
    s, h, w, c = x.shape
    if c != 3:
        raise ValueError(f"frames must be (S, H, W, 3), got {tuple(x.shape)}")
    m = [0.0] * 6 if matrix is None else [float(v) for v in np.asarray(matrix, dtype=np.float64).reshape(-1)]
    if len(m) != 6:
        raise ValueError("matrix must have 6 entries (2, 3)")
    aug = _lib.FrameAugment(int(matrix is not None), (C.c_float * 6)(*m), float(brightness), float(contrast), float(contrast_center),
                            float(shot_factor), int(seed) & 0xFFFFFFFFFFFFFFFF)
    norm = _frame_norm(mean, std)
    out = torch.empty(s, 3, h, w, device=x.device, dtype=torch.float32)
    check(_lib.lib().lp_frames_augment(_p(x), s, h, w, C.byref(aug), C.byref(norm), _p(out), _stream()), "lp_frames_augment")
    return out


def labeled_keypoints(keypoints: torch.Tensor, src_hw: torch.Tensor, height: int, width: int, affine: torch.Tensor | None = None,
                      hflip: torch.Tensor | None = None, swap: torch.Tensor | None = None, visibility: torch.Tensor | None = None,
                      uniform_heatmaps: bool = False) -> tuple[torch.Tensor, torch.Tensor]:
    """(B, K, 2) source-px labels -> (model-px keypoints with out-of-frame points NaN, visibility (B, K) int32)."""
    require_device(keypoints)
    kp = _f32c(keypoints)
    b, k, _ = kp.shape
    dev = kp.device
    hw = _f32c(src_hw.to(dev))
    i32 = lambda t: None if t is None else t.to(device=dev, dtype=torch.int32).contiguous()  # noqa: E731
    aff = None if affine is None else _f32c(affine.to(dev))
    fl, sw, vi = i32(hflip), i32(swap), i32(visibility)
    if hw.shape != (b, 2) or (aff is not None and aff.shape != (b, 2, 3)) or (sw is not None and sw.numel() != k):
        raise ValueError("labeled_keypoints: src_hw must be (B, 2), affine (B, 2, 3), swap (K,)")
    out = torch.empty_like(kp)
    vis = torch.empty(b, k, device=dev, dtype=torch.int32)
    check(_lib.lib().lp_labeled_keypoints(_p(kp), _p(hw), _p(aff), _p(fl), _p(sw), _p(vi), int(uniform_heatmaps), b, k, int(height),
                                          int(width), _p(out), _p(vis), _stream()), "lp_labeled_keypoints")
    return out, vis
