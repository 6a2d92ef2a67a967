"""Self-contained fp32 CPU restatement of the Lightning Pose heatmap-tracker step arithmetic.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): the checker for the HIP path and the
timed CPU baseline.  Never imported by the product package.

Every function cites the reference file:line it restates (paths relative to the reference
tree, paninski-lab/lightning-pose v2.4.0).  The restatement is written as closed-form tensor
expressions (banded interpolation matrices, masked reductions) rather than the reference's
op sequence, so it doubles as an independent derivation; it is pinned against the verbatim
reference modules and the reference's known-answer tests in tests/test_oracle_*.py and
against tests/golden/*.npz.

Parity status: pinned for everything except ``unimodal_mse_loss`` (no such loss exists in the
reference snapshot - SURVEY.md F3 - "parity unpinned") and the image operators of the batch
producers, ``frames_*`` / ``brightness_contrast`` (DALI / imgaug are not vendored - "parity unpinned").
"""

from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import thirdparty as tp

# ======================================================================================
# decode: heatmap -> (keypoints, confidence)
# ======================================================================================

_BINOMIAL5 = (1.0 / 16.0, 4.0 / 16.0, 6.0 / 16.0, 4.0 / 16.0, 1.0 / 16.0)


def bicubic_up2_matrix(n: int, dtype=torch.float64) -> torch.Tensor:
    """(2n, n) matrix of torch's bicubic x2 upsample along one axis.

    F.interpolate(mode="bicubic", align_corners=False): A = -0.75, source coordinate
    (dst + 0.5) / 2 - 0.5, taps floor(src)-1..+2 with border index clamping.
    (models/heads/heatmap.py:94-97 calls it; the arithmetic lives in ATen.)
    """
    a = -0.75
    m = torch.zeros(2 * n, n, dtype=dtype)
    for o in range(2 * n):
        src = (o + 0.5) / 2.0 - 0.5
        i0 = math.floor(src)
        t = src - i0
        w = (
            ((a * (t + 1) - 5 * a) * (t + 1) + 8 * a) * (t + 1) - 4 * a,
            ((a + 2) * t - (a + 3)) * t * t + 1,
            ((a + 2) * (1 - t) - (a + 3)) * (1 - t) * (1 - t) + 1,
            ((a * (2 - t) - 5 * a) * (2 - t) + 8 * a) * (2 - t) - 4 * a,
        )
        for k in range(4):
            idx = min(max(i0 - 1 + k, 0), n - 1)
            m[o, idx] += w[k]
    return m


def binomial_blur_matrix(n: int, dtype=torch.float64) -> torch.Tensor:
    """(n, n) banded matrix of the 1-D [1,4,6,4,1]/16 filter with ZERO padding.

    kornia filter2d(border_type="constant") with the separable pyramid kernel
    (models/heads/heatmap.py:93,99).
    """
    m = torch.zeros(n, n, dtype=dtype)
    for o in range(n):
        for k, wv in enumerate(_BINOMIAL5):
            i = o + k - 2
            if 0 <= i < n:
                m[o, i] += wv
    return m


def upsample_matrix(n: int, levels: int, dtype=torch.float64) -> torch.Tensor:
    """(n * 2**levels, n) composite of `levels` x (bicubic x2 then binomial blur)."""
    u = torch.eye(n, dtype=dtype)
    size = n
    for _ in range(levels):
        u = binomial_blur_matrix(2 * size, dtype) @ bicubic_up2_matrix(size, dtype) @ u
        size *= 2
    return u


def upsample2x(x: torch.Tensor) -> torch.Tensor:
    """models/heads/heatmap.py:86-100 `upsample`, op-for-op (used to pin the matrix form)."""
    h, w = x.shape[-2:]
    up = F.interpolate(x, size=(2 * h, 2 * w), mode="bicubic", align_corners=False)
    return tp.filter2d(up, tp.get_pyramid_gaussian_kernel(), border_type="constant")


def confidence_window(prob: torch.Tensor, locs: torch.Tensor, sigma: float = 1.25,
                      num_stds: int = 2) -> torch.Tensor:
    """data/heatmaps.py:90-142: sum of `prob` over the (2r+1)^2 window at trunc(loc), r=floor(sigma*num_stds),
    zero outside the map."""
    r = int(np.floor(sigma * num_stds))
    b, k, h, w = prob.shape
    padded = F.pad(prob, (r, r, r, r))
    cx = locs[..., 0].to(torch.int64) + r  # trunc toward zero, as .type(int64)
    cy = locs[..., 1].to(torch.int64) + r
    bi = torch.arange(b).view(b, 1).expand(b, k)
    ki = torch.arange(k).view(1, k).expand(b, k)
    total = torch.zeros(b, k, dtype=prob.dtype)
    for dy in range(-r, r + 1):
        for dx in range(-r, r + 1):
            total = total + padded[bi, ki, cy + dy, cx + dx]
    return total


def soft_argmax(heatmaps: torch.Tensor, downsample_factor: int = 2,
                temperature: float = 1000.0) -> tuple[torch.Tensor, torch.Tensor]:
    """models/heads/heatmap.py:103-144 `run_subpixelmaxima`.

    Returns keypoints (B, 2K) in model-input pixels (x0,y0,x1,y1,...) and confidences (B, K).
    """
    x = heatmaps
    for _ in range(downsample_factor):
        x = upsample2x(x)
    b, k, hh, ww = x.shape
    p = F.softmax(x.reshape(b, k, -1) * temperature, dim=-1).reshape(b, k, hh, ww)
    xs = torch.arange(ww, dtype=x.dtype)
    ys = torch.arange(hh, dtype=x.dtype)
    ex = (p.sum(dim=2) * xs).sum(-1)
    ey = (p.sum(dim=3) * ys).sum(-1)
    locs = torch.stack([ex, ey], dim=-1)
    conf = confidence_window(p, locs)
    offset = {1: 0.5, 2: 1.5, 3: 2.5}.get(downsample_factor, 0.0)
    return (locs - offset).reshape(b, 2 * k), conf


# ======================================================================================
# heat-map targets
# ======================================================================================


def generate_heatmaps(keypoints: torch.Tensor, height: int, width: int, output_shape: tuple[int, int],
                      sigma: float = 1.25, visibility: torch.Tensor | None = None,
                      keep_gradients: bool = False) -> torch.Tensor:
    """data/heatmaps.py:11-87.  keypoints (B,K,2) in image px -> (B,K,h,w) targets."""
    kp = keypoints if keep_gradients else keypoints.detach()
    h, w = output_shape
    x = kp[..., 0] * (w / width)
    y = kp[..., 1] * (h / height)
    bad = torch.isnan(x) | (x < -1) | (x > w + 1) | (y < -1) | (y > h + 1)
    xc = x.clamp(-1, w + 1)[..., None, None]
    yc = y.clamp(-1, h + 1)[..., None, None]
    gx = torch.arange(w, dtype=kp.dtype).view(1, 1, 1, w)
    gy = torch.arange(h, dtype=kp.dtype).view(1, 1, h, 1)
    g = torch.exp(-((gx - xc) ** 2 + (gy - yc) ** 2) / (2.0 * sigma ** 2))
    g = g / g.sum(dim=(2, 3), keepdim=True)
    zero = torch.zeros(h, w, dtype=kp.dtype)
    if visibility is None:
        g = torch.where(bad[..., None, None], zero, g)
    else:
        uniform = torch.full((h, w), 1.0 / (h * w), dtype=kp.dtype)
        g = torch.where((visibility == 0)[..., None, None], zero, g)
        g = torch.where((visibility == 1)[..., None, None], uniform, g)
        g = torch.where(((visibility == 2) & bad)[..., None, None], zero, g)
    return g


# ======================================================================================
# geometry
# ======================================================================================


def undo_affine(keypoints: torch.Tensor, transforms: torch.Tensor, is_multiview: bool = False) -> torch.Tensor:
    """data/utils.py:142-234.  keypoints (S,2K); transforms (2,3) | (S,2,3) | (V,2,3) | sentinel (.., 1).

    x_orig = A^-1 (x_aug - t) for the augmentation x_aug = A x_orig + t.
    """
    if transforms.shape[-1] != 3:
        return keypoints
    s = keypoints.shape[0]
    kp = keypoints.reshape(s, -1, 2)

    def _apply(pts: torch.Tensor, tf: torch.Tensor) -> torch.Tensor:
        tf = tf.detach().to(pts.dtype)
        if tf.dim() == 2:
            tf = tf.unsqueeze(0)
        a, t = tf[:, :, :2], tf[:, :, 2]
        det = a[:, 0, 0] * a[:, 1, 1] - a[:, 0, 1] * a[:, 1, 0]
        inv = torch.stack([
            torch.stack([a[:, 1, 1], -a[:, 0, 1]], -1),
            torch.stack([-a[:, 1, 0], a[:, 0, 0]], -1)], 1) / det[:, None, None]
        d = pts - t[:, None, :]
        return torch.einsum("bij,bkj->bki", inv, d)

    if not is_multiview:
        out = _apply(kp, transforms)
    else:
        v = transforms.shape[0]
        per = kp.shape[1] // v
        out = torch.cat([_apply(kp[:, i * per:(i + 1) * per], transforms[i]) for i in range(v)], dim=1)
    return out.reshape(s, -1)


def model_to_frame(keypoints: torch.Tensor, model_height: int, model_width: int, bbox: torch.Tensor,
                   num_views: int = 1) -> torch.Tensor:
    """data/bboxes.py:222-288 (+ norm_to_frame :74-105), out of place.  bbox rows are [x, y, h, w] per view."""
    b = keypoints.shape[0]
    kp = keypoints.reshape(b, -1, 2)
    per = kp.shape[1] // num_views
    outs = []
    for v in range(num_views):
        bb = bbox[:, 4 * v:4 * v + 4].to(kp.dtype)
        sl = kp[:, v * per:(v + 1) * per]
        xs = sl[..., 0] / model_width * bb[:, 3:4] + bb[:, 0:1]
        ys = sl[..., 1] / model_height * bb[:, 2:3] + bb[:, 1:2]
        outs.append(torch.stack([xs, ys], -1))
    return torch.cat(outs, dim=1).reshape(b, -1)


# ======================================================================================
# batch producers.  The image operators live in NVIDIA DALI / imgaug, which
# are not vendored and not installable here: their PUBLISHED definitions are restated, with
# torch's own resampling (F.interpolate antialias, F.grid_sample) as the independent check.
# Parity of the IMAGE half is UNPINNED; the keypoint / visibility half is pinned against the
# verbatim HeatmapDataset.compute_heatmap (tests/golden/labeled_targets.npz).
# ======================================================================================


def frames_resize(frames_u8: torch.Tensor, height: int, width: int, border: str = "renorm") -> torch.Tensor:
    """(S, Hs, Ws, 3) uint8 -> (S, height, width, 3) fp32 in [0, 255]: linear interpolation with antialiasing (triangle
    filter of radius max(1, scale) source pixels, half-pixel centres) = DALI fn.resize defaults (interp_type=INTERP_LINEAR,
    antialias=True; data/video/dali.py:151-152).  border="renorm": the window is cut at the image edge and renormalised
    (exactly torch / PIL antialiased bilinear); border="clamp": edge pixels are replicated under the full window."""
    x = frames_u8.permute(0, 3, 1, 2).to(torch.float64)
    if border == "renorm":
        y = F.interpolate(x, size=(height, width), mode="bilinear", align_corners=False, antialias=True)
        return y.permute(0, 2, 3, 1).to(torch.float32)

    def axis_matrix(n_src: int, n_dst: int) -> torch.Tensor:
        scale = n_src / n_dst
        r = max(1.0, scale)
        m = torch.zeros(n_dst, n_src, dtype=torch.float64)
        for i in range(n_dst):
            c = (i + 0.5) * scale
            lo, hi = math.floor(c - r + 0.5), math.floor(c + r + 0.5)
            w = [max(0.0, 1.0 - abs((j + 0.5 - c) / r)) for j in range(lo, hi)]
            tot = sum(w)
            for j, wj in zip(range(lo, hi), w):
                m[i, min(max(j, 0), n_src - 1)] += wj / tot
        return m

    my, mx = axis_matrix(x.shape[2], height), axis_matrix(x.shape[3], width)
    y = torch.einsum("ij,scjk,lk->scil", my, x, mx)
    return y.permute(0, 2, 3, 1).to(torch.float32)


def frames_resize_cubic(frames_u8: torch.Tensor, height: int, width: int, round_u8: bool = True) -> torch.Tensor:
    """(S, Hs, Ws, 3) uint8 -> (S, height, width, 3) fp32: bicubic without antialiasing (Keys kernel A = -0.75, half-pixel centres, taps
    clamped) = imgaug ``iaa.Resize``'s default interpolation "cubic" = OpenCV INTER_CUBIC, the last imgaug step the reference's dataset
    applies to every labeled image (data/datasets.py:137-143).  torch's bicubic uses the same kernel, centres and clamping; imgaug returns a
    uint8 image, hence ``round_u8`` (OpenCV's 8-bit path evaluates the same polynomial with 11-bit fixed-point weights, so single levels
    may differ from it - imgaug / OpenCV are not installed here: parity unpinned against them, pinned against torch's bicubic)."""
    x = frames_u8.permute(0, 3, 1, 2).to(torch.float64)
    y = F.interpolate(x, size=(height, width), mode="bicubic", align_corners=False)
    if round_u8:
        y = torch.floor(y + 0.5).clamp(0, 255)
    return y.permute(0, 2, 3, 1).to(torch.float32)


IMAGENET_MEAN = (0.485, 0.456, 0.406)  # data/__init__.py:46-47
IMAGENET_STD = (0.229, 0.224, 0.225)


def frames_finish(frames_hwc: torch.Tensor, mean=IMAGENET_MEAN, std=IMAGENET_STD) -> torch.Tensor:
    """(S,H,W,3) in [0,255] -> (S,3,H,W): /255 then fn.crop_mirror_normalize(mean, std, output_layout="FCHW")
    (data/video/dali.py:180-188)."""
    x = frames_hwc.permute(0, 3, 1, 2) / 255.0
    return (x - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)


def frames_warp_affine(frames_hwc: torch.Tensor, matrix: torch.Tensor) -> torch.Tensor:
    """fn.warp_affine(matrix, fill_value=0, inverse_map=False), linear interpolation, output size = input size
    (data/video/dali.py:162-165): ``matrix`` (2,3) maps source to destination coordinates with pixel centres at half
    integers; samples outside the source contribute 0.  Evaluated with torch's grid_sample as the independent sampler."""
    s, h, w, _ = frames_hwc.shape
    a = torch.eye(3, dtype=torch.float64)
    a[:2] = matrix.to(torch.float64)
    inv = torch.linalg.inv(a)
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float64) + 0.5, torch.arange(w, dtype=torch.float64) + 0.5, indexing="ij")
    sx = inv[0, 0] * xs + inv[0, 1] * ys + inv[0, 2]  # source position, pixel-centre-at-half-integer convention
    sy = inv[1, 0] * xs + inv[1, 1] * ys + inv[1, 2]
    grid = torch.stack([2 * sx / w - 1, 2 * sy / h - 1], -1).unsqueeze(0).expand(s, h, w, 2)
    out = F.grid_sample(frames_hwc.permute(0, 3, 1, 2).to(torch.float64), grid, mode="bilinear", padding_mode="zeros", align_corners=False)
    return out.permute(0, 2, 3, 1).to(torch.float32)


def brightness_contrast(x: torch.Tensor, brightness: float, contrast: float, contrast_center: float = 0.5) -> torch.Tensor:
    """fn.brightness_contrast: out = brightness_shift * range + brightness * (center + contrast * (in - center)) with
    brightness_shift = 0; for float input DALI's default contrast_center is 0.5 (data/video/dali.py:167-169 feeds it the
    reader's FLOAT frames in [0, 255])."""
    return brightness * (contrast_center + contrast * (x - contrast_center))


def labeled_keypoints(keypoints: torch.Tensor, src_hw: torch.Tensor, height: int, width: int, affine: torch.Tensor | None = None,
                      hflip: torch.Tensor | None = None, swap: torch.Tensor | None = None, visibility: torch.Tensor | None = None,
                      uniform_heatmaps: bool = False) -> tuple[torch.Tensor, torch.Tensor]:
    """(B,K,2) source px -> model px + visibility, as the labeled dataset does it: optional augmentation affine on source
    px, imgaug Resize keypoint projection ``x / from_w * to_w`` (data/datasets.py:137-142, :279-286), optional horizontal flip
    ``x = W - x`` followed by the left/right keypoint permutation (:288-293), keypoints outside [0,W) x [0,H) -> NaN
    (:496-508); visibility as stored or synthesised from the stored label's NaNs (:465-472), permuted with the flip (:364-366)."""
    kp = keypoints.clone().to(torch.float32)
    b, k, _ = kp.shape
    if visibility is None:
        nan = torch.isnan(kp[:, :, 0])
        vis = torch.where(nan, torch.full_like(nan, 1 if uniform_heatmaps else 0, dtype=torch.long), torch.full_like(nan, 2, dtype=torch.long))
    else:
        vis = visibility.clone().long()
    if affine is not None:
        x = affine[:, None, 0, 0] * kp[..., 0] + affine[:, None, 0, 1] * kp[..., 1] + affine[:, None, 0, 2]
        y = affine[:, None, 1, 0] * kp[..., 0] + affine[:, None, 1, 1] * kp[..., 1] + affine[:, None, 1, 2]
        kp = torch.stack([x, y], -1)
    kp[..., 0] = kp[..., 0] / src_hw[:, None, 1] * width
    kp[..., 1] = kp[..., 1] / src_hw[:, None, 0] * height
    if hflip is not None:
        for i in range(b):
            if bool(hflip[i]):
                kp[i, :, 0] = width - kp[i, :, 0]
                if swap is not None:
                    kp[i] = kp[i][swap.long()]
                    vis[i] = vis[i][swap.long()]
    out = (kp[..., 0] < 0) | (kp[..., 1] < 0) | (kp[..., 0] >= width) | (kp[..., 1] >= height)
    kp[out] = float("nan")
    return kp, vis


# ======================================================================================
# losses
# ======================================================================================


def loss_weight(log_weight: float) -> float:
    """losses/losses.py:89-100: 1 / (2 exp(log_weight))."""
    return 1.0 / (2.0 * math.exp(log_weight))


def _valid_rows(targets: torch.Tensor) -> torch.Tensor:
    return ~(targets.flatten(2) == 0).all(dim=-1)


def heatmap_mse_loss(targets: torch.Tensor, predictions: torch.Tensor) -> torch.Tensor:
    """losses/losses.py:229-290,314-335: mean over valid (b,k) rows and pixels of (t-p)^2 * h * w."""
    v = _valid_rows(targets)
    h, w = targets.shape[-2:]
    return ((targets[v] - predictions[v]) ** 2 * (h * w)).mean()


def heatmap_kl_loss(targets: torch.Tensor, predictions: torch.Tensor) -> torch.Tensor:
    """losses/losses.py:360-379."""
    v = _valid_rows(targets)
    t, p = targets[v].unsqueeze(0) + 1e-10, predictions[v].unsqueeze(0) + 1e-10
    return tp.kl_div_loss_2d(p, t, reduction="none").mean()


def heatmap_js_loss(targets: torch.Tensor, predictions: torch.Tensor) -> torch.Tensor:
    """losses/losses.py:404-423."""
    v = _valid_rows(targets)
    t, p = targets[v].unsqueeze(0) + 1e-10, predictions[v].unsqueeze(0) + 1e-10
    return tp.js_div_loss_2d(p, t, reduction="none").mean()


def temporal_loss(keypoints: torch.Tensor, confidences: torch.Tensor | None = None,
                  epsilon: float | torch.Tensor = 0.0, prob_threshold: float = 0.0) -> torch.Tensor:
    """losses/losses.py:576-703: mean over ALL (S-1)*K of relu(mask * ||kp[t+1]-kp[t]|| - eps_k)."""
    s = keypoints.shape[0]
    d = (keypoints[1:] - keypoints[:-1]).reshape(s - 1, -1, 2)
    dist = torch.linalg.norm(d, ord=2, dim=2)  # same op as the reference: zero sub-gradient at d = 0
    if confidences is not None:
        low = confidences < prob_threshold
        dist = torch.where(low[1:] | low[:-1], torch.zeros_like(dist), dist)
    eps = torch.as_tensor(epsilon, dtype=dist.dtype)
    return F.relu(dist - eps).mean()


def temporal_heatmap_loss(heatmaps: torch.Tensor, confidences: torch.Tensor, epsilon: float | torch.Tensor = 0.0,
                          prob_threshold: float = 0.0, kind: str = "mse") -> torch.Tensor:
    """losses/losses.py:706-869 `TemporalHeatmapLoss`: per (t, k) the pixel-mean squared difference of consecutive heat-maps ("mse",
    :815-819) or kornia's kl_div_loss_2d(pred = h_t + 1e-10, target = h_t+1 + 1e-10) = sum target (log target - log pred) ("kl", :820-826);
    zero where either frame's confidence is below the threshold (:783-791); relu(. - epsilon_k) (:763); mean over all (S-1) K (:865)."""
    a, b = heatmaps[:-1], heatmaps[1:]
    if kind == "mse":
        d = ((a - b) ** 2).mean(dim=(-1, -2))
    else:
        d = ((b + 1e-10) * (torch.log(b + 1e-10) - torch.log(a + 1e-10))).sum(dim=(-1, -2))
    ignore = confidences < prob_threshold
    d = torch.where(ignore[:-1] | ignore[1:], torch.zeros_like(d), d)
    eps = torch.as_tensor(epsilon, dtype=d.dtype).reshape(1, -1)
    return F.relu(d - eps).mean()


def pca_format_singleview(keypoints: torch.Tensor, columns: list[int] | None) -> torch.Tensor:
    """utils/pca.py:124-163 (no centring)."""
    kp = keypoints.reshape(keypoints.shape[0], -1, 2)
    if columns is not None:
        kp = kp[:, list(columns)]
    return kp.reshape(kp.shape[0], -1)


def pca_format_multiview(keypoints: torch.Tensor, mirrored_column_matches: list[list[int]]) -> torch.Tensor:
    """utils/pca.py:97-122,759-792: rows = (frame, selected keypoint), cols = [x_v0,y_v0,x_v1,y_v1,...]."""
    kp = keypoints.reshape(keypoints.shape[0], -1, 2)
    cols = [kp[:, list(m)].reshape(-1, 2) for m in mirrored_column_matches]
    return torch.cat(cols, dim=1)


def pca_reprojection_error(data: torch.Tensor, mean: torch.Tensor, kept: torch.Tensor) -> torch.Tensor:
    """utils/pca.py:266-309: per-2D-keypoint norm of x - ((x-mu) V^T V + mu)."""
    c = data - mean
    resid = c - (c @ kept.T) @ kept
    return torch.linalg.norm(resid.reshape(resid.shape[0], -1, 2), dim=2)


def pca_loss(data: torch.Tensor, mean: torch.Tensor, kept: torch.Tensor, epsilon: float | torch.Tensor) -> torch.Tensor:
    """losses/losses.py:548-573 after formatting."""
    eps = torch.as_tensor(epsilon, dtype=data.dtype)
    return F.relu(pca_reprojection_error(data, mean, kept) - eps).mean()


def rmse_loss(keypoints_targ: torch.Tensor, keypoints_pred: torch.Tensor) -> torch.Tensor:
    """losses/losses.py:880-996: mean over non-NaN targets of per-keypoint Euclidean distance / sqrt(2)."""
    mask = keypoints_targ == keypoints_targ
    t = keypoints_targ[mask].reshape(-1, 2)
    p = keypoints_pred[mask].reshape(-1, 2)
    return torch.sqrt(((t - p) ** 2).mean(dim=1)).mean()


def unimodal_mse_loss(keypoints_pred_augmented: torch.Tensor, heatmaps_pred: torch.Tensor,
                      confidences: torch.Tensor, image_height: int, image_width: int,
                      prob_threshold: float = 0.0, sigma: float = 1.25) -> torch.Tensor:
    """"unimodal_mse" - NOT IN THE REFERENCE SNAPSHOT: parity UNPINNED.

    Definition adopted here (structurally the in-tree ReprojectionHeatmapLoss, losses/losses.py:1129-1260,
    composed from the pinned `generate_heatmaps` and the heatmap-MSE arithmetic):
      ideal = generate_heatmaps(kp_aug, H, W, (h, w), sigma)           (detached)
      keep (s,k) with confidence >= prob_threshold AND a non-zero ideal heatmap
      loss = mean over kept (s,k) and pixels of (ideal - pred)^2 * h * w     (0 if none kept)
    """
    s, k, h, w = heatmaps_pred.shape
    ideal = generate_heatmaps(keypoints_pred_augmented.detach().reshape(s, k, 2), image_height, image_width,
                              (h, w), sigma=sigma)
    keep = (confidences >= prob_threshold) & _valid_rows(ideal)
    if not bool(keep.any()):
        return (heatmaps_pred * 0.0).sum()
    return ((ideal[keep] - heatmaps_pred[keep]) ** 2 * (h * w)).mean()


_HEATMAP_LOSSES = ("heatmap_mse", "heatmap_kl", "heatmap_js")


def factory_total(losses: dict[str, tuple[torch.Tensor, float]], anneal_weight: float | None) -> torch.Tensor:
    """losses/factory.py:229-285: sum_l a_l * w_l * L_l with a_l = 1 for heatmap losses / anneal None."""
    tot = torch.tensor(0.0)
    for name, (val, log_w) in losses.items():
        a = 1.0 if (anneal_weight is None or name in _HEATMAP_LOSSES) else float(anneal_weight)
        tot = tot + a * loss_weight(log_w) * val
    return tot


# ======================================================================================
# PCA fit (CPU, once) - utils/pca.py:205-264,419-564,611-738
# ======================================================================================


def fit_pca(data: np.ndarray, components_to_keep: int | float | None, loss_type: str = "pca_singleview",
            epsilon_percentile: float = 99.0) -> dict[str, np.ndarray | float | int]:
    """NaN-aware covariance-eigh PCA with sklearn's sign convention, component choice and empirical epsilon."""
    x = np.asarray(data, dtype=np.float64)
    mean = np.nanmean(x, axis=0)
    cov = np.ma.cov(np.ma.masked_invalid(x), rowvar=False).data
    evals, evecs = np.linalg.eigh(cov)
    evals, evecs = evals[::-1].copy(), evecs[:, ::-1].copy()
    evals[evals < 0] = 0.0
    vt = evecs.T
    # svd_flip(u_based_decision=False): make the largest-|.| entry of each row positive
    idx = np.argmax(np.abs(vt), axis=1)
    signs = np.sign(vt[np.arange(vt.shape[0]), idx])
    signs[signs == 0] = 1.0
    vt = vt * signs[:, None]
    ncomp_all = min(x.shape)
    vt, evals = vt[:ncomp_all], evals[:ncomp_all]
    ratio = evals / evals.sum() if evals.sum() > 0 else evals
    if loss_type == "pca_multiview":
        n_keep = 3
    elif type(components_to_keep) is int:
        n_keep = components_to_keep
    elif type(components_to_keep) is float:
        n_keep = ncomp_all if components_to_keep == 1.0 else int(np.where(np.cumsum(ratio) >= components_to_keep)[0][0]) + 1
    else:
        raise TypeError("components_to_keep must be int or float")
    kept = vt[:n_keep]
    mean32 = torch.tensor(mean, dtype=torch.float32)
    kept32 = torch.tensor(kept, dtype=torch.float32)
    err = pca_reprojection_error(torch.tensor(x, dtype=torch.float32), mean32, kept32).numpy()
    eps = float(np.nanpercentile(err.flatten(), epsilon_percentile, axis=0))
    return {"mean": mean32.numpy(), "kept_eigenvectors": kept32.numpy(), "epsilon": np.float32(eps),
            "n_components_kept": n_keep, "explained_variance_ratio": ratio}


# ======================================================================================
# model pieces (pure torch nn; the "plain PyTorch fp32 reference" for the backbone/head)
# ======================================================================================


def resnet50_trunk() -> nn.Sequential:
    """models/backbones/factory.py:322-325,337-348: children[:-2] of torchvision resnet50."""
    return nn.Sequential(*list(tp.resnet50(weights=None).children())[:-2])


def make_head(in_channels: int, num_keypoints: int, n_layers: int) -> nn.Sequential:
    """models/heads/heatmap.py:20-83: PixelShuffle(2) + n_layers ConvTranspose2d(k3,s2,p1,op1), xavier(gain .01)."""
    layers: list[nn.Module] = [nn.PixelShuffle(2)]
    cin = in_channels // 4
    for _ in range(n_layers):  # construct everything first (default inits draw from the RNG) ...
        layers.append(nn.ConvTranspose2d(cin, num_keypoints, 3, stride=2, padding=1, output_padding=1))
        cin = num_keypoints
    for ct in layers[1:]:      # ... then re-initialise in order, as the reference does
        nn.init.xavier_uniform_(ct.weight, gain=0.01)
        nn.init.zeros_(ct.bias)
    return nn.Sequential(*layers)


class OracleTracker(nn.Module):
    """HeatmapTracker / SemiSupervisedHeatmapTracker forward + loss assembly, restated.

    models/heatmap_tracker.py:107-153,264-286; models/base.py:504-546,627-701.
    Construction order (seed -> backbone -> head) follows heatmap_tracker.py:69-94 so seeded
    weights equal the reference's.
    """

    def __init__(self, num_keypoints: int, downsample_factor: int = 2, torch_seed: int = 123,
                 image_size: int = 256):
        super().__init__()
        torch.manual_seed(torch_seed)
        self.backbone = resnet50_trunk()
        n_layers = int(math.log2(32)) - downsample_factor - 1
        self.head = nn.Module()
        self.head.upsampling_layers = make_head(2048, num_keypoints, n_layers)
        self.num_keypoints = num_keypoints
        self.downsample_factor = downsample_factor

    def forward(self, images: torch.Tensor) -> torch.Tensor:
        shape = images.shape
        if images.dim() > 4:
            images = images.reshape(-1, *shape[-3:])
        hm = self.head.upsampling_layers(self.backbone(images))
        hm = tp.spatial_softmax2d(hm, 1.0)
        if len(shape) > 4:
            hm = hm.reshape(shape[0], -1, hm.shape[-2], hm.shape[-1])
        return hm


def training_step(model: OracleTracker, batch: dict, unsup_cfg: dict[str, dict] | None,
                  anneal_weight: float | None = 1.0) -> tuple[torch.Tensor, dict[str, torch.Tensor]]:
    """models/base.py:504-546,627-701: total loss and the logged scalars of one (semi-)supervised step.

    batch = {"labeled": {images, keypoints, heatmaps, bbox}, "unlabeled": {frames, transforms, bbox,
    is_multiview}} or just the labeled dict.  unsup_cfg maps loss name -> params:
      temporal       {log_weight, epsilon, prob_threshold}
      pca_singleview {log_weight, mean, kept_eigenvectors, epsilon, columns}
      pca_multiview  {log_weight, mean, kept_eigenvectors, epsilon, mirrored_column_matches}
      unimodal_mse   {log_weight, prob_threshold}
    """
    logs: dict[str, torch.Tensor] = {}
    labeled = batch["labeled"] if "labeled" in batch else batch
    semi = "unlabeled" in batch
    if semi:
        logs["total_unsupervised_importance"] = torch.tensor(float(anneal_weight))

    img = labeled["images"]
    mh, mw = img.shape[-2:]
    nviews = img.shape[1] if img.dim() == 5 else 1
    hm_pred = model(img)
    kp_pred, _conf = soft_argmax(hm_pred, model.downsample_factor, 1000.0)
    kp_pred = model_to_frame(kp_pred, mh, mw, labeled["bbox"], nviews)
    kp_targ = model_to_frame(labeled["keypoints"], mh, mw, labeled["bbox"], nviews)
    mse = heatmap_mse_loss(labeled["heatmaps"], hm_pred)
    loss_sup = factory_total({"heatmap_mse": (mse, 0.0)}, anneal_weight if semi else None)
    logs["train_supervised_loss"] = loss_sup
    logs["train_supervised_rmse"] = rmse_loss(kp_targ, kp_pred)
    logs["train_heatmap_mse_loss"] = mse
    logs["heatmap_mse_weight"] = torch.tensor(loss_weight(0.0))
    logs["train_heatmap_mse_loss_weighted"] = loss_weight(0.0) * mse
    if not semi:
        return loss_sup, logs

    un = batch["unlabeled"]
    fr = un["frames"]
    mh, mw = fr.shape[-2:]
    is_mv = bool(un.get("is_multiview", False))
    nviews = fr.shape[1] if fr.dim() == 5 else 1
    hm_u = model(fr)
    kp_aug, conf = soft_argmax(hm_u, model.downsample_factor, 1000.0)
    kp_u = undo_affine(kp_aug, un["transforms"], is_mv)
    kp_u = model_to_frame(kp_u, mh, mw, un["bbox"], nviews if is_mv else 1)
    vals: dict[str, tuple[torch.Tensor, float]] = {}
    for name, p in (unsup_cfg or {}).items():
        if name == "temporal":
            v = temporal_loss(kp_u, conf, p.get("epsilon", 0.0), p.get("prob_threshold", 0.0))
        elif name == "pca_singleview":
            d = pca_format_singleview(kp_u, p.get("columns"))
            v = pca_loss(d, torch.as_tensor(p["mean"]), torch.as_tensor(p["kept_eigenvectors"]), p["epsilon"])
        elif name == "pca_multiview":
            d = pca_format_multiview(kp_u, p["mirrored_column_matches"])
            v = pca_loss(d, torch.as_tensor(p["mean"]), torch.as_tensor(p["kept_eigenvectors"]), p["epsilon"])
        elif name == "unimodal_mse":
            v = unimodal_mse_loss(kp_aug, hm_u, conf, mh, mw, p.get("prob_threshold", 0.0))
        else:
            raise ValueError(name)
        lw = float(p.get("log_weight", 0.0))
        vals[name] = (v, lw)
        logs[f"train_{name}_loss"] = v
        logs[f"{name}_weight"] = torch.tensor(loss_weight(lw))
        logs[f"train_{name}_loss_weighted"] = loss_weight(lw) * v
    loss_unsup = factory_total(vals, anneal_weight)
    total = loss_sup + loss_unsup
    logs["total_loss"] = total
    return total, logs


# ======================================================================================
# bf16-mixed precision policy oracle
# ======================================================================================
# The reference trains in fp32 only.  The MI355X path stores activations and GEMM operands in bf16
# with fp32 accumulation, fp32 BatchNorm statistics, fp32 logits/softmax/losses and fp32 master weights (DESIGN.md
# "precision policy" - Lightning's `bf16-mixed` semantics).  `forward_bf16_policy` restates exactly that policy on top of
# the fp32 OracleTracker weights by rounding at the same points (forward values AND, through autograd, gradients),
# so the HIP engine can be checked against it to rounding-order accuracy; the fp32 forward stays the end-to-end oracle.


def _q(x: torch.Tensor) -> torch.Tensor:
    """round to bf16, keep fp32 dtype; autograd rounds the incoming gradient to bf16 at the same place"""
    return x.to(torch.bfloat16).float()


class _RoundGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).float()


class _PairRound(torch.autograd.Function):
    """A block output kept as a bf16 PAIR (the "fp32 residual stream" policy, lp_bn_apply_seg_lo): -> (hi, hi + lo) with hi = bf16(y), lo =
    bf16(y - hi).  The convolutions read hi, the next identity shortcut adds hi + lo.  Backward: the two gradients are summed and rounded to
    bf16 once, as the product's data-gradient store pass does (and as `_q` does for the single-word policy)."""

    @staticmethod
    def forward(ctx, y):
        hi = y.to(torch.bfloat16).float()
        lo = (y - hi).to(torch.bfloat16).float()
        return hi, hi + lo

    @staticmethod
    def backward(ctx, g_hi, g_carried):
        return (g_hi + g_carried).to(torch.bfloat16).float()


def _bn_train(z: torch.Tensor, bn: nn.BatchNorm2d, residual: torch.Tensor | None, relu: bool, q=None) -> torch.Tensor:
    y = F.batch_norm(z, bn.running_mean, bn.running_var, bn.weight, bn.bias, training=True, momentum=bn.momentum, eps=bn.eps)
    if residual is not None:
        y = y + residual
    if relu:
        y = F.relu(y)
    return (_q if q is None else q)(y)


def forward_bf16_policy(model: OracleTracker, images: torch.Tensor, rounding: tuple[str, ...] = ("trunk", "head")) -> torch.Tensor:
    """Training-mode forward of `model` under the bf16-mixed policy (updates BN running statistics like train()).

    ``rounding`` (profiles/rounding_ablation.py): which of the policy's rounding points are ON - "trunk" = every convolution operand and
    output of the ResNet trunk, "head" = the operands of the two transposed convolutions and the activation between them.  Both = the
    policy the product implements; () = the fp32 reference itself.  Finer tags switch on parts of "trunk" only (the per-stage ablation,
    profiles/r04_rounding_stages.json): a STAGE - "stem", "layer1" .. "layer4" = every rounding point inside it - or a KIND across all
    stages - "trunk:w" weights, "trunk:z" convolution outputs (the pre-normalisation tensors), "trunk:a" the activations inside a block
    (outputs of bn1 / bn2, the stem's pooled activation), "trunk:res" the residual stream (block outputs, shortcut projections).
    "res32" (with "trunk" / "trunk:res" on): the OPTIONAL policy of round 6 (Engine.residual_fp32, LP_RESIDUAL_FP32=1) - a block output is a
    bf16 pair, the identity shortcut of the next block adds hi + lo, a projection shortcut is added unrounded (`_PairRound`)."""
    ident = lambda t: t  # noqa: E731

    def qt(stage: str, kind: str):
        return _q if ("trunk" in rounding or stage in rounding or f"trunk:{kind}" in rounding) else ident

    qh = _q if "head" in rounding else ident
    bb = model.backbone
    x = qt("stem", "a")(images)
    x = qt("stem", "z")(F.conv2d(x, qt("stem", "w")(bb[0].weight), stride=2, padding=3))
    x = _bn_train(x, bb[1], None, True, q=qt("stem", "a"))
    x = F.max_pool2d(x, 3, 2, 1)
    carried = None   # ("res32": the previous block's output as hi + lo)
    for li, layer in enumerate((bb[4], bb[5], bb[6], bb[7])):
        st = f"layer{li + 1}"
        qw, qz, qa, qr = qt(st, "w"), qt(st, "z"), qt(st, "a"), qt(st, "res")
        pair = "res32" in rounding and qr is _q
        for blk in layer:
            idt = carried if (pair and carried is not None and blk.downsample is None) else x
            o = _bn_train(qz(F.conv2d(x, qw(blk.conv1.weight))), blk.bn1, None, True, q=qa)
            o = _bn_train(qz(F.conv2d(o, qw(blk.conv2.weight), stride=blk.stride, padding=1)), blk.bn2, None, True, q=qa)
            z3 = qz(F.conv2d(o, qw(blk.conv3.weight)))
            if blk.downsample is not None:
                zd = qz(F.conv2d(x, qw(blk.downsample[0].weight), stride=blk.stride))
                idt = _bn_train(zd, blk.downsample[1], None, False, q=ident if pair else qr)
            if pair:
                x, carried = _PairRound.apply(_bn_train(z3, blk.bn3, idt, True, q=ident))
            else:
                x = _bn_train(z3, blk.bn3, idt, True, q=qr)
    x = F.pixel_shuffle(x, 2)
    cts = [m for m in model.head.upsampling_layers if isinstance(m, nn.ConvTranspose2d)]
    for i, ct in enumerate(cts):
        x = F.conv_transpose2d(qh(x), qh(ct.weight), ct.bias, stride=2, padding=1, output_padding=1)
        if "head" in rounding:
            x = _RoundGrad.apply(x) if i == len(cts) - 1 else _q(x)
    return tp.spatial_softmax2d(x, 1.0)
