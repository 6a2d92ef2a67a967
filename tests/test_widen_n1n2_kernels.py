"""Kernel-logic tests of csrc/frames.hip (the batch producers either side of the step's inputs).

Image operators (DALI resize / warp_affine / brightness_contrast / noise.shot / crop_mirror_normalize): DALI is not vendored,
so the check is the oracle's restatement, whose samplers are torch's own (F.interpolate antialias, F.grid_sample) - parity with
DALI itself is UNPINNED.  Labeled keypoints / visibility / targets: golden of the verbatim HeatmapDataset
(tests/golden/labeled_targets.npz)."""

import numpy as np
import pytest
import torch

from lightning_pose_amd import _lib
from oracle import restated as O
from tests.hipemu import emu

pytestmark = pytest.mark.usefixtures("kernel_backend")


def _frames(seed, s, h, w):
    g = torch.Generator().manual_seed(seed)
    # smooth structure + noise, full u8 range
    ys, xs = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing="ij")
    base = 127 + 100 * torch.sin(xs / 5.0 + torch.arange(s).view(s, 1, 1)) * torch.cos(ys / 7.0)
    img = base.unsqueeze(-1) + 40 * torch.randn(s, h, w, 3, generator=g)
    return img.clamp(0, 255).to(torch.uint8)


@pytest.mark.parametrize("hs,ws,h,w", [(50, 70, 16, 24), (37, 41, 64, 64), (64, 64, 64, 64), (203, 198, 64, 48), (30, 200, 32, 40)])
def test_cubic_resize_matches_torch_bicubic(hs, ws, h, w):
    """the labeled images' resize (imgaug iaa.Resize default = OpenCV INTER_CUBIC): Keys A = -0.75, half-pixel centres, clamped taps, no
    antialiasing - the same definition torch's bicubic implements; then the uint8 rounding of the image imgaug hands back"""
    src = _frames(11, 2, hs, ws)
    raw = emu.frames_resize_cubic(src.numpy(), h, w, round_u8=0)
    np.testing.assert_allclose(raw, O.frames_resize_cubic(src, h, w, round_u8=False).numpy(), atol=2e-3)
    got = emu.frames_resize_cubic(src.numpy(), h, w, round_u8=1)
    want = O.frames_resize_cubic(src, h, w).numpy()
    assert got.min() >= 0 and got.max() <= 255 and np.array_equal(got, np.round(got))
    assert np.abs(got - want).max() <= 1 and (got != want).mean() < 2e-3     # fp32 vs fp64 at an exact .5: a level, rarely
    fin = emu.frames_resize_cubic(src.numpy(), h, w, round_u8=1, norm=emu.frame_norm())
    np.testing.assert_allclose(fin, O.frames_finish(torch.from_numpy(got)).numpy(), atol=1e-5)


@pytest.mark.parametrize("hs,ws,h,w", [(50, 70, 16, 24), (37, 41, 64, 64), (64, 64, 64, 64), (203, 198, 64, 48), (30, 200, 32, 40)])
def test_resize_renorm_matches_torch_antialias(hs, ws, h, w):
    src = _frames(1, 2, hs, ws)
    got = emu.frames_resize(src.numpy(), h, w, _lib.BORDER_RENORM)
    want = O.frames_resize(src, h, w, "renorm").numpy()
    np.testing.assert_allclose(got, want, atol=2e-3)  # on the [0, 255] scale; fp32 weights vs fp64


@pytest.mark.parametrize("hs,ws,h,w", [(50, 70, 16, 24), (37, 41, 64, 64), (101, 67, 32, 32)])
def test_resize_clamp_matches_restatement(hs, ws, h, w):
    src = _frames(2, 2, hs, ws)
    got = emu.frames_resize(src.numpy(), h, w, _lib.BORDER_CLAMP)
    np.testing.assert_allclose(got, O.frames_resize(src, h, w, "clamp").numpy(), atol=2e-3)
    if hs < h:  # up-scaling: two taps, both border rules coincide
        np.testing.assert_allclose(got, O.frames_resize(src, h, w, "renorm").numpy(), atol=2e-3)


def test_resize_finish_is_normalised_planes():
    """imgaug='default' path of video_pipe: resize -> /255 -> normalise -> FCHW in ONE launch"""
    src = _frames(3, 3, 90, 120)
    got = emu.frames_resize(src.numpy(), 32, 32, _lib.BORDER_CLAMP, norm=emu.frame_norm())
    want = O.frames_finish(O.frames_resize(src, 32, 32, "clamp")).numpy()
    assert got.shape == (3, 3, 32, 32)
    np.testing.assert_allclose(got, want, atol=5e-5)
    # constant image: any normalised filter reproduces the constant exactly
    const = np.full((1, 45, 77, 3), 200, np.uint8)
    flat = emu.frames_resize(const, 32, 48, _lib.BORDER_RENORM)
    np.testing.assert_allclose(flat, 200.0, atol=1e-4)


def _matrix(angle_deg, sx, sy, h, w):
    """fn.transforms.rotation(angle, center) then fn.transforms.scale(scale, center) as the reference composes them
    (data/video/dali.py:158-161): source -> destination, centre (h/2, w/2) passed as (x, y) = `size` (sic)."""
    th = np.deg2rad(angle_deg)
    cx, cy = h / 2, w / 2  # the reference passes center = (resize_dims[0] / 2, resize_dims[1] / 2)
    rot = np.array([[np.cos(th), np.sin(th), 0], [-np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    t0 = np.array([[1, 0, -cx], [0, 1, -cy], [0, 0, 1.0]])
    t1 = np.array([[1, 0, cx], [0, 1, cy], [0, 0, 1.0]])
    sc = np.diag([sx, sy, 1.0])
    return (t1 @ sc @ t0 @ t1 @ rot @ t0)[:2]


@pytest.mark.parametrize("angle,sx,sy", [(0.0, 1.0, 1.0), (7.5, 1.1, 0.9), (-10.0, 0.8, 1.2), (33.0, 0.5, 0.5)])
def test_augment_warp_matches_grid_sample(angle, sx, sy):
    h, w = 40, 56
    src = O.frames_resize(_frames(4, 2, h, w), h, w)  # fp32 HWC in [0, 255]
    m = _matrix(angle, sx, sy, h, w)
    got = emu.frames_augment(src.numpy(), matrix=m)
    want = O.frames_finish(O.frames_warp_affine(src, torch.from_numpy(m))).numpy()
    np.testing.assert_allclose(got, want, atol=2e-4)
    if angle == 0.0 and sx == 1.0:
        np.testing.assert_allclose(got, O.frames_finish(src).numpy(), atol=1e-5)  # identity matrix = no resampling


def test_augment_translation_fills_zero():
    """integer shift by (+5, -3): destination (x, y) shows source (x - 5, y + 3); uncovered pixels are fill_value = 0"""
    h, w = 24, 32
    src = O.frames_resize(_frames(5, 1, h, w), h, w)
    m = np.array([[1.0, 0.0, 5.0], [0.0, 1.0, -3.0]])
    got = emu.frames_augment(src.numpy(), matrix=m, norm=emu.frame_norm((0, 0, 0), (1, 1, 1))) * 255.0
    want = np.zeros((1, 3, h, w), np.float32)
    want[:, :, : h - 3, 5:] = src.permute(0, 3, 1, 2).numpy()[:, :, 3:, : w - 5]
    np.testing.assert_allclose(got, want, atol=1e-3)


def test_augment_brightness_contrast():
    h, w = 16, 64
    src = O.frames_resize(_frames(6, 2, h, w), h, w)
    got = emu.frames_augment(src.numpy(), brightness=1.2, contrast=0.8)
    want = O.frames_finish(O.brightness_contrast(src, 1.2, 0.8, 0.5)).numpy()
    np.testing.assert_allclose(got, want, atol=2e-5)
    got = emu.frames_augment(src.numpy(), brightness=0.75, contrast=1.25, contrast_center=128.0)
    np.testing.assert_allclose(got, O.frames_finish(O.brightness_contrast(src, 0.75, 1.25, 128.0)).numpy(), atol=2e-5)


def test_shot_noise_statistics_and_determinism():
    """out = Poisson(in / factor) * factor: mean in, variance in * factor; both sampler branches (lambda < 12 and above);
    the stream depends only on (seed, frame, pixel)"""
    h, w = 64, 128
    unit = emu.frame_norm((0, 0, 0), (1, 1, 1))
    for level, factor in ((40.0, 8.0), (200.0, 2.0), (6.0, 0.5), (90.0, 10.0)):
        src = np.full((2, h, w, 3), level, np.float32)
        a = emu.frames_augment(src, shot_factor=factor, seed=7, norm=unit) * 255.0
        n = a.size
        assert abs(a.mean() - level) < 5 * np.sqrt(level * factor / n) + 0.02 * factor, (level, factor, a.mean())
        assert a.var() == pytest.approx(level * factor, rel=0.05), (level, factor)
        assert np.allclose(a / factor, np.round(a / factor), atol=1e-3)  # integer counts times the factor
        b = emu.frames_augment(src, shot_factor=factor, seed=7, norm=unit) * 255.0
        np.testing.assert_array_equal(a, b)
        c = emu.frames_augment(src, shot_factor=factor, seed=8, norm=unit) * 255.0
        assert (a != c).mean() > 0.5
        assert abs(np.corrcoef(a[0].ravel(), a[1].ravel())[0, 1]) < 0.02  # frames draw independent noise
        assert abs(np.corrcoef(a[0, 0].ravel(), a[0, 1].ravel())[0, 1]) < 0.02  # channels too
    # factor 0 switches the noise off
    src = O.frames_resize(_frames(8, 1, 8, 64), 8, 64).numpy()
    np.testing.assert_allclose(emu.frames_augment(src, shot_factor=0.0, norm=unit) * 255.0, src.transpose(0, 3, 1, 2), atol=1e-4)


def test_argument_errors():
    lib = emu.lib()
    z = emu.Z((1, 4, 4, 3))
    u8 = emu.Buf(np.zeros((1, 4, 4, 3), np.uint8))
    assert lib.lp_frames_resize(None, 1, 4, 4, 48, 12, 4, 4, 0, None, z.p, emu.stream()) == -1
    assert lib.lp_frames_resize(u8.p, 1, 4, 4, 48, 11, 4, 4, 0, None, z.p, emu.stream()) == -1   # row stride < 3 * width
    assert lib.lp_frames_resize(u8.p, 1, 4, 4, 48, 12, 4, 4, 5, None, z.p, emu.stream()) == -1   # unknown border rule
    bad = _lib.FrameNorm((emu.C.c_float * 3)(0, 0, 0), (emu.C.c_float * 3)(1, 0, 1))
    assert lib.lp_frames_resize(u8.p, 1, 4, 4, 48, 12, 4, 4, 0, emu.C.byref(bad), z.p, emu.stream()) == -1  # std = 0
    sing = _lib.FrameAugment(1, (emu.C.c_float * 6)(1, 2, 0, 2, 4, 0), 1.0, 1.0, 0.5, 0.0, 0)
    nrm = emu.frame_norm()
    assert lib.lp_frames_augment(z.p, 1, 4, 4, emu.C.byref(sing), emu.C.byref(nrm), z.p, emu.stream()) == -1  # singular matrix


# ---------------------------------------------------------------------------------------------------- labeled producer (N2)
@pytest.mark.parametrize("tag,uniform", [("u0", False), ("u1", True)])
def test_labeled_keypoints_and_targets_match_verbatim_dataset(golden, tag, uniform):
    g = golden("labeled_targets")
    kp, vis = emu.labeled_keypoints(g["kp_src"], g["src_hw"], 256, 256, affine=g["affine"], uniform=uniform)
    np.testing.assert_array_equal(vis, g[f"{tag}_vis"])
    want = g[f"{tag}_kp_model_nan"]
    np.testing.assert_array_equal(np.isnan(kp), np.isnan(want))
    assert np.isnan(want).any() and (~np.isnan(want)).any()
    np.testing.assert_allclose(np.nan_to_num(kp), np.nan_to_num(want), atol=5e-5)
    hm = emu.heatmap_gen(kp, vis, 256, 256, 64, 64)
    np.testing.assert_allclose(hm, g[f"{tag}_heatmaps"], atol=2e-6)
    # three kinds of target as the reference builds them: Gaussians, all-zero (invisible / pushed out), uniform (uniform_heatmaps)
    sums = g[f"{tag}_heatmaps"].reshape(8, 17, -1).sum(-1)
    peak = g[f"{tag}_heatmaps"].reshape(8, 17, -1).max(-1)
    assert (sums == 0).any() and (np.abs(sums - 1) < 1e-4).any()
    assert (np.abs(peak - 1.0 / 4096) < 1e-9).any() == uniform


def test_oracle_labeled_keypoints_pinned_to_verbatim_dataset(golden):
    g = golden("labeled_targets")
    for tag, uniform in (("u0", False), ("u1", True)):
        kp, vis = O.labeled_keypoints(g.t("kp_src"), g.t("src_hw"), 256, 256, affine=g.t("affine"), uniform_heatmaps=uniform)
        np.testing.assert_array_equal(vis.numpy(), g[f"{tag}_vis"])
        np.testing.assert_allclose(np.nan_to_num(kp.numpy()), np.nan_to_num(g[f"{tag}_kp_model_nan"]), atol=5e-5)
        hm = O.generate_heatmaps(kp, 256, 256, (64, 64), 1.25, vis)
        np.testing.assert_allclose(hm.numpy(), g[f"{tag}_heatmaps"], atol=2e-6)


def test_labeled_keypoints_hflip_swap_and_explicit_visibility(golden):
    g = golden("labeled_targets")
    kp_src, src_hw = g.t("kp_src"), g.t("src_hw")
    swap = torch.tensor([3, 2, 1, 0, 4, 5, 6, 7, 11, 10, 9, 8, 12, 13, 14, 15, 16])  # left/right pairs of the mirror-mouse names
    flip = torch.tensor([1, 0, 1, 1, 0, 0, 1, 0])
    gen = torch.Generator().manual_seed(2)
    vis_in = torch.randint(0, 3, (8, 17), generator=gen)
    want_kp, want_vis = O.labeled_keypoints(kp_src, src_hw, 256, 256, hflip=flip, swap=swap, visibility=vis_in)
    kp, vis = emu.labeled_keypoints(kp_src.numpy(), src_hw.numpy(), 256, 256, hflip=flip.numpy(), swap=swap.numpy(), vis=vis_in.numpy())
    np.testing.assert_array_equal(vis, want_vis.numpy())
    np.testing.assert_array_equal(np.isnan(kp), np.isnan(want_kp.numpy()))
    np.testing.assert_allclose(np.nan_to_num(kp), np.nan_to_num(want_kp.numpy()), atol=5e-5)
    # un-flipped samples are just the resize projection
    np.testing.assert_allclose(np.nan_to_num(kp[1]), np.nan_to_num((kp_src[1] * torch.tensor([256 / 396, 256 / 406])).numpy()), atol=5e-5)


# ------------------------------------------------------------------------------------ TemporalHeatmapLoss (losses/losses.py:706-869)
@pytest.mark.parametrize("tag,kind,eps,thr", [("mse_plain", _lib.HM_MSE, 0.0, 0.0), ("kl_plain", _lib.HM_KL, 0.0, 0.0),
                                              ("mse_thr", _lib.HM_MSE, 0.0, 0.4), ("kl_thr_eps", _lib.HM_KL, 0.5, 0.4),
                                              ("mse_eps_list", _lib.HM_MSE, [0.0, 2e-5, 1e-4, 1.0], 0.2)])
def test_temporal_heatmap_loss_matches_verbatim_class(golden, tag, kind, eps, thr):
    g = golden("temporal_heatmap")
    loss, grad = emu.temporal_heatmap(kind, g["hm"], g["conf"], eps, thr, gout=0.7)
    assert loss == pytest.approx(float(g[f"{tag}_loss"]), rel=2e-5, abs=1e-9)
    scale = np.abs(g[f"{tag}_grad"]).max()
    np.testing.assert_allclose(grad, g[f"{tag}_grad"], atol=2e-5 * max(scale, 1e-12), rtol=2e-4)
    assert np.abs(g[f"{tag}_grad"]).max() > 0


def test_temporal_heatmap_single_frame_is_nan():
    hm = np.full((1, 2, 4, 4), 1 / 16, np.float32)
    loss, grad = emu.temporal_heatmap(_lib.HM_MSE, hm, np.ones((1, 2), np.float32), 0.0, 0.0)
    assert np.isnan(loss)  # the reference takes the mean of an empty (0, K) tensor
    assert not grad.any()
