"""Kernel-logic tests of csrc/decode.hip on the CPU emulator (tests/hipemu) against the oracle and the golden
vectors of the verbatim reference.  These run the SAME kernel source the GPU runs; the -m gpu tests repeat the
comparison on the real device through the product path."""

import numpy as np
import pytest
import torch

from oracle import restated as O
from tests.hipemu import emu

pytestmark = pytest.mark.usefixtures("kernel_backend")
from lightning_pose_amd import _lib


@pytest.mark.parametrize("tag,dss", [("a", (1, 2, 3)), ("b", (1, 2)), ("edge", (2,)), ("flat", (2,))])
def test_decode_fwd_matches_reference_golden(golden, tag, dss):
    g = golden("decode")
    heat = g[f"{tag}_in"]
    for ds in dss:
        kp_aug, kp_frame, conf, _ = emu.decode_fwd(heat, ds)
        want_kp = g[f"{tag}_kp_ds{ds}"].reshape(kp_aug.shape)
        np.testing.assert_allclose(kp_aug, want_kp, atol=1e-4, rtol=0)
        np.testing.assert_allclose(kp_frame, want_kp, atol=1e-4, rtol=0)  # identity frame map
        np.testing.assert_allclose(conf, g[f"{tag}_conf_ds{ds}"], atol=2e-5, rtol=0)


def test_decode_fwd_kat_delta():
    """reference tests/models/heads/test_heatmap.py:124-170: delta at (2,2),(4,4) on 8x8 -> val * 2^ds, conf 1."""
    x = np.zeros((1, 2, 8, 8), np.float32)
    x[0, 0, 2, 2] = 1
    x[0, 1, 4, 4] = 1
    for ds in (1, 2):
        kp, _, conf, _ = emu.decode_fwd(x, ds)
        np.testing.assert_allclose(kp.reshape(-1), np.array([2, 2, 4, 4]) * 2.0 ** ds, atol=1e-4)
        np.testing.assert_allclose(conf, 1.0, atol=1e-6)


def test_decode_fwd_wide_map_two_strips():
    """W = 4*80 = 320 > 256 exercises the second (partial) column strip."""
    g = torch.Generator().manual_seed(5)
    heat = torch.softmax(3 * torch.randn(1, 3, 20 * 80, generator=g), -1).reshape(1, 3, 20, 80)
    heat[0, 1] = 0
    heat[0, 1, 10, 75] = 1.0  # peak in the second strip
    kp, _, conf, _ = emu.decode_fwd(heat.numpy(), 2)
    wkp, wconf = O.soft_argmax(heat, 2, 1000.0)
    np.testing.assert_allclose(kp.reshape(1, -1), wkp.numpy(), atol=1e-4)
    np.testing.assert_allclose(conf, wconf.numpy(), atol=2e-5)


def test_decode_frame_map(golden):
    g = golden("geometry")
    gen = torch.Generator().manual_seed(6)
    s, k = 6, 4
    heat = torch.softmax(4 * torch.randn(s, k, 16 * 16, generator=gen), -1).reshape(s, k, 16, 16)
    kp_aug, _ = O.soft_argmax(heat, 2, 1000.0)
    bbox = g.t("bbox")
    for mode, tf in ((_lib.TF_SINGLE, g.t("A")), (_lib.TF_PER_FRAME, g.t("As")), (_lib.TF_NONE, None)):
        fm, keep = emu.frame_map(tf.numpy() if tf is not None else None, mode, bbox.numpy(), 1, k, 64, 64)
        _, kp_frame, _, _ = emu.decode_fwd(heat.numpy(), 2, fm=fm)
        want = O.model_to_frame(O.undo_affine(kp_aug, tf if tf is not None else torch.tensor([-1.0])), 64, 64, bbox)
        np.testing.assert_allclose(kp_frame.reshape(s, -1), want.numpy(), atol=2e-4)
    # multiview: 2 views x 2 keypoints, per-view transforms and bboxes
    fm, keep = emu.frame_map(g.t("As")[:2].numpy(), _lib.TF_PER_VIEW, g.t("bbox2").numpy(), 2, k, 64, 64)
    _, kp_frame, _, _ = emu.decode_fwd(heat.numpy(), 2, fm=fm)
    want = O.model_to_frame(O.undo_affine(kp_aug, g.t("As")[:2], True), 64, 64, g.t("bbox2"), 2)
    np.testing.assert_allclose(kp_frame.reshape(s, -1), want.numpy(), atol=2e-4)


@pytest.mark.parametrize("shape,ds", [((2, 3, 16, 16), 2), ((1, 2, 12, 20), 1), ((1, 2, 16, 16), 3), ((1, 2, 20, 80), 2)])
def test_decode_bwd_matches_autograd(shape, ds):
    gen = torch.Generator().manual_seed(7)
    b, k, h, w = shape
    heat = torch.softmax(2 * torch.randn(b, k, h * w, generator=gen), -1).reshape(shape).requires_grad_(True)
    kp, _ = O.soft_argmax(heat, ds, 1000.0)
    gk = torch.randn(kp.shape, generator=gen)
    (kp * gk).sum().backward()
    _, _, _, stats = emu.decode_fwd(heat.detach().numpy(), ds)
    g_heat = emu.decode_bwd(heat.detach().numpy(), ds, stats, g_aug=gk.reshape(b, k, 2).numpy())
    ref = heat.grad.numpy()
    scale = np.abs(ref).max()
    np.testing.assert_allclose(g_heat, ref, atol=2e-3 * scale, rtol=0)


@pytest.mark.parametrize("shape,ds", [((1, 2, 65, 65), 1), ((1, 2, 64, 64), 2), ((1, 3, 48, 48), 2), ((2, 17, 96, 96), 2), ((1, 2, 40, 56), 3)])
def test_decode_bwd_strip_without_atomics_vs_fp64_autograd(kernel_backend, shape, ds):
    """Round 6: the backward's LDS strip is written by owner stores and guest adds instead of LDS atomics (csrc/decode.hip).  Maps tall enough
    for several waves - 96 x 96 is BASELINE's; 65 x 65 at downsample_factor 1 is the one size up to 256 where THREE waves meet in a strip row
    (_tables.axis_tables checks the bound) - against autograd in fp64 through the restated reference decode; with the moments taken about the
    tile's maximum the kernel is closer to the exact gradient than torch's own fp32, and it repeats bit for bit."""
    if kernel_backend == "emu" and shape[1] * shape[2] * shape[3] > 20000:
        shape = (1, 2) + shape[2:]   # (the emulator runs work-items as fibers: two maps of the full size are enough there)
    gen = torch.Generator().manual_seed(sum(shape) + ds)
    b, k, h, w = shape
    heat = torch.softmax(3 * torch.randn(b, k, h * w, generator=gen), -1).reshape(shape)
    h64 = heat.double().requires_grad_(True)
    kp, _ = O.soft_argmax(h64, ds, 1000.0)
    gk = torch.randn(kp.shape, generator=gen)
    (kp * gk.double()).sum().backward()
    _, _, _, stats = emu.decode_fwd(heat.numpy(), ds)
    outs = [emu.decode_bwd(heat.numpy(), ds, stats, g_aug=gk.reshape(b, k, 2).numpy()) for _ in range(2)]
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))
    ref = h64.grad.numpy()
    assert np.abs(outs[0] - ref).max() <= 5e-5 * np.abs(ref).max(), np.abs(outs[0] - ref).max() / np.abs(ref).max()


def test_decode_bwd_through_frame_map(golden):
    g = golden("geometry")
    gen = torch.Generator().manual_seed(8)
    s, k = 6, 4
    heat = torch.softmax(2 * torch.randn(s, k, 16 * 16, generator=gen), -1).reshape(s, k, 16, 16).requires_grad_(True)
    kp_aug, _ = O.soft_argmax(heat, 2, 1000.0)
    kp_fr = O.model_to_frame(O.undo_affine(kp_aug, g.t("As")), 64, 64, g.t("bbox"))
    ga, gf = torch.randn(kp_aug.shape, generator=gen), torch.randn(kp_fr.shape, generator=gen)
    ((kp_aug * ga).sum() + (kp_fr * gf).sum()).backward()
    fm, keep = emu.frame_map(g["As"], _lib.TF_PER_FRAME, g["bbox"], 1, k, 64, 64)
    _, _, _, stats = emu.decode_fwd(heat.detach().numpy(), 2, fm=fm)
    g_heat = emu.decode_bwd(heat.detach().numpy(), 2, stats, g_aug=ga.reshape(s, k, 2).numpy(),
                            g_frame=gf.reshape(s, k, 2).numpy(), fm=fm)
    ref = heat.grad.numpy()
    np.testing.assert_allclose(g_heat, ref, atol=2e-3 * np.abs(ref).max(), rtol=0)


def _peaked(g, b, k, h, w, sigma=1.25, amp=1.0):
    """maps of a trained network: a normalised Gaussian of the target's width at a random sub-pixel position (+ a little noise)"""
    ys = torch.arange(h).view(1, 1, h, 1).float()
    xs = torch.arange(w).view(1, 1, 1, w).float()
    cx = torch.rand(b, k, 1, 1, generator=g) * (w - 1)
    cy = torch.rand(b, k, 1, 1, generator=g) * (h - 1)
    m = torch.exp(-((xs - cx) ** 2 + (ys - cy) ** 2) / (2 * sigma ** 2))
    m = m / m.sum(dim=(2, 3), keepdim=True) * amp
    return m + 1e-5 * torch.rand(b, k, h, w, generator=g)


@pytest.mark.parametrize("ds,h,w", [(2, 24, 32), (2, 96, 96), (1, 16, 16), (3, 12, 12)])
def test_exact_pruning_changes_nothing(kernel_backend, ds, h, w):
    """At T = 1000 every term more than 104 / T below the peak is exactly 0 in fp32: the kernels skip the row groups / column waves /
    strips whose bound says so (csrc/decode.hip, "exact pruning").  Peaked maps (a trained network's), incl. peaks on the border and a
    two-peak map: forward outputs and the backward gradient equal the unpruned kernels' to fp32 rounding."""
    gen = torch.Generator().manual_seed(ds * 100 + h)
    x = _peaked(gen, 2, 5, h, w)
    x[0, 0] = 0.5 * (_peaked(gen, 1, 1, h, w)[0, 0] + _peaked(gen, 1, 1, h, w)[0, 0])      # two peaks of equal height
    x[1, 1] = 0.0
    x[1, 1, 0, 0] = 0.3                                                                        # a single pixel in the corner
    x[1, 2] = 0.0
    x[1, 2, h - 1, w // 2] = 0.2
    hm = x.numpy()
    res = {}
    for flag in ("0", "1"):            # the `prune` argument of lp_decode_fwd / lp_decode_bwd (a call argument since round 5: no process state)
        kp_aug, kp_frame, conf, stats = emu.decode_fwd(hm, ds, prune=int(flag))
        g = emu.decode_bwd(hm, ds, stats, g_aug=np.ones((2, 5, 2), np.float32) * np.array([1.0, -0.5], np.float32), prune=int(flag))
        res[flag] = (kp_aug, conf, stats, g)
    np.testing.assert_allclose(res["1"][0], res["0"][0], atol=2e-5, rtol=0)     # keypoints, px
    np.testing.assert_allclose(res["1"][1], res["0"][1], atol=1e-6, rtol=1e-5)  # confidences
    np.testing.assert_allclose(res["1"][2][..., :2], res["0"][2][..., :2], rtol=1e-6)  # max and sum of exponentials
    scale = float(np.abs(res["0"][3]).max())
    np.testing.assert_allclose(res["1"][3], res["0"][3], atol=2e-6 * scale, rtol=0)
    # and the pruned result is still the reference's (oracle restatement of run_subpixelmaxima)
    want_kp, want_conf = O.soft_argmax(torch.from_numpy(hm), ds, 1000.0)
    np.testing.assert_allclose(res["1"][0].reshape(2, -1), want_kp.numpy(), atol=2e-4)
    np.testing.assert_allclose(res["1"][1], want_conf.numpy(), atol=2e-5)


def test_decode_pruning_is_chosen_from_the_maps(stack_backend, monkeypatch):
    """ops.decode watches its own sum-of-exponentials output and switches to the pruned kernels once most maps are peaked (and back on flat
    maps); LP_DECODE_PRUNE=0 / 1 pins the choice.  Keypoints do not depend on which kernels ran."""
    from lightning_pose_amd import ops

    dev = stack_backend
    monkeypatch.delenv("LP_DECODE_PRUNE", raising=False)
    auto = ops._DecodePruneAuto()                           # an owner's own chooser (a tracker holds one; ops.decode's default is shared)
    gen = torch.Generator().manual_seed(3)
    peaked = _peaked(gen, 2, 4, 24, 24).to(dev)
    flat = torch.softmax(torch.randn(2, 4, 24 * 24, generator=gen) * 0.1, -1).reshape(2, 4, 24, 24).to(dev)
    fm = ops.DecodeFrameMap(None, False, None, 1, 96, 96, 4)
    first = ops.decode(peaked, 2, 1000.0, fm, auto)[0].clone()
    assert auto.state == 0                                  # nothing observed yet: the plain kernels
    for _ in range(3):
        out = ops.decode(peaked, 2, 1000.0, fm, auto)[0]
        if dev.type == "cuda":
            torch.cuda.synchronize()
    assert auto.state == 1                                  # observed at the 2nd call, picked up by a later one
    torch.testing.assert_close(out, first, atol=2e-5, rtol=0)
    other = ops._DecodePruneAuto()                          # a second owner (another model) starts from the plain kernels and is
    ops.decode(flat, 2, 1000.0, fm, other)                  # not steered by the first one's peaked maps ...
    assert auto.want == 1 and other.want == 0 and other.state == 0
    ops.decode(peaked, 2, 1000.0, fm, auto)                 # ... and the first one gets its own choice back on its next call
    assert auto.state == 1
    auto.calls = auto.PERIOD - 1
    for _ in range(3):
        ops.decode(flat, 2, 1000.0, fm, auto)
        if dev.type == "cuda":
            torch.cuda.synchronize()
    assert auto.state == 0 and auto.want == 0
    monkeypatch.setenv("LP_DECODE_PRUNE", "1")
    ops.decode(flat, 2, 1000.0, fm, auto)
    assert auto.state == 1 and auto.want == 0               # pinned by the environment (the owner's own choice is kept for later)
    monkeypatch.setenv("LP_DECODE_PRUNE", "yes")
    with pytest.raises(ValueError):                         # auto / 0 / 1 only: a typo is an error, not a silent default
        ops.decode(flat, 2, 1000.0, fm, auto)
    monkeypatch.delenv("LP_DECODE_PRUNE")
