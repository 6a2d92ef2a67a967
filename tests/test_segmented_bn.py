"""Two BatchNorm segments in ONE launch (lp_bn_fuse.seg_images): the labeled and the unlabeled frames of a semi-supervised step go through
every layer together but keep their own batch statistics, as the reference's two forward calls do (models/base.py:682-695).  Kernel level:
the fused reductions of a joint launch equal those of one launch per segment.  Engine level: a joint pass equals two separate passes."""

import numpy as np
import pytest
import torch

from tests.hipemu import emu

bf = lambda t: t.to(torch.bfloat16).float()  # noqa: E731

SEG_CASES = [
    # B, seg, Hi, Wi, Ci, Co, R, stride, pad      (seg * rows per image must be a multiple of the 128-row tile in every launch)
    (4, 2, 8, 8, 64, 64, 1, 1, 0),        # 1x1: 64 rows per image, boundary at row 128; N = 64 tiles
    (6, 4, 8, 8, 64, 192, 3, 1, 1),       # 3x3, unequal segments (256 + 128 rows), ragged N
    (4, 2, 16, 16, 64, 128, 3, 2, 1),     # stride 2: forward 64 rows / image; data gradient = 4 parity classes of 64 rows / image
    (3, 1, 32, 16, 128, 64, 1, 2, 0),     # 1x1 stride 2 (projection shortcut): 3 of the 4 parity classes have no tap
]


@pytest.mark.parametrize("case", SEG_CASES)
def test_fused_reductions_per_segment(case, kernel_backend):
    B, seg, Hi, Wi, Ci, Co, R, st, pad = case
    gen = torch.Generator().manual_seed(11 + sum(case))
    g = emu.geom(B, Hi, Wi, Ci, Co, R, R, st, pad)
    x = emu.to_bf16_bits(torch.randn(B, Hi, Wi, Ci, generator=gen))
    w = torch.randn(Co, R, R, Ci, generator=gen) / (Ci * R * R) ** 0.5
    wg, wd = emu.to_bf16_bits(w), emu.to_bf16_bits(w.permute(3, 1, 2, 0))
    # ---- forward: one joint launch vs one launch per segment
    z, sums = emu.conv_fwd_bn(x, wg, g, seg=seg)
    assert sums.shape == (2, 2, Co)
    rows_o = g.Ho * g.Wo
    for si, (i0, n) in enumerate(((0, seg), (seg, B - seg))):
        gs = emu.geom(n, Hi, Wi, Ci, Co, R, R, st, pad)
        zs, ss = emu.conv_fwd_bn(x[i0:i0 + n], wg, gs)
        assert np.array_equal(z[i0 * rows_o:(i0 + n) * rows_o], zs)
        np.testing.assert_allclose(sums[si], ss, rtol=1e-5, atol=1e-4)
    # ---- backward: dx = gradient of relu(BN_seg(zin) + residual), each segment normalised with ITS moments
    rows_i = Hi * Wi
    Mi = B * rows_i
    zin = torch.randn(Mi, Ci, generator=gen)
    zin[seg * rows_i:] = zin[seg * rows_i:] * 1.7 + 0.4          # visibly different statistics per segment
    zin_bits = emu.to_bf16_bits(zin)
    gamma, beta = (torch.rand(Ci, generator=gen) + 0.5).numpy(), (torch.randn(Ci, generator=gen) * 0.3).numpy()
    mean, invstd = np.zeros((2, Ci), np.float32), np.zeros((2, Ci), np.float32)
    for si, (i0, n) in enumerate(((0, seg), (seg, B - seg))):
        _, mean[si], invstd[si] = emu.bn_forward(zin_bits[i0 * rows_i:(i0 + n) * rows_i], n * rows_i, Ci, gamma, beta, relu=True)
    dy = emu.to_bf16_bits(torch.randn(B * rows_o, Co, generator=gen))
    add = emu.to_bf16_bits(torch.randn(Mi, Ci, generator=gen))
    dx, bsums, dbeta, dgamma = emu.conv_dgrad_bn(dy, wd, g, zin_bits, mean, invstd, gamma, beta, addend_bits=add, seg=seg)
    assert bsums.shape == (2, 2, Ci)
    tot = np.zeros((2, Ci), np.float64)
    for si, (i0, n) in enumerate(((0, seg), (seg, B - seg))):
        gs = emu.geom(n, Hi, Wi, Ci, Co, R, R, st, pad)
        dxs, ss, _, _ = emu.conv_dgrad_bn(dy[i0 * rows_o:(i0 + n) * rows_o], wd, gs, zin_bits[i0 * rows_i:(i0 + n) * rows_i], mean[si], invstd[si],
                                          gamma, beta, addend_bits=add[i0 * rows_i:(i0 + n) * rows_i])
        assert np.array_equal(dx[i0 * rows_i:(i0 + n) * rows_i], dxs)
        np.testing.assert_allclose(bsums[si], ss, rtol=1e-4, atol=2e-3)
        tot += ss
    np.testing.assert_allclose(dbeta, tot[0], rtol=1e-4, atol=2e-3)      # d beta / d gamma collect both segments
    np.testing.assert_allclose(dgamma, tot[1], rtol=1e-4, atol=2e-3)


def test_stem_statistics_per_segment(kernel_backend):
    """the 7x7 stem's fused sums (conv_stem2d_kernel / conv_igemm_kernel<64, stem>), per segment"""
    gen = torch.Generator().manual_seed(5)
    B, seg, H = 3, 1, 32                                          # 16 x 16 = 256 output rows per image
    g = emu.geom(B, H, H, 4, 64, 7, 7, 2, 3)
    x4 = torch.randn(B, H, H, 4, generator=gen)
    x4[..., 3] = 0
    x4[seg:] += 0.5
    w = torch.zeros(64, 8, 8, 4)
    w[:, :7, :7, :3] = torch.randn(64, 7, 7, 3, generator=gen) / 12
    xb, wb = emu.to_bf16_bits(x4), emu.to_bf16_bits(w)
    z, sums = emu.stem_fwd_bn(xb, wb, g, seg=seg)
    for si, (i0, n) in enumerate(((0, seg), (seg, B - seg))):
        zs, ss = emu.stem_fwd_bn(xb[i0:i0 + n], wb, emu.geom(n, H, H, 4, 64, 7, 7, 2, 3))
        assert np.array_equal(z[i0 * 256:(i0 + n) * 256], zs)
        np.testing.assert_allclose(sums[si], ss, rtol=1e-5, atol=1e-4)


def _sums_case(kernel_backend, gen):
    """a launch with several tiles per workgroup AND several workgroups (LP_CONV_MAX_WGS is read per call)"""
    B, H, Ci, Co = (32, 32, 64, 256) if kernel_backend == "gpu" else (4, 16, 64, 128)
    g = emu.geom(B, H, H, Ci, Co, 1, 1, 1, 0)
    x = emu.to_bf16_bits(torch.randn(B, H, H, Ci, generator=gen))
    w = emu.to_bf16_bits(torch.randn(Co, 1, 1, Ci, generator=gen) / 8)
    return g, x, w, B, H, Co


def test_fused_sums_repeat_bit_for_bit(kernel_backend, monkeypatch):
    """Round 4: the store passes add their sums into the totals in FIXED POINT with 64-bit integer atomics (lp_fxsum; lp_common.h: fx_add).
    Integer addition commutes, so the totals do not depend on the order workgroups arrive in: repeated launches give the same BITS (the
    fp32 atomics of rounds 2 - 3 did not on the device), and lp_bn_finalize2 turns them into the moments of the stored values."""
    gen = torch.Generator().manual_seed(11)
    g, x, w, B, H, Co = _sums_case(kernel_backend, gen)
    if kernel_backend != "gpu":
        monkeypatch.setenv("LP_CONV_MAX_WGS", "3")            # 4 tiles of 256 rows over 3 workgroups: one of them walks two
    seg = B // 4                                                  # two BatchNorm segments, boundary on a tile boundary
    runs = [emu.conv_fwd_bn(x, w, g, seg=seg, raw=True) for _ in range(6 if kernel_backend == "gpu" else 2)]
    for z, words in runs[1:]:
        assert np.array_equal(z, runs[0][0]) and np.array_equal(words, runs[0][1])
    z1, words = runs[0]
    s1 = emu.fx(words)
    zf = emu.from_bf16_bits(z1).reshape(B, H * H, Co).double()
    np.testing.assert_allclose(s1[0][0], zf[:seg].sum((0, 1)).numpy(), rtol=2e-5, atol=2e-3)
    np.testing.assert_allclose(s1[1][1], (zf[seg:] ** 2).sum((0, 1)).numpy(), rtol=2e-5, atol=2e-3)
    counts = (seg * H * H, (B - seg) * H * H)
    rm0, rv0 = torch.randn(Co, generator=gen).numpy(), (torch.rand(Co, generator=gen) + 0.5).numpy()
    sb, m2, v2, rm2, rv2 = emu.Buf(words), emu.Z((2, Co)), emu.Z((2, Co)), emu.Buf(rm0), emu.Buf(rv0)
    emu.ok(emu.lib().lp_bn_finalize2(sb.p, float(counts[0]), float(counts[1]), Co, 1e-5, 0.1, m2.p, v2.p, rm2.p, rv2.p, emu.stream()))
    np.testing.assert_allclose(m2.np()[0], zf[:seg].mean((0, 1)).numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(v2.np()[1], (zf[seg:].var((0, 1), unbiased=False) + 1e-5).rsqrt().numpy(), rtol=1e-4)


def test_fixed_point_sums_keep_small_and_large_terms(kernel_backend):
    """lp_fxsum: hi counts 2^-12, lo counts 2^-60 - a column of 1e-6s next to a column of 3e4s, and a column whose terms cancel exactly"""
    M, Cn = 4096, 8
    x = torch.zeros(M, Cn)
    x[:, 0] = 1e-6
    x[:, 1] = 3e4
    x[:, 2] = torch.where(torch.arange(M) % 2 == 0, 257.0, -257.0)
    x[:, 3] = torch.randn(M, generator=torch.Generator().manual_seed(0))
    xb = emu.to_bf16_bits(x)
    xd = emu.from_bf16_bits(xb).double()
    words = emu.bn_stats(xb, M, Cn, raw=True)
    got = emu.fx(words)
    np.testing.assert_allclose(got[0], xd.sum(0).numpy(), rtol=3e-6, atol=1e-12)
    np.testing.assert_allclose(got[1], (xd * xd).sum(0).numpy(), rtol=3e-6, atol=1e-18)
    assert got[0][2] == 0.0 and np.array_equal(words[0, 2], [0, 0])
    assert np.array_equal(emu.bn_stats(xb, M, Cn, raw=True), words)


def test_non_finite_partials_poison_the_fixed_point_sum(kernel_backend):
    """a NaN / inf reaching a BatchNorm sum must stay visible (the fp32 atomics of rounds 2 - 3 propagated it; fmin / fmax in the
    first fx_add turned it into -2^49): the sum is poisoned, the moments and the running statistics of THAT channel read NaN, the others
    are untouched - through the stand-alone reduction and through a convolution's fused sums."""
    M, Cn = 1024, 16
    x = torch.randn(M, Cn, generator=torch.Generator().manual_seed(4))
    x[17, 1] = float("nan")
    x[500, 2] = float("inf")
    run = [np.zeros(Cn, np.float32), np.ones(Cn, np.float32)]
    _, mean, invstd = emu.bn_forward(emu.to_bf16_bits(x), M, Cn, np.ones(Cn, np.float32), np.zeros(Cn, np.float32), running=run)
    bad = np.zeros(Cn, bool)
    bad[[1, 2]] = True
    assert np.isnan(mean[bad]).all() and np.isfinite(mean[~bad]).all() and np.isfinite(invstd[~bad]).all()
    assert np.isnan(run[0][bad]).all() and np.isfinite(run[0][~bad]).all()
    np.testing.assert_allclose(mean[~bad], emu.from_bf16_bits(emu.to_bf16_bits(x)).double().mean(0).numpy()[~bad], atol=1e-6)
    # a convolution whose input holds a NaN: every output channel of that pixel is NaN, so every fused sum is poisoned
    g = emu.geom(1, 16, 16, 64, 64, 1, 1, 1, 0)
    xin = torch.randn(1, 16, 16, 64, generator=torch.Generator().manual_seed(5))
    xin[0, 3, 3, 7] = float("nan")
    w = emu.to_bf16_bits(torch.randn(64, 1, 1, 64, generator=torch.Generator().manual_seed(6)) / 8)
    _, words = emu.conv_fwd_bn(emu.to_bf16_bits(xin), w, g, raw=True)
    assert (np.abs(words[..., 1].astype(np.float64)) >= 2.0 ** 61).all()   # the `lo` words carry the poison


@pytest.mark.parametrize("M,Cn,seg", [(640, 64, 0), (1024, 256, 384), (257, 16, 0)])
def test_block_output_as_a_bf16_pair(kernel_backend, M, Cn, seg):
    """lp_bn_apply_seg_lo (round 6: the optional "fp32 residual stream" policy).  y is what lp_bn_apply_seg would store when the residual it
    adds is hi + lo; y + y_lo reproduces the fp32 result to 2^-17 of it; without the lo words in and out it IS lp_bn_apply_seg; with a
    projection shortcut (zd) the shortcut is added unrounded."""
    gen = torch.Generator().manual_seed(M + Cn)
    x = torch.randn(M, Cn, generator=gen) * 1.5 + 0.2
    res = torch.randn(M, Cn, generator=gen) * 2.0
    xb = emu.to_bf16_bits(x)
    rhi = emu.to_bf16_bits(res)
    rlo = emu.to_bf16_bits(res - emu.from_bf16_bits(rhi))
    gamma, beta = (torch.rand(Cn, generator=gen) + 0.5), torch.randn(Cn, generator=gen)
    nseg = 2 if seg else 1
    bounds = [(0, M)] if not seg else [(0, seg), (seg, M - seg)]
    xf = emu.from_bf16_bits(xb).reshape(M, Cn)
    mean = torch.cat([xf[r0:r0 + n].mean(0) for r0, n in bounds])
    invstd = torch.cat([(xf[r0:r0 + n].var(0, unbiased=False) + 1e-5).rsqrt() for r0, n in bounds])
    y, ylo, bits = emu.bn_apply_lo(xb, mean.numpy(), invstd.numpy(), gamma.numpy(), beta.numpy(), M, Cn, residual_bits=rhi, residual_lo_bits=rlo, seg_rows=seg)
    want = torch.empty(M, Cn)
    for si, (r0, n) in enumerate(bounds):
        mu, iv = mean[si * Cn:(si + 1) * Cn], invstd[si * Cn:(si + 1) * Cn]
        want[r0:r0 + n] = torch.relu((xf[r0:r0 + n] - mu) * (iv * gamma) + beta + (emu.from_bf16_bits(rhi) + emu.from_bf16_bits(rlo)).reshape(M, Cn)[r0:r0 + n])
    got_hi, got = emu.from_bf16_bits(y).reshape(M, Cn), emu.from_bf16_bits(y).reshape(M, Cn) + emu.from_bf16_bits(ylo).reshape(M, Cn)
    assert float((got_hi - want).abs().max()) <= 2.0 ** -8 * float(want.abs().max()) + 1e-6       # hi alone: one bf16 rounding
    assert float((got - want).abs().max()) <= 2.0 ** -16 * float(want.abs().max()) + 1e-6         # the pair: 16 mantissa bits
    assert np.array_equal(np.unpackbits(bits, bitorder="little").reshape(M, Cn).astype(bool), (got_hi > 0).numpy())
    # no lo in, no lo out: lp_bn_apply_seg itself
    y1, ylo1, b1 = emu.bn_apply_lo(xb, mean.numpy(), invstd.numpy(), gamma.numpy(), beta.numpy(), M, Cn, residual_bits=rhi, want_lo=False, seg_rows=seg)
    yb = emu.Z((M, Cn), np.uint16)
    bb = emu.Z(M * Cn // 8, np.uint8)
    mb, vb, gb, beb = [emu.Buf(emu.f32(a)) for a in (mean.numpy(), invstd.numpy(), gamma.numpy(), beta.numpy())]
    xbuf, rbuf = emu.Buf(xb), emu.Buf(rhi)
    emu.ok(emu.lib().lp_bn_apply_seg(xbuf.p, mb.p, vb.p, gb.p, beb.p, rbuf.p, 1, M, Cn, seg, yb.p, bb.p, emu.stream()))
    assert ylo1 is None and np.array_equal(y1, yb.np()) and np.array_equal(b1, bb.np())
    # projection shortcut normalised in the pass and added UNROUNDED
    zd = torch.randn(M, Cn, generator=gen) * 3.0
    zb = emu.to_bf16_bits(zd)
    zf = emu.from_bf16_bits(zb).reshape(M, Cn)
    md = torch.cat([zf[r0:r0 + n].mean(0) for r0, n in bounds])
    vd = torch.cat([(zf[r0:r0 + n].var(0, unbiased=False) + 1e-5).rsqrt() for r0, n in bounds])
    gd, bd = (torch.rand(Cn, generator=gen) + 0.5), torch.randn(Cn, generator=gen)
    y2, ylo2, _ = emu.bn_apply_lo(xb, mean.numpy(), invstd.numpy(), gamma.numpy(), beta.numpy(), M, Cn, zd_bits=zb, dbn=(md.numpy(), vd.numpy(), gd.numpy(), bd.numpy()),
                                  seg_rows=seg)
    want2 = torch.empty(M, Cn)
    for si, (r0, n) in enumerate(bounds):
        mu, iv = mean[si * Cn:(si + 1) * Cn], invstd[si * Cn:(si + 1) * Cn]
        sh = (zf[r0:r0 + n] - md[si * Cn:(si + 1) * Cn]) * (vd[si * Cn:(si + 1) * Cn] * gd) + bd
        want2[r0:r0 + n] = torch.relu((xf[r0:r0 + n] - mu) * (iv * gamma) + beta + sh)
    got2 = emu.from_bf16_bits(y2).reshape(M, Cn) + emu.from_bf16_bits(ylo2).reshape(M, Cn)
    assert float((got2 - want2).abs().max()) <= 2.0 ** -16 * float(want2.abs().max()) + 2e-6


@pytest.mark.parametrize("world", [2, 4, 8, 15])
def test_poison_survives_the_sum_over_ranks(kernel_backend, world):
    """SyncBatchNorm ADDS the ranks' words modulo 2^64 (engine._sync_stats: all-reduce, or all-gather + torch.sum).  Round 5's
    poison 2^62 wrapped to 0 at exactly 4 and 8 poisoned ranks - the moments read finite garbage.  With 5 * 2^60 the sum of any number of
    poisoned ranks (1 .. world <= 15), next to the other ranks' finite words, still reads NaN in the poisoned channel and only there."""
    M, Cn = 512, 16
    gen = torch.Generator().manual_seed(40 + world)
    clean = [torch.randn(M, Cn, generator=gen) for _ in range(world)]
    finite = [emu.bn_stats(emu.to_bf16_bits(x), M, Cn, raw=True) for x in clean]
    bad_x = []
    for x in clean:
        y = x.clone()
        y[3, 5] = float("nan")
        bad_x.append(y)
    poisoned = [emu.bn_stats(emu.to_bf16_bits(x), M, Cn, raw=True) for x in bad_x]
    want = np.mean([emu.from_bf16_bits(emu.to_bf16_bits(x)).double().mean(0).numpy() for x in clean], axis=0)
    for k in range(1, world + 1):   # k ranks diverged, world - k did not
        msg = np.zeros_like(finite[0])
        with np.errstate(over="ignore"):
            for r in range(world):
                msg = msg + (poisoned[r] if r < k else finite[r])   # int64: wraps like the collective's SUM
        mean, invstd = emu.bn_finalize_words(msg, M * world)
        assert np.isnan(mean[5]) and np.isnan(invstd[5]), (world, k, msg[:, 5])
        ok_ch = np.arange(Cn) != 5
        assert np.isfinite(mean[ok_ch]).all() and np.isfinite(invstd[ok_ch]).all()
        np.testing.assert_allclose(mean[ok_ch], want[ok_ch], atol=1e-6)


def test_standalone_reductions_repeat_bit_for_bit(kernel_backend):
    """lp_bn_stats / lp_bn_bwd_reduce (fixed-point sums): same bits twice; lp_bn_bwd_apply adds the sums into d beta / d gamma on top of
    what is already there; sums = NULL (eval-mode BatchNorm) drops the batch-statistics terms: dx = dy * gamma * invstd"""
    gen = torch.Generator().manual_seed(12)
    M, Cn = (200_000, 256) if kernel_backend == "gpu" else (3000, 40)
    x = torch.randn(M, Cn, generator=gen)
    dy = torch.randn(M, Cn, generator=gen)
    xb, db = emu.to_bf16_bits(x), emu.to_bf16_bits(dy)
    mean, invstd = x.mean(0).numpy(), (x.var(0, unbiased=False) + 1e-5).rsqrt().numpy()
    gamma = (torch.rand(Cn, generator=gen) + 0.5).numpy()
    outs = [emu.bn_backward(db, None, xb, mean, invstd, gamma, M, Cn) for _ in range(2)]
    for a, b in zip(outs[0], outs[1]):
        if a is not None:
            assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b)
    dyf, xf = emu.from_bf16_bits(db).double(), emu.from_bf16_bits(xb).double()
    xhat = (xf - torch.from_numpy(mean).double()) * torch.from_numpy(invstd).double()
    np.testing.assert_allclose(outs[0][3], dyf.sum(0).numpy(), rtol=1e-5, atol=1e-3)                 # d beta
    np.testing.assert_allclose(outs[0][2], (dyf * xhat).sum(0).numpy(), rtol=1e-4, atol=5e-3)        # d gamma
    base = (np.full(Cn, 2.0, np.float32), np.full(Cn, -1.0, np.float32))
    acc = emu.bn_backward(db, None, xb, mean, invstd, gamma, M, Cn, acc0=base)
    np.testing.assert_allclose(acc[3], outs[0][3] + 2.0, rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(acc[2], outs[0][2] - 1.0, rtol=1e-6, atol=1e-5)
    ev = emu.bn_backward(db, None, xb, mean, invstd, gamma, M, Cn, eval_mode=True)
    want = emu.to_bf16_bits((dyf * torch.from_numpy(gamma * invstd).double()).float())
    got = emu.from_bf16_bits(ev[0]).float()
    np.testing.assert_allclose(got.numpy(), emu.from_bf16_bits(want).numpy(), rtol=1e-2, atol=1e-6)
    assert np.array_equal(ev[3].view(np.uint32), outs[0][3].view(np.uint32))                       # the parameter gradients do not change


@pytest.mark.parametrize("Cn,seg", [(256, 0), (256, 3), (2048, 2), (512, 0)])
def test_bn_backward_apply_also_reduces_for_the_projection_shortcut(kernel_backend, Cn, seg):
    """lp_bn_bwd_apply_seg_ds == lp_bn_bwd_apply_seg (dx bit for bit, the same parameter gradients) + lp_bn_bwd_reduce of the masked gradient
    against a second BatchNorm's tensor, per segment (sums equal up to the fp32 order of the partial sums), with and without the ReLU mask"""
    import ctypes as C
    gen = torch.Generator().manual_seed(Cn + seg)
    rows_img, Bn = 40, 5
    M = Bn * rows_img
    dy = emu.to_bf16_bits(torch.randn(M, Cn, generator=gen))
    x = emu.to_bf16_bits(torch.randn(M, Cn, generator=gen))
    zd = emu.to_bf16_bits(torch.randn(M, Cn, generator=gen) * 1.5 + 0.2)
    y = emu.to_bf16_bits(torch.randn(M, Cn, generator=gen))                # ~half of it <= 0: the mask
    nseg = 2 if seg else 1
    mean, invstd = np.random.default_rng(1).normal(size=(nseg, Cn)).astype(np.float32), (np.random.default_rng(2).random((nseg, Cn)) + 0.5).astype(np.float32)
    mean_d, invstd_d = np.random.default_rng(3).normal(size=(nseg, Cn)).astype(np.float32), (np.random.default_rng(4).random((nseg, Cn)) + 0.5).astype(np.float32)
    gamma = (np.random.default_rng(5).random(Cn) + 0.5).astype(np.float32)
    lib = emu.lib()
    for masked in (False, True):
        db, xb, zb, yb = emu.Buf(dy), emu.Buf(x), emu.Buf(zd), (emu.Buf(y) if masked else None)
        mb, vb, gb, mdb, vdb = (emu.Buf(v) for v in (mean, invstd, gamma, mean_d, invstd_d))
        sums = emu.Buf(emu.to_fx(np.random.default_rng(6).normal(size=(nseg, 2, Cn)).astype(np.float32) * 3))
        counts = (float(seg * rows_img), float((Bn - seg) * rows_img)) if seg else (float(M), float(M))
        outs = []
        for fused in (False, True):
            dx, dres = emu.Z((M, Cn), np.uint16), emu.Z((M, Cn), np.uint16)
            dbeta, dgamma, tws = emu.Z(Cn), emu.Z(Cn), emu.Z(nseg * 2 * Cn)
            args = (db.p, emu.ptr(yb), xb.p, mb.p, vb.p, gb.p, sums.p, counts[0], counts[1], M, Cn, seg * rows_img, dx.p, dres.p, sums.p, dbeta.p,
                    dgamma.p, tws.p)
            sd = emu.ZX((nseg, 2, Cn))
            if fused:
                nws = lib.lp_bn_bwd_ds_workspace_bytes(M, Cn)
                ws = emu.Z(nws, np.uint8)
                emu.ok(lib.lp_bn_bwd_apply_seg_ds(*args, zb.p, mdb.p, vdb.p, sd.p, ws.p, nws, emu.stream()))
            else:
                emu.ok(lib.lp_bn_bwd_apply_seg(*args, emu.stream()))
                dresb = emu.Buf(dres.np())
                rws = emu._reduce_ws(M, Cn)
                for si, (r0, r1) in enumerate([(0, seg * rows_img), (seg * rows_img, M)] if seg else [(0, M)]):
                    part, zpart, m1, v1 = emu.Buf(dresb.np()[r0:r1]), emu.Buf(zd[r0:r1]), emu.Buf(mean_d[si]), emu.Buf(invstd_d[si])
                    one = emu.ZX((2, Cn))
                    emu.ok(lib.lp_bn_bwd_reduce(part.p, None, zpart.p, m1.p, v1.p, r1 - r0, Cn, one.p, rws.p, rws.nbytes, emu.stream()))
                    outs.append(("ref", si, emu.fx(one.np())))   # (.np() synchronises: the argument buffers above stay alive until here)
            outs.append(("dx", fused, dx.np(), dres.np(), dbeta.np(), dgamma.np(), emu.fx(sd.np())))
        ref = {si: v for tag, si, v in [o for o in outs if o[0] == "ref"]}
        plain, fusd = [o for o in outs if o[0] == "dx"]
        assert np.array_equal(plain[2], fusd[2]) and np.array_equal(plain[3], fusd[3])
        assert np.array_equal(plain[4], fusd[4]) and np.array_equal(plain[5], fusd[5])
        for si in range(nseg):
            np.testing.assert_allclose(fusd[6][si], ref[si], rtol=2e-5, atol=2e-4)
        assert np.abs(fusd[6]).max() > 1


@pytest.mark.parametrize("Cn,seg", [(256, 0), (64, 2), (2048, 3)])
def test_bn_apply_normalises_the_projection_shortcut_in_the_same_pass(kernel_backend, Cn, seg):
    """lp_bn_apply_seg_rbn == lp_bn_apply_seg(zd, no ReLU) -> lp_bn_apply_seg(z, residual = that), bit for bit: output and ReLU bits"""
    gen = torch.Generator().manual_seed(3 * Cn + seg)
    rows_img, Bn = 24, 5
    M = Bn * rows_img
    z = emu.to_bf16_bits(torch.randn(M, Cn, generator=gen) * 2)
    zd = emu.to_bf16_bits(torch.randn(M, Cn, generator=gen) * 1.5 + 0.3)
    nseg = 2 if seg else 1
    rng = np.random.default_rng(Cn)
    mean, mean_d = rng.normal(size=(nseg, Cn)).astype(np.float32), rng.normal(size=(nseg, Cn)).astype(np.float32)
    invstd, invstd_d = (rng.random((nseg, Cn)) + 0.5).astype(np.float32), (rng.random((nseg, Cn)) + 0.5).astype(np.float32)
    gamma, gamma_d = (rng.random(Cn) + 0.5).astype(np.float32), (rng.random(Cn) + 0.5).astype(np.float32)
    beta, beta_d = rng.normal(size=Cn).astype(np.float32) * 0.3, rng.normal(size=Cn).astype(np.float32) * 0.3
    lib = emu.lib()
    zb, zdb = emu.Buf(z), emu.Buf(zd)
    mb, vb, gb, bb, mdb, vdb, gdb, bdb = (emu.Buf(v) for v in (mean, invstd, gamma, beta, mean_d, invstd_d, gamma_d, beta_d))
    for relu in (1, 0):
        idt, y0, y1 = emu.Z((M, Cn), np.uint16), emu.Z((M, Cn), np.uint16), emu.Z((M, Cn), np.uint16)
        b0, b1 = emu.Z(M * Cn // 8, np.uint8), emu.Z(M * Cn // 8, np.uint8)
        emu.ok(lib.lp_bn_apply_seg(zdb.p, mdb.p, vdb.p, gdb.p, bdb.p, None, 0, M, Cn, seg * rows_img, idt.p, None, emu.stream()))
        emu.ok(lib.lp_bn_apply_seg(zb.p, mb.p, vb.p, gb.p, bb.p, idt.p, relu, M, Cn, seg * rows_img, y0.p, b0.p, emu.stream()))
        emu.ok(lib.lp_bn_apply_seg_rbn(zb.p, mb.p, vb.p, gb.p, bb.p, zdb.p, mdb.p, vdb.p, gdb.p, bdb.p, relu, M, Cn, seg * rows_img, y1.p, b1.p,
                                       emu.stream()))
        assert np.array_equal(y0.np(), y1.np()) and np.array_equal(b0.np(), b1.np())
        assert y1.np().any()


def test_segment_boundary_must_be_tile_aligned(kernel_backend):
    g = emu.geom(4, 6, 6, 64, 64, 1, 1, 1, 0)                    # 36 rows per image: 2 * 36 is not a multiple of 128
    x = emu.to_bf16_bits(torch.randn(4, 6, 6, 64))
    w = emu.to_bf16_bits(torch.randn(64, 1, 1, 64))
    assert emu.conv_fwd_bn(x, w, g, seg=2, rc=True) == -2        # LP_ERR_UNSUPPORTED: run the segments as two calls
    assert emu.conv_fwd_bn(x, w, g, seg=4, rc=True) == -2        # the second segment would be empty
    z = emu.to_bf16_bits(torch.randn(4 * 36, 64))
    m = np.zeros((2, 64), np.float32)
    assert emu.conv_dgrad_bn(x.reshape(-1, 64), w, g, z, m, m + 1, m[0] + 1, m[0], seg=2, rc=True) == -2


def test_finalize_two_segments_updates_running_statistics_in_order(kernel_backend):
    import ctypes as C

    Cn, gen = 24, torch.Generator().manual_seed(3)
    xs = [torch.randn(40, Cn, generator=gen) * 2 + 1, torch.randn(72, Cn, generator=gen) * 0.5 - 2]
    sums = np.stack([np.stack([x.sum(0).numpy(), (x * x).sum(0).numpy()]) for x in xs]).astype(np.float32)
    rm0, rv0 = torch.randn(Cn, generator=gen).numpy(), (torch.rand(Cn, generator=gen) + 0.5).numpy()
    sb, mean, invstd, rm, rv = emu.Buf(emu.to_fx(sums)), emu.Z((2, Cn)), emu.Z((2, Cn)), emu.Buf(rm0), emu.Buf(rv0)
    emu.ok(emu.lib().lp_bn_finalize2(sb.p, 40.0, 72.0, Cn, 1e-5, 0.1, mean.p, invstd.p, rm.p, rv.p, emu.stream()))
    want_rm, want_rv = torch.from_numpy(rm0.copy()), torch.from_numpy(rv0.copy())
    for si, x in enumerate(xs):                                  # what two forward calls of nn.BatchNorm2d do, in this order
        np.testing.assert_allclose(mean.np()[si], x.mean(0).numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(invstd.np()[si], (x.var(0, unbiased=False) + 1e-5).rsqrt().numpy(), rtol=1e-4)
        want_rm = 0.9 * want_rm + 0.1 * x.mean(0)
        want_rv = 0.9 * want_rv + 0.1 * x.var(0, unbiased=True)
    np.testing.assert_allclose(rm.np(), want_rm.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(rv.np(), want_rv.numpy(), rtol=1e-4, atol=1e-5)


NORM_CASES = [
    # B, seg, H, W, Ci, Co     (1x1 / stride 1 convolutions fed with the pre-normalisation tensor of the BatchNorm in front of them)
    (4, 0, 8, 8, 64, 256),        # one segment, N = 2 column tiles
    (4, 2, 8, 8, 128, 64),        # two segments, two K steps, N = 64 tiles
    (6, 4, 8, 16, 192, 200),      # unequal segments, three K steps, ragged N
    (3, 0, 10, 9, 64, 128),       # M = 270: ragged last tile / K step
]


@pytest.mark.parametrize("with_ws", [True, False])
@pytest.mark.parametrize("M,seg_rows,Cn", [(256, 128, 64), (300, 100, 24), (5000, 1800, 256), (96, 95, 8), (40, 24, 2048)])
def test_elementwise_batchnorm_kernels_with_two_segments(kernel_backend, M, seg_rows, Cn, with_ws):
    """lp_bn_apply_seg / lp_bn_bwd_apply_seg == the one-segment entry points called once per segment, bit for bit (one launch walks both
    segments, each with its own per-channel terms in registers).  with_ws: the two-launch form of the backward (terms_ws: the correction
    terms converted by bn_bwd_terms_kernel, round 5) and the self-contained one (NULL) give the same bits."""
    gen = torch.Generator().manual_seed(M + Cn)
    bits16 = lambda t: emu.to_bf16_bits(t)  # noqa: E731
    x, res, dy = (bits16(torch.randn(M, Cn, generator=gen)) for _ in range(3))
    mean = (torch.randn(2, Cn, generator=gen) * 0.3).numpy()
    invstd = (torch.rand(2, Cn, generator=gen) + 0.5).numpy()
    gamma, beta = (torch.rand(Cn, generator=gen) + 0.5).numpy(), (torch.randn(Cn, generator=gen) * 0.2).numpy()
    sums = torch.randn(2, 2, Cn, generator=gen).numpy()
    counts = (float(seg_rows), float(M - seg_rows))
    lib, st = emu.lib(), emu.stream()
    B = emu.Buf
    mb, ib, gb, bb, sb = B(mean), B(invstd), B(gamma), B(beta), B(emu.to_fx(sums))
    dbj, dgj, dbs, dgs = emu.Z(Cn), emu.Z(Cn), emu.Z(Cn), emu.Z(Cn)
    ws = emu.Z(4 * Cn) if with_ws else None
    wsp = ws.p if with_ws else None
    xb, rb, db = B(x), B(res), B(dy)
    nb = -(-M * Cn // 8)
    # joint launches
    y, bits, dx, dres = emu.Z((M, Cn), np.uint16), emu.Z(nb, np.uint8), emu.Z((M, Cn), np.uint16), emu.Z((M, Cn), np.uint16)
    emu.ok(lib.lp_bn_apply_seg(xb.p, mb.p, ib.p, gb.p, bb.p, rb.p, 1, M, Cn, seg_rows, y.p, bits.p if Cn % 8 == 0 and (seg_rows * Cn) % 8 == 0 else None, st))
    emu.ok(lib.lp_bn_bwd_apply_seg(db.p, y.p, xb.p, mb.p, ib.p, gb.p, sb.p, counts[0], counts[1], M, Cn, seg_rows, dx.p, dres.p, sb.p, dbj.p, dgj.p, wsp, st))
    got = (y.np().copy(), dx.np().copy(), dres.np().copy())
    # one call per segment
    want = [np.zeros((M, Cn), np.uint16) for _ in range(3)]
    for si, (r0, n) in enumerate(((0, seg_rows), (seg_rows, M - seg_rows))):
        ys, dxs, drs = emu.Z((n, Cn), np.uint16), emu.Z((n, Cn), np.uint16), emu.Z((n, Cn), np.uint16)
        xs, rs, ds = B(x[r0:r0 + n]), B(res[r0:r0 + n]), B(dy[r0:r0 + n])
        ms, is_, ss = B(mean[si]), B(invstd[si]), B(emu.to_fx(sums[si]))
        emu.ok(lib.lp_bn_apply(xs.p, ms.p, is_.p, gb.p, bb.p, rs.p, 1, n, Cn, ys.p, None, st))
        emu.ok(lib.lp_bn_bwd_apply(ds.p, ys.p, xs.p, ms.p, is_.p, gb.p, ss.p, counts[si], n, Cn, dxs.p, drs.p, ss.p, dbs.p, dgs.p, None, st))   # (always the self-contained form: the reference bits)
        for dst, src in zip(want, (ys, dxs, drs)):
            dst[r0:r0 + n] = src.np()
    for a, b_ in zip(got, want):
        assert np.array_equal(a, b_)
    # d beta / d gamma: the sums of both segments, whether one launch adds them or two
    tot = emu.fx(emu.to_fx(sums)).sum(0)
    for a, b_, t in ((dbj, dbs, tot[0]), (dgj, dgs, tot[1])):
        np.testing.assert_allclose(a.np(), t, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(b_.np(), t, rtol=1e-6, atol=1e-6)


def test_bn_bwd_apply_refuses_more_channels_than_its_term_buffer_holds(kernel_backend):
    """WITHOUT its terms_ws workspace bn_bwd_apply_kernel stages the launch's correction terms in 32 KB of LDS: C <= 2048 (ResNet-50's widest
    BatchNorm); beyond -> LP_ERR_UNSUPPORTED"""
    M, Cn = 8, 2056
    z16 = emu.Z((M, Cn), np.uint16)
    f = emu.Z(Cn)
    sums = emu.ZX((2, Cn))
    assert emu.lib().lp_bn_bwd_apply(z16.p, None, z16.p, f.p, f.p, f.p, sums.p, float(M), M, Cn, z16.p, None, None, None, None, None, emu.stream()) == -2
    ws = emu.Z(2 * Cn)   # ... with the caller's workspace for the terms there is no such limit
    assert emu.lib().lp_bn_bwd_apply(z16.p, None, z16.p, f.p, f.p, f.p, sums.p, float(M), M, Cn, z16.p, None, None, None, None, ws.p, emu.stream()) == 0
