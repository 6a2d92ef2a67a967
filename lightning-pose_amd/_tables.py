"""Host-side tap tables for the fused decode kernel (csrc/decode.hip).

The reference up-samples a heat-map ``downsample_factor`` times with ``F.interpolate(bicubic, align_corners=False)``
followed by a zero-padded 5x5 binomial filter (lightning_pose/models/heads/heatmap.py:86-100, 122-124).  Both steps
are linear and separable, so along each axis the whole chain is ONE banded matrix ``U`` (n*2^ds x n).  It is built
here in float64, then cut into the three banded forms the kernel consumes (include/lp_hip.h ``lp_decode_tables``).
Pure numpy: no device needed; the result is cached per (n, ds).
"""

from __future__ import annotations

import functools
import math

import numpy as np

COL_TAPS = 12  # kTXM in csrc/decode.hip


def _bicubic_x2(n: int) -> np.ndarray:
    """ATen upsample_bicubic2d, scale 2, align_corners=False, along one axis: A=-0.75, index clamping."""
    a = -0.75
    m = np.zeros((2 * n, n))
    for o in range(2 * n):
        src = (o + 0.5) * 0.5 - 0.5
        i0 = math.floor(src)
        t = src - i0
        cw = (
            ((a * (t + 1) - 5 * a) * (t + 1) + 8 * a) * (t + 1) - 4 * a,
            ((a + 2) * t - (a + 3)) * t * t + 1,
            ((a + 2) * (1 - t) - (a + 3)) * (1 - t) * (1 - t) + 1,
            ((a * (2 - t) - 5 * a) * (2 - t) + 8 * a) * (2 - t) - 4 * a,
        )
        for k in range(4):
            m[o, min(max(i0 - 1 + k, 0), n - 1)] += cw[k]
    return m


def _binomial5(n: int) -> np.ndarray:
    """[1,4,6,4,1]/16 along one axis with zero padding (kornia filter2d border_type='constant')."""
    m = np.zeros((n, n))
    for k, wv in enumerate((1 / 16, 4 / 16, 6 / 16, 4 / 16, 1 / 16)):
        idx = np.arange(n) + k - 2
        ok = (idx >= 0) & (idx < n)
        m[np.arange(n)[ok], idx[ok]] += wv
    return m


@functools.lru_cache(maxsize=None)
def upsample_matrix(n: int, ds: int) -> np.ndarray:
    u = np.eye(n)
    size = n
    for _ in range(ds):
        u = _binomial5(2 * size) @ _bicubic_x2(size) @ u
        size *= 2
    return u


def decode_window(ds: int, n: int) -> int:
    """Must equal lp_decode_window() in csrc/decode.hip."""
    if ds == 1:
        return 8 if n >= 8 else 0
    if ds == 2:
        return 9 if n >= 9 else (8 if n == 8 else 0)
    if ds == 3:
        return 11 if n >= 11 else 0
    return 0


@functools.lru_cache(maxsize=None)
def axis_tables(n: int, ds: int) -> dict[str, np.ndarray | int]:
    """Banded forms of ``upsample_matrix(n, ds)``: grouped rows, per-output-column taps, per-input-column taps."""
    r = 1 << ds
    ty = decode_window(ds, n)
    if ty == 0:
        raise NotImplementedError(f"decode: heat-map axis of {n} px is not supported with downsample_factor={ds}")
    u = upsample_matrix(n, ds)
    big = n * r
    nz = np.abs(u) > 0
    lo = np.array([np.flatnonzero(row)[0] for row in nz])
    hi = np.array([np.flatnonzero(row)[-1] for row in nz])

    # grouped rows: group j = output rows j*r .. j*r+r-1 share one window of `ty` input rows
    row_base = np.zeros(n, dtype=np.int32)
    row_taps = np.zeros((n, r, ty), dtype=np.float32)
    for j in range(n):
        glo, ghi = lo[j * r:(j + 1) * r].min(), hi[j * r:(j + 1) * r].max()
        base = max(0, min(glo, n - ty))
        assert ghi - base < ty, "window too small for the band"
        row_base[j] = base
        row_taps[j] = u[j * r:(j + 1) * r, base:base + ty]
    steps = np.diff(row_base)
    assert row_base[0] == 0 and row_base[-1] == n - ty and set(steps.tolist()) <= {0, 1}, "kernel needs unit window steps"
    # lp_decode_bwd splits the row groups over up to 8 waves (csrc/decode.hip: decode_bwd_threads); a wave's rows of the LDS strip overlap its
    # neighbours'.  The strip accumulation (owner stores, then the guests add in two phases by wave parity) is race-free while no input row has
    # more than three contributing waves - true for every n up to 256 at every downsample factor (two everywhere except n = 65, ds = 1)
    waves = max(1, min(8, n // ty))
    seg = -(-n // waves)
    cover = np.zeros(n, dtype=np.int32)
    for k in range(waves):
        j0, j1 = k * seg, min(n, (k + 1) * seg)
        if j0 < j1:
            cover[row_base[j0]:row_base[j1 - 1] + ty] += 1
    if cover.min() < 1 or cover.max() > 3:
        raise NotImplementedError(f"decode tables for n = {n}, downsample_factor = {ds}: a strip row with {int(cover.max())} contributing waves")

    # per output column: start + COL_TAPS taps
    tx = min(COL_TAPS, n)
    col_start = np.zeros(big, dtype=np.int32)
    col_taps = np.zeros((big, COL_TAPS), dtype=np.float32)
    for c in range(big):
        s = max(0, min(lo[c], n - tx))
        assert hi[c] - s < tx
        col_start[c] = s
        col_taps[c, :tx] = u[c, s:s + tx]

    # transposed: for each input column q the contiguous range of output columns it feeds
    clo = np.array([np.flatnonzero(nz[:, q])[0] for q in range(n)])
    chi = np.array([np.flatnonzero(nz[:, q])[-1] for q in range(n)])
    tc = int((chi - clo + 1).max())
    colt_start = np.zeros(n, dtype=np.int32)
    colt_taps = np.zeros((n, tc), dtype=np.float32)
    for q in range(n):
        s = max(0, min(clo[q], big - tc))
        colt_start[q] = s
        colt_taps[q] = u[s:s + tc, q]
    return {"row_base": row_base, "row_taps": row_taps, "col_start": col_start, "col_taps": col_taps,
            "colT_start": colt_start, "colT_taps": colt_taps, "ty": ty, "tx": tx, "tc": tc}
