// Fused soft-argmax decode for heatmap trackers (forward + backward), gfx950.
//
// Replaces, in ONE launch per pass, the reference chain (paths relative to the reference tree)
//   models/heads/heatmap.py:103-144  run_subpixelmaxima
//     :86-100   upsample x downsample_factor  (F.interpolate bicubic x2 + kornia filter2d 5x5, zero pad)
//     :126-127  spatial_softmax2d(T) + spatial_expectation2d
//     :129      data/heatmaps.py:90-142 evaluate_heatmaps_at_location (5x5 confidence window)
//     :131-136  sub-pixel offset
//   data/utils.py:142-234   undo_affine_transform_batch
//   data/bboxes.py:222-288  model_to_frame_batch
// which materialises the (B,K,H,W) input-resolution map five times in HBM.
//
// Algorithm.  Both resampling steps are linear and separable, so the upsampled map is Y = Uy * Hm * Ux^T
// with banded composite matrices U (<= 11 non-zeros per row, built in fp64 on the host and passed as
// tap tables).  One workgroup owns one (frame, keypoint) heatmap: the h x w tile is staged once in LDS
// (the only HBM read: h*w*4 algorithmic bytes) and each lane walks DOWN one output column keeping a sliding
// window of Z = Hm * Ux^T in registers; a new window row costs TX LDS reads + FMAs and is produced by the lane
// itself, so neither Z nor Y is stored anywhere and the LDS footprint is just the tile (4 workgroups per CU).
// The softmax(T*Y) expectation is accumulated online (running max / sum / sum*x / sum*y per lane, merged with
// wave shuffles).  The 25-tap confidence window is recomputed from the LDS tile once the global max and sum are
// known.  Backward recomputes Y the same way (8 waves = 8 row segments of a 64-column strip), forms
// G = T*p*(gx*(x-ex) + gy*(y-ey)) on the fly, scatters Uy^T G into a [h][64] LDS strip and applies Ux^T from there.
//
// This kernel is fp32-VALU bound by construction (about 350 FLOP per algorithmic byte): see DESIGN.md.
#include <stdlib.h>

#include "lp_common.h"

namespace lp {

constexpr int kTXM = 12;        // padded column-tap count

struct DecodeTables {
    const int* row_base;     // [h]            first input row of group j's window
    const float* row_taps;   // [h][R][TY]     taps of output row j*R+rr relative to row_base[j]
    const int* col_start;    // [W]            first input column of output column c
    const float* col_taps;   // [W][kTXM]
    const int* colT_start;   // [w]            first OUTPUT column touching input column q   (backward)
    const float* colT_taps;  // [w][TC]
    int TX;                  // valid taps per output column (<= kTXM, <= w)
    int TC;                  // valid taps per input column in the transposed table
};

struct FrameMap {            // undo-affine + model->frame epilogue
    const float* tf;         // transforms or nullptr
    int tf_mode;             // LP_TF_NONE / SINGLE / PER_FRAME / PER_VIEW
    const float* bbox;       // [B][4*views] rows [x, y, h, w] or nullptr (identity)
    int bbox_stride;
    int kp_per_view;         // K / views
    float model_h, model_w;
};

__device__ __forceinline__ void merge_softmax(float& m, float& s, float& sx, float& sy, float m2, float s2, float sx2,
                                              float sy2) {
    const float mm = fmaxf(m, m2);
    const float a = (m == -INFINITY) ? 0.f : __expf(m - mm);
    const float b = (m2 == -INFINITY) ? 0.f : __expf(m2 - mm);
    s = s * a + s2 * b;
    sx = sx * a + sx2 * b;
    sy = sy * a + sy2 * b;
    m = mm;
}

// Index of the tile's maximum (first one in row-major order), the same in every thread.  Round 6: the soft-argmax accumulates its two first
// moments RELATIVE to this point - (x0, y0) = R * (column, row) + R / 2 in up-sampled pixels - instead of from the map's corner: the
// expectation of a peaked map is then a sum of offsets of a few pixels (fp32 resolves 1e-7 px there) instead of coordinates up to 384 (3e-5),
// and the backward pass's (c - E[x]) keeps the digits the old form lost - its gradient noise against exact arithmetic fell from 5e-5 to the
// reference's own 5e-6.  Forward and backward both derive the point from the tile itself, so the
// `stats` they exchange carry only the small offsets.  Order-independent (value, then lowest index): both kernels find the same point.
#ifndef LP_DEC_CENTER
#define LP_DEC_CENTER 1   // (A/B builds: 0 = moments from the map's corner, rounds 1 - 5)
#endif
// The tile is staged into LDS and its maximum found in the SAME pass: every thread tracks the largest value it copied, the waves' candidates
// meet in `am` (2 x 8 words of their own: nothing else ever touches them) across the barrier that publishes the tile anyway.  (The first
// form of round 6 scanned the staged tile in a pass of its own between three more barriers: forward +13 %, profiles/r06d_decode_ab.txt.)
// Call `tile_stage` in every thread, then a `__syncthreads()`, then `tile_argmax_read`.
__device__ __forceinline__ void tile_stage(const float* __restrict__ src, float* hs, int n, float* am) {
    const int tid = threadIdx.x, nthreads = blockDim.x, lane = tid & 63, wave = tid >> 6;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = tid; i < n; i += nthreads) {
        const float v = src[i];
        hs[i] = v;
        if (LP_DEC_CENTER && v > bv) bv = v, bi = i;
    }
    if (!LP_DEC_CENTER) return;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const float ov = __shfl_xor(bv, d, 64);
        const int oi = __shfl_xor(bi, d, 64);
        if (ov > bv || (ov == bv && oi < bi)) bv = ov, bi = oi;
    }
    if (lane == 0) {
        am[wave] = bv;
        am[8 + wave] = __int_as_float(bi);
    }
}
__device__ __forceinline__ int tile_argmax_read(const float* am) {
    if (!LP_DEC_CENTER) return -1;
    const int nwaves = blockDim.x >> 6;
    float bv = am[0];
    int bi = __float_as_int(am[8]);
    for (int i = 1; i < nwaves; ++i) {
        const float ov = am[i];
        const int oi = __float_as_int(am[8 + i]);
        if (ov > bv || (ov == bv && oi < bi)) bv = ov, bi = oi;
    }
    return __builtin_amdgcn_readfirstlane(bi == 0x7fffffff ? 0 : bi);   // (a tile of NaNs / -inf: any point will do, the outputs are NaN anyway; the same in every lane: a scalar)
}

// One element of Z = Hm * Ux^T for this lane's output column: `hcol` = tile + first input column of the lane's taps.
template <bool FULLTX>
__device__ __forceinline__ float z_value(const float* hcol, int r, int w, const float (&tx)[kTXM], int TX) {
    const float* row = hcol + r * w;
    float a = 0.f;
#pragma unroll
    for (int t = 0; t < kTXM; ++t)
        if (FULLTX || t < TX) a = fmaf(row[t], tx[t], a);
    return a;
}

// One upsampled value from the LDS tile (used for the confidence window).
__device__ __forceinline__ float upsampled_at(const float* hs, int w, int oy, int ox, int R, int TY, const DecodeTables& tb) {
    const int j = oy / R, rr = oy - j * R;
    const int yb = tb.row_base[j];
    const float* ty = tb.row_taps + (j * R + rr) * TY;
    const int xs = tb.col_start[ox];
    const float* tx = tb.col_taps + ox * kTXM;
    float y = 0.f;
    for (int a = 0; a < TY; ++a) {
        const float* row = hs + (yb + a) * w + xs;
        float z = 0.f;
        for (int t = 0; t < tb.TX; ++t) z = fmaf(row[t], tx[t], z);
        y = fmaf(ty[a], z, y);
    }
    return y;
}

__device__ __forceinline__ void to_frame(float x, float y, int b, int k, const FrameMap& fm, float& xf, float& yf) {
    const int v = fm.kp_per_view > 0 ? k / fm.kp_per_view : 0;
    if (fm.tf_mode != LP_TF_NONE) {
        const float* a = fm.tf;
        if (fm.tf_mode == LP_TF_PER_FRAME) a += b * 6;
        if (fm.tf_mode == LP_TF_PER_VIEW) a += v * 6;
        const float det = a[0] * a[4] - a[1] * a[3];
        const float i00 = a[4] / det, i01 = -a[1] / det, i10 = -a[3] / det, i11 = a[0] / det;
        const float ox = -(i00 * a[2] + i01 * a[5]), oy = -(i10 * a[2] + i11 * a[5]);
        const float xn = x * i00 + y * i01 + ox;
        const float yn = x * i10 + y * i11 + oy;
        x = xn;
        y = yn;
    }
    if (fm.bbox != nullptr) {
        const float* bb = fm.bbox + b * fm.bbox_stride + 4 * v;
        x = (x / fm.model_w) * bb[3] + bb[0];
        y = (y / fm.model_h) * bb[2] + bb[1];
    }
    xf = x;
    yf = y;
}

// d(frame)/d(aug) is the constant 2x2 matrix J = diag(bw/Wm, bh/Hm) * Ainv; returns J^T g.
__device__ __forceinline__ void frame_grad_to_aug(float gxf, float gyf, int b, int k, const FrameMap& fm, float& gx, float& gy) {
    const int v = fm.kp_per_view > 0 ? k / fm.kp_per_view : 0;
    float sxs = 1.f, sys = 1.f;
    if (fm.bbox != nullptr) {
        const float* bb = fm.bbox + b * fm.bbox_stride + 4 * v;
        sxs = bb[3] / fm.model_w;
        sys = bb[2] / fm.model_h;
    }
    gxf *= sxs;
    gyf *= sys;
    if (fm.tf_mode != LP_TF_NONE) {
        const float* a = fm.tf;
        if (fm.tf_mode == LP_TF_PER_FRAME) a += b * 6;
        if (fm.tf_mode == LP_TF_PER_VIEW) a += v * 6;
        const float det = a[0] * a[4] - a[1] * a[3];
        const float i00 = a[4] / det, i01 = -a[1] / det, i10 = -a[3] / det, i11 = a[0] / det;
        gx = i00 * gxf + i10 * gyf;
        gy = i01 * gxf + i11 * gyf;
    } else {
        gx = gxf;
        gy = gyf;
    }
}

// ------------------------------------------------------------------------------------------------
// exact pruning at high temperature
// ------------------------------------------------------------------------------------------------
// A term exp(T*y - m) with T*y < m - kPruneMargin is smaller than e^-50 = 2e-22 times the largest term; all of them together (at most
// H*W = 147 456 at 384 x 384) are below 3e-17 of the sum - nine orders of magnitude under fp32's resolution (6e-8), so skipping them moves
// no result by more than the last bit (tests/test_emu_decode.py::test_exact_pruning_changes_nothing; parity tolerance 1e-4 px).  With the
// reference's T = 1000 that is everything more than 0.05 below the peak of the up-sampled map - for the peaked maps of a trained network
// (peak ~0.1) all but the few rows / columns around the peak.  The bound used to decide is
//     |y(j, c)| <= (max_row sum_t |ty|) * (sum_t |tx_c|) * min( max |Hm| over the rows of window j , max |Hm| over the columns of c's taps )
// from per-row / per-column maxima of the staged tile (2 * h * w LDS reads, once).  Row groups are skipped per wave (the row bound is
// wave-uniform), whole waves / strips when none of their columns can contribute.  On flat maps (an untrained network) nothing is skipped.
constexpr float kPruneMargin = 50.f;
constexpr int kPruneMaxDim = 256;  // h, w up to this use pruning (LDS for the row / column maxima)

struct PruneState {
    float rmx[kPruneMaxDim];   // max |Hm[r][:]|
    float cmx[kPruneMaxDim];   // max |Hm[:][q]|
    float rwin[kPruneMaxDim];  // max of rmx over the input rows of output-row group j's window
    float red[16];
};

// fills ps.rmx / cmx / rwin from the staged tile and returns Ly = max over output rows of sum_t |row tap| (block-wide)
template <int R, int TY>
__device__ __forceinline__ float prune_setup(PruneState& ps, const float* hs, int h, int w, const DecodeTables& tb) {
    const int tid = threadIdx.x, nthreads = blockDim.x, lane = tid & 63, wave = tid >> 6, nwaves = nthreads >> 6;
    for (int r = tid; r < h; r += nthreads) {
        float a = 0.f;
        for (int q = 0; q < w; ++q) a = fmaxf(a, fabsf(hs[r * w + q]));
        ps.rmx[r] = a;
    }
    for (int q = tid; q < w; q += nthreads) {
        float a = 0.f;
        for (int r = 0; r < h; ++r) a = fmaxf(a, fabsf(hs[r * w + q]));
        ps.cmx[q] = a;
    }
    float ly = 0.f;
    for (int i = tid; i < h * R; i += nthreads) {
        float a = 0.f;
        for (int t = 0; t < TY; ++t) a += fabsf(tb.row_taps[(size_t)i * TY + t]);
        ly = fmaxf(ly, a);
    }
    ly = wave_max(ly);
    if (lane == 0) ps.red[wave] = ly;
    __syncthreads();
    ly = ps.red[0];
    for (int i = 1; i < nwaves; ++i) ly = fmaxf(ly, ps.red[i]);
    for (int j = tid; j < h; j += nthreads) {
        const int b0 = tb.row_base[j];
        float a = 0.f;
        for (int t = 0; t < TY; ++t) a = fmaxf(a, ps.rmx[b0 + t]);
        ps.rwin[j] = a;
    }
    __syncthreads();
    return ly;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int R, int TY, bool FULLTX, bool PRUNE>
__global__ __launch_bounds__(512) void decode_fwd_kernel(const float* __restrict__ heat, int K, int h, int w, float temperature,
                                                         float offset, DecodeTables tb, FrameMap fm,
                                                         float* __restrict__ kp_aug, float* __restrict__ kp_frame,
                                                         float* __restrict__ conf, float* __restrict__ stats) {
    HIP_DYNAMIC_SHARED(float, smem)
    float* hs = smem;                 // [h][w]
    __shared__ float red[8 * 4];
    // (PRUNE is a template parameter so that the un-pruned instantiation stays the round-1 kernel, register for register: the pruning
    // state costs ~20 VGPRs, i.e. one of the four resident workgroups per CU)
    PruneState& ps = *reinterpret_cast<PruneState*>(smem + h * w);  // behind the tile (the launcher sizes the dynamic LDS for it)
    constexpr bool prune = PRUNE;

    const int bk = blockIdx.x;
    const int b = bk / K, k = bk - b * K;
    const int H = h * R, W = w * R;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nthreads = blockDim.x, nwaves = nthreads >> 6;

    const auto row_base = as_uniform(tb.row_base);   // (wave-uniform indices: scalar loads, lp_common.h: as_uniform)
    const auto row_taps = as_uniform(tb.row_taps);
    const float* src = heat + (size_t)bk * h * w;
    __shared__ float am[16];
    tile_stage(src, hs, h * w, am);
    __syncthreads();
    // the point the moments are taken about (the tile's maximum): whole numbers, so (column - x0) and (row - y0) are exact
    const int amax = tile_argmax_read(am);
    const float y0 = amax < 0 ? 0.f : (float)((amax / w) * R + R / 2), x0 = amax < 0 ? 0.f : (float)((amax % w) * R + R / 2);

    // ---- pruning set-up: bounds from the tile, and a LOWER bound of the final maximum from the row group at the tile's largest row
    float ly = 0.f, m_lb = -INFINITY;
    if (prune) {
        ly = prune_setup<R, TY>(ps, hs, h, w, tb);
        int jstar = 0;
        float best = -1.f;
        for (int r = 0; r < h; ++r)
            if (ps.rmx[r] > best) {
                best = ps.rmx[r];
                jstar = r;
            }
        float ym = -INFINITY;
        for (int c = tid; c < W; c += nthreads) {  // (no wave collectives inside: lanes past W simply sit out)
            float tx[kTXM];
#pragma unroll
            for (int t = 0; t < kTXM; ++t) tx[t] = tb.col_taps[c * kTXM + t];
            const float* hcol = hs + tb.col_start[c];
            const int b0 = row_base[jstar];
            float wv[TY];
#pragma unroll
            for (int t = 0; t < TY; ++t) wv[t] = z_value<FULLTX>(hcol, b0 + t, w, tx, tb.TX);
            const auto taps = row_taps + (size_t)jstar * R * TY;
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                float y = 0.f;
#pragma unroll
                for (int t = 0; t < TY; ++t) y = fmaf(taps[rr * TY + t], wv[t], y);
                ym = fmaxf(ym, y * temperature);
            }
        }
        ym = wave_max(ym);
        if (lane == 0) ps.red[8 + wave] = ym;
        __syncthreads();
        m_lb = ps.red[8];
        for (int i = 1; i < nwaves; ++i) m_lb = fmaxf(m_lb, ps.red[8 + i]);
    }
    const float cut = (m_lb - kPruneMargin) / fmaxf(temperature, 1e-30f);  // |y| bounds below this contribute exactly 0

    float m = -INFINITY, s = 0.f, sx = 0.f, sy = 0.f;
    for (int c0 = 0; c0 < W; c0 += nthreads) {  // one trip when the block covers the row (the host sizes it so)
        // every lane takes every trip (the pruning decisions are wave collectives); lanes past W repeat the last column and add nothing
        const bool valid = c0 + tid < W;
        const int c = valid ? c0 + tid : W - 1;
        float tx[kTXM];
#pragma unroll
        for (int t = 0; t < kTXM; ++t) tx[t] = tb.col_taps[c * kTXM + t];
        const float* hcol = hs + tb.col_start[c];
        const float xc = (float)c - x0;
        float lx = 0.f;       // wave-wide bound factor of the columns: max over the wave of sum_t |tx|, and of their column maxima
        bool wave_live = true;
        if (prune) {
            float cm = 0.f;
#pragma unroll
            for (int t = 0; t < kTXM; ++t) {
                lx += fabsf(tx[t]);
                if (t < tb.TX) cm = fmaxf(cm, ps.cmx[tb.col_start[c] + t]);
            }
            const float colb = wave_max(ly * lx * cm);  // no column of this wave can exceed it, whatever the row
            lx = wave_max(lx);
            wave_live = !(colb < cut);
        }
        if (!wave_live) continue;  // (wave-uniform: every lane of the wave has the same colb)
        float win[TY];
        int base = row_base[0];
        bool have = false;         // is `win` the window of `base`?
        for (int j = 0; j < h; ++j) {
            const int nb = row_base[j];
            if (prune && ly * lx * ps.rwin[j] < cut) {  // the whole row group is below the cut for every column of the wave
                have = false;
                continue;
            }
            if (!have) {
#pragma unroll
                for (int t = 0; t < TY; ++t) win[t] = z_value<FULLTX>(hcol, nb + t, w, tx, tb.TX);
                base = nb;
                have = true;
            }
            if (nb != base) {  // windows advance by exactly one input row (host asserts it)
#pragma unroll
                for (int t = 0; t < TY - 1; ++t) win[t] = win[t + 1];
                win[TY - 1] = z_value<FULLTX>(hcol, nb + TY - 1, w, tx, tb.TX);
                base = nb;
            }
            const auto taps = row_taps + (size_t)j * R * TY;
            float z[R];
            float gm = m;
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                float y = 0.f;
#pragma unroll
                for (int t = 0; t < TY; ++t) y = fmaf(taps[rr * TY + t], win[t], y);
                z[rr] = y * temperature;
                gm = fmaxf(gm, z[rr]);
            }
            if (!valid) continue;
            const float sc = (m == -INFINITY) ? 0.f : __expf(m - gm);
            m = gm;
            // the row group's R terms are summed first (their column is the lane's, their rows are j R + 0 .. R - 1): one update of each moment
            // per row GROUP instead of per row - 10 vector operations where the per-row form took 16 (round 6: with the moments taken about
            // (x0, y0) the per-row form cost the kernel 12 %, profiles/r06g_decode_ab.txt)
            float es = 0.f, ew = 0.f;
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                const float e = __expf(z[rr] - gm);
                es += e;
                ew = fmaf(e, (float)rr, ew);
            }
            s = fmaf(s, sc, es);
            sx = fmaf(sx, sc, es * xc);
            sy = fmaf(sy, sc, fmaf(es, (float)(j * R) - y0, ew));
        }
    }

    // merge the per-lane online-softmax states: wave shuffles, then across the waves through LDS
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const float m2 = __shfl_xor(m, d, 64), s2 = __shfl_xor(s, d, 64);
        const float sx2 = __shfl_xor(sx, d, 64), sy2 = __shfl_xor(sy, d, 64);
        merge_softmax(m, s, sx, sy, m2, s2, sx2, sy2);
    }
    if (lane == 0) {
        red[wave * 4 + 0] = m;
        red[wave * 4 + 1] = s;
        red[wave * 4 + 2] = sx;
        red[wave * 4 + 3] = sy;
    }
    __syncthreads();
    m = red[0];
    s = red[1];
    sx = red[2];
    sy = red[3];
    for (int i = 1; i < nwaves; ++i) merge_softmax(m, s, sx, sy, red[i * 4 + 0], red[i * 4 + 1], red[i * 4 + 2], red[i * 4 + 3]);
    const float dex = sx / s, dey = sy / s;       // E[x] - x0, E[y] - y0
    const float ex = x0 + dex, ey = y0 + dey;

    // confidence: softmax mass in the 5x5 window at trunc(ex, ey), zero outside the map
    float cpart = 0.f;
    if (tid < 25) {
        const int oy = (int)ey + tid / 5 - 2;
        const int ox = (int)ex + tid % 5 - 2;
        if (oy >= 0 && oy < H && ox >= 0 && ox < W) {
            const float y = upsampled_at(hs, w, oy, ox, R, TY, tb);
            cpart = __expf(y * temperature - m) / s;
        }
    }
    if (wave == 0) {
        cpart = wave_sum(cpart);
        if (lane == 0) {
            const float xa = ex - offset, ya = ey - offset;
            kp_aug[bk * 2 + 0] = xa;
            kp_aug[bk * 2 + 1] = ya;
            float xf, yf;
            to_frame(xa, ya, b, k, fm, xf, yf);
            kp_frame[bk * 2 + 0] = xf;
            kp_frame[bk * 2 + 1] = yf;
            conf[bk] = cpart;
            stats[bk * 4 + 0] = m;
            stats[bk * 4 + 1] = s;
            stats[bk * 4 + 2] = dex;   // (the offsets from the tile's own maximum: lp_decode_bwd rebuilds (x0, y0) from the tile)
            stats[bk * 4 + 3] = dey;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward: d(loss)/d(heat) from d(loss)/d(kp_aug) (+ d(loss)/d(kp_frame) chained through the frame map)
// ------------------------------------------------------------------------------------------------
#ifndef LP_DEC_RR_FULL
#define LP_DEC_RR_FULL 16   // (A/B builds) row-group tap tables up to this many scalars are walked fully unrolled ...
#endif
#ifndef LP_DEC_TX_REGS
#define LP_DEC_TX_REGS 2   // column taps of a lane: 0 = re-read per use, 1 = in registers, 2 = in registers in the plain kernels only
#endif
#ifndef LP_DEC_RR_PART
#define LP_DEC_RR_PART 2    // ... larger ones this many output rows at a time
#endif
#ifndef LP_DEC_PROBE
#define LP_DEC_PROBE 0   // (timing builds only, profiles/r06b.sh: 1 = no strip product (phase B), 2 = no per-row arithmetic, 4 = no window values, 8 = no phase A)
#endif
constexpr int kBwdStrip = 64;          // output columns per strip = one wave of lanes; the block's waves split the rows
constexpr int kBwdLd = kBwdStrip + 1;  // LDS row stride of the strip (odd: lanes that walk down a column hit distinct banks)

__host__ __device__ inline int bwd_tap_ld(int TC) { return TC | 1; }  // odd row stride of the staged transposed tap table

// (second launch bound = waves per SIMD: two 8-wave workgroups per CU for the usual map sizes, LDS allows it)
template <int R, int TY, int NE, bool FULLTX, bool PRUNE>
__global__ __launch_bounds__(512, (NE <= 18 ? 4 : 2)) void decode_bwd_kernel(const float* __restrict__ heat, int K, int h, int w,
                                                                             float temperature, DecodeTables tb, FrameMap fm,
                                                                             const float* __restrict__ stats,
                                                                             const float* __restrict__ g_aug,
                                                                             const float* __restrict__ g_frame,
                                                                             float* __restrict__ g_heat, int accumulate) {
    HIP_DYNAMIC_SHARED(float, smem)
    constexpr bool prune = PRUNE;
    constexpr int SC = kBwdStrip, LD = kBwdLd;
    const int TCP = bwd_tap_ld(tb.TC);
    float* hs = smem;            // [h][w]   heat-map tile; reused at the end to transpose the result for coalesced stores
    float* zs = hs + h * w;      // [h][65]  strip of Uy^T G, summed over the row segments with LDS float adds (every element
                                 //          has at most two contributing segments, so the sum does not depend on their order)
    float* lt = zs + h * LD;     // [w][TCP] transposed column taps
    int* lcs = reinterpret_cast<int*>(lt + w * TCP);  // [w] first output column touching input column q
    PruneState& ps = *reinterpret_cast<PruneState*>(lcs + w);  // (PRUNE only: the launcher sizes the dynamic LDS for it)

    const int bk = blockIdx.x;
    const int b = bk / K, k = bk - b * K;
    const int W = w * R;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nthreads = blockDim.x, nwaves = nthreads >> 6;
    const int seg = (h + nwaves - 1) / nwaves;  // row groups per wave (host keeps it >= TY)
    const int j0 = wave * seg, j1 = min(h, j0 + seg);

    // the row tables are indexed by the (wave-uniform) row-group counter: scalar loads (lp_common.h: as_uniform)
    const auto row_base = as_uniform(tb.row_base);
    const auto row_taps = as_uniform(tb.row_taps);

    const float* src = heat + (size_t)bk * h * w;
    __shared__ float am[16];
    tile_stage(src, hs, h * w, am);
    for (int i = tid; i < w * tb.TC; i += nthreads) {
        const int q = i / tb.TC;
        lt[q * TCP + (i - q * tb.TC)] = tb.colT_taps[i];
    }
    for (int i = tid; i < w; i += nthreads) lcs[i] = tb.colT_start[i];

    float gx = 0.f, gy = 0.f;
    if (g_aug != nullptr) {
        gx = g_aug[bk * 2 + 0];
        gy = g_aug[bk * 2 + 1];
    }
    if (g_frame != nullptr) {
        float ax, ay;
        frame_grad_to_aug(g_frame[bk * 2 + 0], g_frame[bk * 2 + 1], b, k, fm, ax, ay);
        gx += ax;
        gy += ay;
    }
    const float m = stats[bk * 4 + 0], inv_s = 1.f / stats[bk * 4 + 1];
    const float dex = stats[bk * 4 + 2], dey = stats[bk * 4 + 3];   // E[x] - x0, E[y] - y0 (x0, y0: the tile's maximum, below)
    const float gxt = gx * temperature, gyt = gy * temperature;

    __syncthreads();  // the staged tile is complete
    const int amax = tile_argmax_read(am);
    const float y0 = amax < 0 ? 0.f : (float)((amax / w) * R + R / 2), x0 = amax < 0 ? 0.f : (float)((amax % w) * R + R / 2);
    // pruning (see "exact pruning at high temperature"): here the exact maximum is known from the forward pass
    float ly = 0.f;
    if (prune) ly = prune_setup<R, TY>(ps, hs, h, w, tb);
    const float cut = prune ? (m - kPruneMargin) / fmaxf(temperature, 1e-30f) : -INFINITY;

    // this thread's elements of dH: e = tid + i * nthreads -> (q, r) = (e / h, e % h): consecutive lanes walk down one input
    // column q, so the tap reads broadcast and the strip reads are conflict-free
    float dacc[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) dacc[i] = 0.f;
    const float inv_h = 1.f / (float)h;

    for (int c0 = 0; c0 < W; c0 += SC) {
        // a strip none of whose columns can reach the cut contributes nothing (every wave sees the same 64 columns: block-uniform)
        float lx = 0.f;
        {
            const int cc = min(c0 + lane, W - 1);
            float cm = 0.f;
#pragma unroll
            for (int t = 0; t < kTXM; ++t) {
                lx += fabsf(tb.col_taps[cc * kTXM + t]);
                if (prune && t < tb.TX) cm = fmaxf(cm, ps.cmx[tb.col_start[cc] + t]);
            }
            const float colb = wave_max(ly * lx * cm);
            lx = wave_max(lx);
            if (prune && colb < cut) continue;
        }
        __syncthreads();  // (first trip: the tile / table staging; later trips: the previous strip's product is done reading zs)
        // Who writes what into the strip (round 6: no LDS atomics).  A wave's row range overlaps only its neighbours' - by the TY - 1 rows of a
        // window - so every strip element has one or two contributors.  The HIGHER wave of a shared row owns it: owners write with plain
        // stores (the rows that leave a wave's window while it walks are never shared with the wave below it, and the rows it still holds
        // at the end are stored there unless the wave below shares them); the guests - a wave's last rows, from the first row of the next
        // wave on - are ADDED after a barrier, by then initialised.  Every element is written exactly once before it is read: no zero
        // fill either.  (Rounds 1 - 5 zeroed the strip and let every wave ds_add_f32 into it: 21 LDS atomics per lane and strip at ~190 LDS
        // cycles per wave instruction - the LDS was 96 % busy and 0.93 ms of the 1.45-ms launch remained with ALL arithmetic compiled out,
        // profiles/r06b_decode_probe.txt, r06b_decode_pmc2.json.)
        const int c = c0 + lane;
        const bool live = c < W;                 // (lanes past the map's last column still write their zeros)
        const int cc = live ? c : W - 1;
        const int guest_from = j1 < h ? row_base[j1] : h;   // first row the wave below shares (wave-uniform); rows from here on are its to store
        float acc[TY];
        int base = row_base[j0 < h ? j0 : 0];
        if (!(LP_DEC_PROBE & 8) && j0 < j1) {    // (wave-uniform)
            // (the lane's 12 column taps are re-read - three 16-B loads from L1 - for each of the one or two window rows a row group adds,
            //  instead of living in 12 registers beside the two windows and the NE gradient accumulators: with them the NE = 18
            //  instantiations, capped at 128 registers for two workgroups per CU, spilled 15 - 48 registers)
            const float* txg = tb.col_taps + (size_t)cc * kTXM;
            const float* hcol = hs + tb.col_start[cc];
            // the plain kernels keep the 12 taps in registers (with the row groups walked two output rows at a time they fit: 127 of 128); the
            // pruning ones, which carry their bounds as well, re-read them per use (measured, profiles/r05d_decode_variants.txt)
            constexpr bool kTxRegs = LP_DEC_TX_REGS == 1 || (LP_DEC_TX_REGS == 2 && !PRUNE && R == 4);   // (ds = 1 / 3 tables: the re-read form, one register short otherwise)
            float txr[kTxRegs ? kTXM : 1];
            if (kTxRegs) {
#pragma unroll
                for (int t = 0; t < (kTxRegs ? kTXM : 0); ++t) txr[kTxRegs ? t : 0] = txg[t];
            }
            auto zv = [&](int r) __attribute__((always_inline)) {
                if constexpr ((LP_DEC_PROBE & 4) != 0) return hcol[r];
                if constexpr (kTxRegs) return z_value<FULLTX>(hcol, r, w, txr, tb.TX);
                const float* tp = txg;
                LP_OPAQUE(tp);
                float tx[kTXM];
#pragma unroll
                for (int t4 = 0; t4 < kTXM / 4; ++t4) {
                    const f32x4 v4 = *reinterpret_cast<const f32x4*>(tp + 4 * t4);
                    tx[4 * t4] = v4[0], tx[4 * t4 + 1] = v4[1], tx[4 * t4 + 2] = v4[2], tx[4 * t4 + 3] = v4[3];
                }
                return z_value<FULLTX>(hcol, r, w, tx, tb.TX);
            };
            const float dxc = live ? gxt * (((float)c - x0) - dex) : 0.f;
            const float gyl = live ? gyt : 0.f;
            float win[TY];
            bool have = false;  // is `win` the window of `base`? (acc always belongs to `base`)
#pragma unroll
            for (int t = 0; t < TY; ++t) acc[t] = 0.f;
            for (int j = j0; j < j1; ++j) {
                const int nb = row_base[j];
                const bool skip = prune && ly * lx * ps.rwin[j] < cut;  // wave-uniform: no row of this group reaches the cut
                if (nb != base) {
                    zs[base * LD + lane] = acc[0];  // this segment is done with input row `base` (base < guest_from: its owner)
#pragma unroll
                    for (int t = 0; t < TY - 1; ++t) {
                        win[t] = win[t + 1];
                        acc[t] = acc[t + 1];
                    }
                    if (have && !skip) win[TY - 1] = zv(nb + TY - 1);
                    acc[TY - 1] = 0.f;
                    base = nb;
                }
                if (skip) {
                    have = false;
                    continue;
                }
                if (!have) {
#pragma unroll
                    for (int t = 0; t < TY; ++t) win[t] = zv(base + t);
                    have = true;
                }
                const auto taps = row_taps + (size_t)j * R * TY;
                // T (gx (c - E[x]) + gy (row j R - E[y])): the row group's part, once.  (R = 2 - downsample_factor 1 - keeps the per-row form: with one
                // more value alive across its row loop that instantiation spilled a register, tests/test_kernel_resources.py)
                constexpr bool kGroupCoef = R != 2;
                const float gj = kGroupCoef ? fmaf(gyl, ((float)(j * R) - y0) - dey, dxc) : 0.f;
                // (one output row at a time for the widest tables: unrolled, the R x TY taps of a row group - 88 scalars at ds = 3 - do not
                //  fit the scalar file and come back as spilled VECTOR registers)
#pragma unroll (R * TY <= LP_DEC_RR_FULL ? R : LP_DEC_RR_PART)
                for (int rr = 0; rr < ((LP_DEC_PROBE & 2) ? 0 : R); ++rr) {
                    float y = 0.f;
#pragma unroll
                    for (int t = 0; t < TY; ++t) y = fmaf(taps[rr * TY + t], win[t], y);
                    const float p = __expf(y * temperature - m) * inv_s;
                    const float g = kGroupCoef ? p * fmaf(gyl, (float)rr, gj) : p * (dxc + gyl * (((float)(j * R + rr) - y0) - dey));
#pragma unroll
                    for (int t = 0; t < TY; ++t) acc[t] = fmaf(taps[rr * TY + t], g, acc[t]);
                }
            }
#pragma unroll
            for (int t = 0; t < TY; ++t)   // the rows still in the window that nobody below shares: this wave's to store
                if (base + t < guest_from) zs[(base + t) * LD + lane] = acc[t];
        }
        __syncthreads();   // every owner has stored
        // ... and the shared rows are added to what their owner stored: even waves, then odd waves.  (Neighbours only share with each other for
        // every map size the trackers produce; where the clamped windows at the map's edge make THREE consecutive waves meet in a row - h = 65 at
        // downsample_factor 1 is the one case up to 256 - its two guests are neighbours, hence of different parity, hence in different phases.
        // ops.py refuses a table whose rows have four contributors.)
#pragma unroll
        for (int phase = 0; phase < 2; ++phase) {
            if (!(LP_DEC_PROBE & 8) && j0 < j1 && (wave & 1) == phase) {
#pragma unroll
                for (int t = 0; t < TY; ++t)
                    if (base + t >= guest_from) zs[(base + t) * LD + lane] += acc[t];
            }
            __syncthreads();
        }
        // dH[r][q] += sum_c Wst[r][c] * Ux[c][q]  over this strip's columns (columns past W hold zeros)
#pragma unroll
        for (int i = 0; i < ((LP_DEC_PROBE & 1) ? 0 : NE); ++i) {
            int e = tid + i * nthreads;
            LP_OPAQUE(e);  // (q, r, cs) are recomputed per strip instead of living in registers per element
            if (e < h * w) {
                const int q = (int)(((float)e + 0.5f) * inv_h), r = e - q * h;  // exact for e < 2^22
                const int cs = lcs[q];
                int t0 = c0 - cs;
                if (t0 < 0) t0 = 0;
                int t1 = c0 + SC - cs;
                if (t1 > tb.TC) t1 = tb.TC;
                const float* zr = zs + r * LD + (cs - c0);
                const float* tq = lt + q * TCP;
                float a = dacc[i];
#pragma unroll 4
                for (int t = t0; t < t1; ++t) a = fmaf(zr[t], tq[t], a);
                dacc[i] = a;
            }
        }
        __syncthreads();
    }
    // transpose through LDS (the tile AND the strip behind it are dead) so the map leaves in full rows.  Rows padded to an odd pitch: consecutive
    // lanes hold consecutive rows r of one column, and with the tile's own pitch w (96 words: a multiple of the 32 banks) all 32 lanes of a
    // ds_write group hit ONE bank (round 6: ~9 k LDS cycles per map); pitch w | 1 spreads them over 32 banks.  h * (w | 1) <= h * (w + 65) floats.
    const int wp = w | 1;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int e = tid + i * nthreads;
        if (e < h * w) {
            const int q = (int)(((float)e + 0.5f) * inv_h), r = e - q * h;
            hs[r * wp + q] = dacc[i];
        }
    }
    __syncthreads();
    float* dst = g_heat + (size_t)bk * h * w;
    const float inv_w = 1.f / (float)w;
    for (int i = tid; i < h * w; i += nthreads) {
        const int r = (int)(((float)i + 0.5f) * inv_w);   // exact for i < 2^22
        const float v = hs[r * wp + (i - r * w)];
        dst[i] = accumulate ? dst[i] + v : v;
    }
}

// standalone keypoint epilogue (targets, or callers that decode elsewhere): kp_out = frame_map(kp_in); inverse-transpose
// for the gradient
__global__ __launch_bounds__(256) void frame_map_kernel(const float* __restrict__ in, int n, int K, FrameMap fm, int backward,
                                                        float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int b = i / K, k = i - b * K;
    float x, y;
    if (backward) frame_grad_to_aug(in[i * 2], in[i * 2 + 1], b, k, fm, x, y);
    else to_frame(in[i * 2], in[i * 2 + 1], b, k, fm, x, y);
    out[i * 2] = x;
    out[i * 2 + 1] = y;
}

static size_t decode_smem_bytes(int h, int w, int strip_cols) { return (size_t)(h * w + h * strip_cols) * sizeof(float); }

// forward: one lane per output column, whole waves, at most 512 threads (wider maps take several trips)
static int decode_fwd_threads(int W) {
    const int t = (W + 63) / 64 * 64;
    return t > 512 ? 512 : t;
}

// backward: as many waves (= row segments) as keep a segment at least one window tall, at most 8
static int decode_bwd_threads(int h, int TY) {
    int waves = h / TY;
    waves = waves < 1 ? 1 : (waves > 8 ? 8 : waves);
    return waves * 64;
}

// Exact pruning of the T = 1000 soft-argmax (kernels above): pays on the peaked maps of a trained head (forward 1.4x, backward 1.8x),
// costs on the FLAT maps of an untrained network, where nothing can be skipped and the pruning state costs one resident workgroup per CU
// (fwd 94 vs 71 VGPRs) - DESIGN.md section 4.2 for the measured numbers.  The CALLER decides, per call: the `prune` argument of
// lp_decode_fwd / lp_decode_bwd (round 5; until round 4 a process-wide switch, lp_decode_set_prune + LP_DECODE_PRUNE).  The product's
// default is automatic: ops.py watches the decode's own sum-of-exponentials output, a direct measure of how many pixels carry weight.
template <typename Kern>
static void allow_large_lds(Kern kern, size_t bytes) {
    // opt in to > 64 KiB of dynamic LDS (gfx950 has 160 KiB per workgroup)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace lp

extern "C" int lp_decode_window(int downsample_factor, int n) {
    // window (taps per output-row group) the kernels are instantiated for; host tables must use it
    switch (downsample_factor) {
        case 1: return n >= 8 ? 8 : 0;
        case 2: return n >= 9 ? 9 : (n == 8 ? 8 : 0);
        case 3: return n >= 11 ? 11 : 0;
        default: return 0;
    }
}

extern "C" int lp_decode_fwd(const float* heat, int B, int K, int h, int w, int downsample_factor, float temperature,
                             const lp_decode_tables* t, const lp_frame_map* f, float* kp_aug, float* kp_frame, float* conf,
                             float* stats, int prune_mode, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(heat && t && f && kp_aug && kp_frame && conf && stats);
    LP_REQUIRE(B >= 0 && K > 0 && h > 0 && w > 0);
    if (B == 0) return LP_OK;
    const int TY = lp_decode_window(downsample_factor, h);
    if (TY == 0 || t->ty != TY || t->tx > kTXM || t->tx > w) return LP_ERR_UNSUPPORTED;
    const int R = 1 << downsample_factor;
    DecodeTables tb{t->row_base, t->row_taps, t->col_start, t->col_taps, t->colT_start, t->colT_taps, t->tx, t->tc};
    FrameMap fm{f->transforms, f->tf_mode, f->bbox, f->bbox_stride, f->kp_per_view, f->model_h, f->model_w};
    const float offset = downsample_factor == 1 ? 0.5f : downsample_factor == 2 ? 1.5f : 2.5f;
    const size_t smem = decode_smem_bytes(h, w, 0);
    if (smem > 150 * 1024) return LP_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(B * K), block(decode_fwd_threads(w * R));
    const bool full = t->tx == kTXM;
    const bool prune = prune_mode != 0 && h <= kPruneMaxDim && w <= kPruneMaxDim;
#define LP_LAUNCH_FWD2(RR, TT, FF, PP)                                                                                      \
    allow_large_lds(decode_fwd_kernel<RR, TT, FF, PP>, smem + (PP ? sizeof(PruneState) : 0));                               \
    hipLaunchKernelGGL((decode_fwd_kernel<RR, TT, FF, PP>), grid, block, smem + (PP ? sizeof(PruneState) : 0), st, heat, K, h, w, \
                       temperature, offset, tb, fm, kp_aug, kp_frame, conf, stats)
#define LP_LAUNCH_FWD(RR, TT)                                     \
    do {                                                          \
        if (prune) {                                              \
            if (full) { LP_LAUNCH_FWD2(RR, TT, true, true); }     \
            else { LP_LAUNCH_FWD2(RR, TT, false, true); }         \
        } else {                                                  \
            if (full) { LP_LAUNCH_FWD2(RR, TT, true, false); }    \
            else { LP_LAUNCH_FWD2(RR, TT, false, false); }        \
        }                                                         \
    } while (0)
    if (R == 2 && TY == 8) { LP_LAUNCH_FWD(2, 8); }
    else if (R == 4 && TY == 8) { LP_LAUNCH_FWD(4, 8); }
    else if (R == 4 && TY == 9) { LP_LAUNCH_FWD(4, 9); }
    else if (R == 8 && TY == 11) { LP_LAUNCH_FWD(8, 11); }
    else return LP_ERR_UNSUPPORTED;
#undef LP_LAUNCH_FWD
#undef LP_LAUNCH_FWD2
    return launch_status();
}

extern "C" int lp_decode_bwd(const float* heat, int B, int K, int h, int w, int downsample_factor, float temperature,
                             const lp_decode_tables* t, const lp_frame_map* f, const float* stats, const float* g_aug,
                             const float* g_frame, float* g_heat, int accumulate, int prune_mode, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(heat && t && f && stats && g_heat && (g_aug || g_frame));
    LP_REQUIRE(B >= 0 && K > 0 && h > 0 && w > 0);
    if (B == 0) return LP_OK;
    const int TY = lp_decode_window(downsample_factor, h);
    if (TY == 0 || t->ty != TY || t->tx > kTXM || t->tx > w || !t->colT_start || !t->colT_taps) return LP_ERR_UNSUPPORTED;
    const int R = 1 << downsample_factor;
    const int nthreads = decode_bwd_threads(h, TY);
    const int ne = (h * w + nthreads - 1) / nthreads;
    if (ne > 64) return LP_ERR_UNSUPPORTED;
    DecodeTables tb{t->row_base, t->row_taps, t->col_start, t->col_taps, t->colT_start, t->colT_taps, t->tx, t->tc};
    FrameMap fm{f->transforms, f->tf_mode, f->bbox, f->bbox_stride, f->kp_per_view, f->model_h, f->model_w};
    const size_t smem = ((size_t)h * w + (size_t)h * kBwdLd + (size_t)w * bwd_tap_ld(t->tc) + (size_t)w) * sizeof(float);
    if (smem > 150 * 1024) return LP_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(B * K), block(nthreads);
    const bool full = t->tx == kTXM;
    const bool prune = prune_mode != 0 && h <= kPruneMaxDim && w <= kPruneMaxDim;
#define LP_LAUNCH_BWD(RR, TT, NN, FF, PP)                                                                                   \
    allow_large_lds(decode_bwd_kernel<RR, TT, NN, FF, PP>, smem + (PP ? sizeof(PruneState) : 0));                           \
    hipLaunchKernelGGL((decode_bwd_kernel<RR, TT, NN, FF, PP>), grid, block, smem + (PP ? sizeof(PruneState) : 0), st, heat, K, h, w, \
                       temperature, tb, fm, stats, g_aug, g_frame, g_heat, accumulate)
#define LP_DISPATCH_NE2(RR, TT, PP)                                  \
    do {                                                             \
        /* (rounds 1 - 5 also instantiated NE = 8 for maps of up to 64 x 64: the same 128-register cap and occupancy as NE = 18, whose ten */ \
        /*  spare accumulators cost nothing - and in round 6's form the NE = 8 pruning kernels spilled 5 - 7 registers where NE = 18 has none) */ \
        if (full) {                                                  \
            if (ne <= 18) { LP_LAUNCH_BWD(RR, TT, 18, true, PP); }   \
            else { LP_LAUNCH_BWD(RR, TT, 64, true, PP); }            \
        } else {                                                     \
            if (ne <= 18) { LP_LAUNCH_BWD(RR, TT, 18, false, PP); }  \
            else { LP_LAUNCH_BWD(RR, TT, 64, false, PP); }           \
        }                                                            \
    } while (0)
#define LP_DISPATCH_NE(RR, TT)                       \
    do {                                             \
        if (prune) { LP_DISPATCH_NE2(RR, TT, true); } \
        else { LP_DISPATCH_NE2(RR, TT, false); }     \
    } while (0)
    if (R == 2 && TY == 8) { LP_DISPATCH_NE(2, 8); }
    else if (R == 4 && TY == 8) { LP_DISPATCH_NE(4, 8); }
    else if (R == 4 && TY == 9) { LP_DISPATCH_NE(4, 9); }
    else if (R == 8 && TY == 11) { LP_DISPATCH_NE(8, 11); }
    else return LP_ERR_UNSUPPORTED;
#undef LP_DISPATCH_NE
#undef LP_DISPATCH_NE2
#undef LP_LAUNCH_BWD
    return launch_status();
}

extern "C" int lp_frame_map_apply(const float* kp_in, int B, int K, const lp_frame_map* f, int backward, float* kp_out,
                                  lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(kp_in && kp_out && f && B >= 0 && K > 0);
    if (B == 0) return LP_OK;
    FrameMap fm{f->transforms, f->tf_mode, f->bbox, f->bbox_stride, f->kp_per_view, f->model_h, f->model_w};
    hipLaunchKernelGGL(frame_map_kernel, dim3((B * K + 255) / 256), dim3(256), 0, (hipStream_t)stream, kp_in, B * K, K, fm, backward,
                       kp_out);
    return launch_status();
}
