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
// (the only HBM read: h*w*4 algorithmic bytes), Z = Hm * Ux^T is produced per strip of 256 output columns
// in LDS, and each lane walks DOWN one output column keeping a sliding window of Z in registers, so Y is
// never stored anywhere.  The softmax(T*Y) expectation is accumulated online (running max / sum / sum*x /
// sum*y per lane, merged with wave shuffles).  The 25-tap confidence window is recomputed from the LDS
// tile once the global max and sum are known.  Backward recomputes Y the same way, forms
// G = T*p*(gx*(x-ex) + gy*(y-ey)) on the fly and applies the transposed operators in place in LDS.
//
// This kernel is fp32-VALU bound by construction (about 350 FLOP per algorithmic byte): see DESIGN.md.
#include "lp_common.h"

namespace lp {

constexpr int kTXM = 12;        // padded column-tap count
constexpr int kStripCols = 256; // output columns per strip = 4 waves x 64 lanes

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

// Z strip: zs[r][cl] = sum_t Hm[r][xs+t] * tap[t]   for the strip's columns (one column per thread).
__device__ __forceinline__ void build_z_strip(const float* hs, float* zs, int h, int w, int W, int c0, int strip_cols,
                                              const DecodeTables& tb) {
    const int cl = threadIdx.x;
    const int c = c0 + cl;
    if (cl < strip_cols && c < W) {
        float tx[kTXM];
        const int xs = tb.col_start[c];
#pragma unroll
        for (int t = 0; t < kTXM; ++t) tx[t] = tb.col_taps[c * kTXM + t];
        for (int r = 0; r < h; ++r) {
            const float* row = hs + r * w + xs;
            float acc = 0.f;
#pragma unroll
            for (int t = 0; t < kTXM; ++t)
                if (t < tb.TX) acc = fmaf(row[t], tx[t], acc);
            zs[r * strip_cols + cl] = acc;
        }
    }
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
// forward
// ------------------------------------------------------------------------------------------------
template <int R, int TY>
__global__ __launch_bounds__(256) void decode_fwd_kernel(const float* __restrict__ heat, int K, int h, int w, float temperature,
                                                         float offset, DecodeTables tb, FrameMap fm, int strip_cols,
                                                         float* __restrict__ kp_aug, float* __restrict__ kp_frame,
                                                         float* __restrict__ conf, float* __restrict__ stats) {
    HIP_DYNAMIC_SHARED(float, smem)
    float* hs = smem;                 // [h][w]
    float* zs = smem + h * w;         // [h][strip_cols]
    __shared__ float red[4 * 4 + 8];

    const int bk = blockIdx.x;
    const int b = bk / K, k = bk - b * K;
    const int H = h * R, W = w * R;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const float* src = heat + (size_t)bk * h * w;
    for (int i = tid; i < h * w; i += 256) hs[i] = src[i];
    __syncthreads();

    float m = -INFINITY, s = 0.f, sx = 0.f, sy = 0.f;
    for (int c0 = 0; c0 < W; c0 += strip_cols) {
        build_z_strip(hs, zs, h, w, W, c0, strip_cols, tb);
        __syncthreads();
        const int cl = wave * 64 + lane;
        const int c = c0 + cl;
        // Every lane runs the column walk (lanes past the map re-read column 0 and are discarded afterwards): with no
        // divergent branch around it the tap table is fetched with wave-uniform SCALAR loads instead of per-lane vector loads.
        const bool valid = cl < strip_cols && c < W;
        const int clc = valid ? cl : 0;
        const float m_in = m, s_in = s, sx_in = sx, sy_in = sy;
        {
            const float xc = (float)c;
            float win[TY];
            int base = tb.row_base[0];
#pragma unroll
            for (int t = 0; t < TY; ++t) win[t] = zs[(base + t) * strip_cols + clc];
            for (int j = 0; j < h; ++j) {
                const int nb = tb.row_base[j];
                if (nb != base) {  // windows advance by exactly one input row (host asserts it)
#pragma unroll
                    for (int t = 0; t < TY - 1; ++t) win[t] = win[t + 1];
                    win[TY - 1] = zs[(nb + TY - 1) * strip_cols + clc];
                    base = nb;
                }
                const float* taps = tb.row_taps + (size_t)j * R * TY;
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
                const float sc = (m == -INFINITY) ? 0.f : __expf(m - gm);
                s *= sc;
                sx *= sc;
                sy *= sc;
                m = gm;
#pragma unroll
                for (int rr = 0; rr < R; ++rr) {
                    const float e = __expf(z[rr] - gm);
                    s += e;
                    sx = fmaf(e, xc, sx);
                    sy = fmaf(e, (float)(j * R + rr), sy);
                }
            }
        }
        if (!valid) {
            m = m_in;
            s = s_in;
            sx = sx_in;
            sy = sy_in;
        }
        __syncthreads();  // zs is rebuilt by the next strip
    }

    // merge the per-lane online-softmax states: wave shuffles, then across the 4 waves through LDS
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
#pragma unroll
    for (int i = 1; i < 4; ++i) merge_softmax(m, s, sx, sy, red[i * 4 + 0], red[i * 4 + 1], red[i * 4 + 2], red[i * 4 + 3]);
    const float ex = sx / s, ey = sy / s;

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
            stats[bk * 4 + 2] = ex;
            stats[bk * 4 + 3] = ey;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward: d(loss)/d(heat) from d(loss)/d(kp_aug) (+ d(loss)/d(kp_frame) chained through the frame map)
// ------------------------------------------------------------------------------------------------
template <int R, int TY, int NE>
__global__ __launch_bounds__(256) void decode_bwd_kernel(const float* __restrict__ heat, int K, int h, int w, float temperature,
                                                         DecodeTables tb, FrameMap fm, int strip_cols,
                                                         const float* __restrict__ stats, const float* __restrict__ g_aug,
                                                         const float* __restrict__ g_frame, float* __restrict__ g_heat,
                                                         int accumulate) {
    HIP_DYNAMIC_SHARED(float, smem)
    float* hs = smem;
    float* zs = smem + h * w;  // Z strip, overwritten in place by the W = Uy^T G strip

    const int bk = blockIdx.x;
    const int b = bk / K, k = bk - b * K;
    const int W = w * R;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const float* src = heat + (size_t)bk * h * w;
    for (int i = tid; i < h * w; i += 256) hs[i] = src[i];

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
    const float ex = stats[bk * 4 + 2], ey = stats[bk * 4 + 3];
    const float gxt = gx * temperature, gyt = gy * temperature;

    float dacc[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) dacc[i] = 0.f;
    __syncthreads();

    for (int c0 = 0; c0 < W; c0 += strip_cols) {
        build_z_strip(hs, zs, h, w, W, c0, strip_cols, tb);
        __syncthreads();
        const int cl = wave * 64 + lane;
        const int c = c0 + cl;
        const bool valid = cl < strip_cols && c < W;  // see the forward kernel: the walk itself is branch-free
        const int clc = valid ? cl : 0;
        {
            const float dxc = gxt * ((float)c - ex);
            float win[TY], acc[TY];
            int base = tb.row_base[0];
#pragma unroll
            for (int t = 0; t < TY; ++t) {
                win[t] = zs[(base + t) * strip_cols + clc];
                acc[t] = 0.f;
            }
            for (int j = 0; j < h; ++j) {
                const int nb = tb.row_base[j];
                if (nb != base) {
                    if (valid) zs[base * strip_cols + cl] = acc[0];  // input row `base` is complete and no longer read
#pragma unroll
                    for (int t = 0; t < TY - 1; ++t) {
                        win[t] = win[t + 1];
                        acc[t] = acc[t + 1];
                    }
                    win[TY - 1] = zs[(nb + TY - 1) * strip_cols + clc];
                    acc[TY - 1] = 0.f;
                    base = nb;
                }
                const float* taps = tb.row_taps + (size_t)j * R * TY;
#pragma unroll
                for (int rr = 0; rr < R; ++rr) {
                    float y = 0.f;
#pragma unroll
                    for (int t = 0; t < TY; ++t) y = fmaf(taps[rr * TY + t], win[t], y);
                    const float p = __expf(y * temperature - m) * inv_s;
                    const float g = p * (dxc + gyt * ((float)(j * R + rr) - ey));
#pragma unroll
                    for (int t = 0; t < TY; ++t) acc[t] = fmaf(taps[rr * TY + t], g, acc[t]);
                }
            }
            if (valid) {
#pragma unroll
                for (int t = 0; t < TY; ++t) zs[(base + t) * strip_cols + cl] = acc[t];
            }
        }
        if (!valid && cl < strip_cols) {
            for (int r = 0; r < h; ++r) zs[r * strip_cols + cl] = 0.f;  // columns past W contribute nothing
        }
        __syncthreads();
        // dH[r][q] += sum_c Wst[r][c] * Ux[c][q]  over this strip's columns
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = tid + i * 256;
            if (e < h * w) {
                const int r = e / w, q = e - r * w;
                const int cs = tb.colT_start[q];
                const float* tt = tb.colT_taps + (size_t)q * tb.TC;
                int t0 = c0 - cs;
                if (t0 < 0) t0 = 0;
                int t1 = c0 + strip_cols - cs;
                if (t1 > tb.TC) t1 = tb.TC;
                float a = dacc[i];
                for (int t = t0; t < t1; ++t) a = fmaf(zs[r * strip_cols + (cs + t - c0)], tt[t], a);
                dacc[i] = a;
            }
        }
        __syncthreads();
    }
    float* dst = g_heat + (size_t)bk * h * w;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int e = tid + i * 256;
        if (e < h * w) dst[e] = accumulate ? dst[e] + dacc[i] : dacc[i];
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

// Largest strip (256/128/64 output columns) whose Z buffer fits next to the heatmap tile in LDS, trimmed to the map.
static int pick_strip(int h, int w, int W) {
    int sc = 0;
    for (int cand = kStripCols; cand >= 64; cand >>= 1) {
        if (decode_smem_bytes(h, w, cand) <= 150 * 1024) {
            sc = cand;
            break;
        }
    }
    if (sc == 0) return 0;
    const int need = (W + 63) / 64 * 64;
    return need < sc ? need : sc;
}

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
                             float* stats, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(heat && t && f && kp_aug && kp_frame && conf && stats);
    LP_REQUIRE(B >= 0 && K > 0 && h > 0 && w > 0);
    if (B == 0) return LP_OK;
    const int TY = lp_decode_window(downsample_factor, h);
    if (TY == 0 || t->ty != TY || t->tx > kTXM || t->tx > w) return LP_ERR_UNSUPPORTED;
    const int R = 1 << downsample_factor;
    const int sc = pick_strip(h, w, w * R);
    if (sc == 0) return LP_ERR_UNSUPPORTED;
    DecodeTables tb{t->row_base, t->row_taps, t->col_start, t->col_taps, t->colT_start, t->colT_taps, t->tx, t->tc};
    FrameMap fm{f->transforms, f->tf_mode, f->bbox, f->bbox_stride, f->kp_per_view, f->model_h, f->model_w};
    const float offset = downsample_factor == 1 ? 0.5f : downsample_factor == 2 ? 1.5f : 2.5f;
    const size_t smem = decode_smem_bytes(h, w, sc);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(B * K), block(256);
#define LP_LAUNCH_FWD(RR, TT)                                                                                           \
    allow_large_lds(decode_fwd_kernel<RR, TT>, smem);                                                                   \
    hipLaunchKernelGGL((decode_fwd_kernel<RR, TT>), grid, block, smem, st, heat, K, h, w, temperature, offset, tb, fm, sc, \
                       kp_aug, kp_frame, conf, stats)
    if (R == 2 && TY == 8) { LP_LAUNCH_FWD(2, 8); }
    else if (R == 4 && TY == 8) { LP_LAUNCH_FWD(4, 8); }
    else if (R == 4 && TY == 9) { LP_LAUNCH_FWD(4, 9); }
    else if (R == 8 && TY == 11) { LP_LAUNCH_FWD(8, 11); }
    else return LP_ERR_UNSUPPORTED;
#undef LP_LAUNCH_FWD
    return launch_status();
}

extern "C" int lp_decode_bwd(const float* heat, int B, int K, int h, int w, int downsample_factor, float temperature,
                             const lp_decode_tables* t, const lp_frame_map* f, const float* stats, const float* g_aug,
                             const float* g_frame, float* g_heat, int accumulate, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(heat && t && f && stats && g_heat && (g_aug || g_frame));
    LP_REQUIRE(B >= 0 && K > 0 && h > 0 && w > 0);
    if (B == 0) return LP_OK;
    const int TY = lp_decode_window(downsample_factor, h);
    if (TY == 0 || t->ty != TY || t->tx > kTXM || t->tx > w || !t->colT_start || !t->colT_taps) return LP_ERR_UNSUPPORTED;
    const int R = 1 << downsample_factor;
    const int sc = pick_strip(h, w, w * R);
    if (sc == 0) return LP_ERR_UNSUPPORTED;
    const int ne = (h * w + 255) / 256;
    if (ne > 64) return LP_ERR_UNSUPPORTED;
    DecodeTables tb{t->row_base, t->row_taps, t->col_start, t->col_taps, t->colT_start, t->colT_taps, t->tx, t->tc};
    FrameMap fm{f->transforms, f->tf_mode, f->bbox, f->bbox_stride, f->kp_per_view, f->model_h, f->model_w};
    const size_t smem = decode_smem_bytes(h, w, sc);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(B * K), block(256);
#define LP_LAUNCH_BWD(RR, TT, NN)                                                                                       \
    allow_large_lds(decode_bwd_kernel<RR, TT, NN>, smem);                                                               \
    hipLaunchKernelGGL((decode_bwd_kernel<RR, TT, NN>), grid, block, smem, st, heat, K, h, w, temperature, tb, fm, sc, stats, \
                       g_aug, g_frame, g_heat, accumulate)
#define LP_DISPATCH_NE(RR, TT)                   \
    do {                                         \
        if (ne <= 16) { LP_LAUNCH_BWD(RR, TT, 16); } \
        else if (ne <= 36) { LP_LAUNCH_BWD(RR, TT, 36); } \
        else { LP_LAUNCH_BWD(RR, TT, 64); }      \
    } while (0)
    if (R == 2 && TY == 8) { LP_DISPATCH_NE(2, 8); }
    else if (R == 4 && TY == 8) { LP_DISPATCH_NE(4, 8); }
    else if (R == 4 && TY == 9) { LP_DISPATCH_NE(4, 9); }
    else if (R == 8 && TY == 11) { LP_DISPATCH_NE(8, 11); }
    else return LP_ERR_UNSUPPORTED;
#undef LP_DISPATCH_NE
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
