// Heat-map side of the step: Gaussian target generation, masked heat-map MSE (targets supplied or generated
// on the fly for the unimodal loss) and the head's spatial softmax, forward + backward.  gfx950, fp32.
//
// Reference arithmetic (paths relative to the reference tree):
//   data/heatmaps.py:11-87                  generate_heatmaps
//   losses/losses.py:229-290,314-335        HeatmapLoss.remove_nans / HeatmapMSELoss.compute_loss
//   losses/losses.py:1129-1260              ReprojectionHeatmapLoss (template for unimodal_mse)
//   models/heads/heatmap.py:209-211         spatial_softmax2d(T=1) at the end of HeatmapHead.forward
//
// All of these are HBM-bound streaming kernels: one workgroup per (frame, keypoint) map, float4 accesses,
// wave-shuffle reductions, no host synchronisation (the masked-mean denominator stays on the device).
#include "lp_common.h"

namespace lp {

struct GaussSpec {
    float sx, sy;        // heat-map px per image px  (w / width, h / height)
    float inv_two_var;   // 1 / (2 sigma^2)
    int h, w;
};

// Centre of the Gaussian on the heat-map grid and the reference's validity rule.
__device__ __forceinline__ bool gauss_centre(float kx, float ky, const GaussSpec& g, float& cx, float& cy) {
    const float x = kx * g.sx, y = ky * g.sy;
    const bool bad = (x != x) || (x < -1.f) || (x > (float)g.w + 1.f) || (y < -1.f) || (y > (float)g.h + 1.f);
    cx = fminf(fmaxf(x, -1.f), (float)g.w + 1.f);
    cy = fminf(fmaxf(y, -1.f), (float)g.h + 1.f);
    return !bad;
}

__device__ __forceinline__ float gauss_at(int i, float cx, float cy, const GaussSpec& g) {
    const int r = i / g.w, c = i - r * g.w;
    const float dx = (float)c - cx, dy = (float)r - cy;
    return expf(-(dx * dx + dy * dy) * g.inv_two_var);
}

constexpr int kGaussAxisMax = 512;   // heat-map axes up to this length take the separable path of heatmap_gen_kernel

// ---- generate_heatmaps -----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void heatmap_gen_kernel(const float* __restrict__ kp, const int* __restrict__ vis, GaussSpec g,
                                                          float* __restrict__ out) {
    __shared__ float red[4];
    __shared__ float ex[kGaussAxisMax], ey[kGaussAxisMax];
    const int bk = blockIdx.x, n = g.h * g.w;
    float cx, cy;
    const bool ok = gauss_centre(kp[bk * 2], kp[bk * 2 + 1], g, cx, cy);
    int mode = ok ? 2 : 0;  // 2 gaussian, 1 uniform, 0 zeros
    if (vis != nullptr) {
        const int v = vis[bk];
        if (v == 0) mode = 0;
        else if (v == 1) mode = 1;
        // v == 2: gaussian unless out of bounds / NaN
    }
    float* dst = out + (size_t)bk * n;
    if (mode != 2) {
        const float val = mode == 1 ? 1.f / (float)n : 0.f;
        for (int i = threadIdx.x; i < n; i += 256) dst[i] = val;
        return;
    }
    // The Gaussian factors into a row and a column profile: h + w exponentials per map instead of 2 h w, and the map's sum is the product of
    // the two profile sums.  (exp(-(dx^2 + dy^2) k) vs exp(-dx^2 k) exp(-dy^2 k): a few fp32 ulp, far inside the 2e-7 the golden test allows.)
    if (g.h <= kGaussAxisMax && g.w <= kGaussAxisMax) {
        float px = 0.f, py = 0.f;
        for (int i = threadIdx.x; i < g.w; i += 256) {
            const float d = (float)i - cx;
            px += ex[i] = expf(-(d * d) * g.inv_two_var);
        }
        for (int i = threadIdx.x; i < g.h; i += 256) {
            const float d = (float)i - cy;
            py += ey[i] = expf(-(d * d) * g.inv_two_var);
        }
        const float sx = block_sum<4>(px, red);   // (block_sum's barriers also publish ex / ey)
        const float sy = block_sum<4>(py, red);
        const float inv = 1.f / (sx * sy);
        if ((g.w & 3) == 0) {
            const int w4 = g.w >> 2;
            for (int i = threadIdx.x; i < (n >> 2); i += 256) {
                const int r = i / w4, c = (i - r * w4) << 2;
                const float sr = ey[r] * inv;
                float4 v;
                v.x = ex[c] * sr; v.y = ex[c + 1] * sr; v.z = ex[c + 2] * sr; v.w = ex[c + 3] * sr;
                *reinterpret_cast<float4*>(dst + (size_t)i * 4) = v;
            }
        } else {
            for (int i = threadIdx.x; i < n; i += 256) {
                const int r = i / g.w, c = i - r * g.w;
                dst[i] = ex[c] * (ey[r] * inv);
            }
        }
        return;
    }
    float part = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) part += gauss_at(i, cx, cy, g);
    const float total = block_sum<4>(part, red);
    for (int i = threadIdx.x; i < n; i += 256) dst[i] = gauss_at(i, cx, cy, g) / total;
}

// ---- generate_heatmaps, backward (keep_gradients=True with a live graph: data/heatmaps.py:37-40 keeps the keypoints attached) -----------
// H = E / S with E[r][c] = exp(-((c - cx)^2 + (r - cy)^2) / (2 sigma^2)):  dH/dcx = H ((c - cx) - m_x) / sigma^2, m_x = sum_c ex[c] (c - cx) / sum ex.
// d loss / d keypoint = (that contracted with the incoming gradient) * heat-map px per image px.  Zero / uniform maps (NaN, out of bounds,
// visibility < 2) are constants; inside [-1, w + 1] the reference's clamp is the identity.
__global__ __launch_bounds__(256) void heatmap_gen_bwd_kernel(const float* __restrict__ kp, const int* __restrict__ vis, GaussSpec g,
                                                              const float* __restrict__ gout, float* __restrict__ gkp) {
    __shared__ float red[4];
    __shared__ float ex[kGaussAxisMax], ey[kGaussAxisMax];
    const int bk = blockIdx.x, n = g.h * g.w;
    float cx, cy;
    const bool ok = gauss_centre(kp[bk * 2], kp[bk * 2 + 1], g, cx, cy);
    const bool gaussian = ok && (vis == nullptr || vis[bk] == 2);
    if (!gaussian) {   // (workgroup-uniform)
        if (threadIdx.x < 2) gkp[bk * 2 + threadIdx.x] = 0.f;
        return;
    }
    float px = 0.f, py = 0.f, qx = 0.f, qy = 0.f;
    for (int i = threadIdx.x; i < g.w; i += 256) {
        const float d = (float)i - cx, e = expf(-(d * d) * g.inv_two_var);
        ex[i] = e;
        px += e;
        qx = fmaf(e, d, qx);
    }
    for (int i = threadIdx.x; i < g.h; i += 256) {
        const float d = (float)i - cy, e = expf(-(d * d) * g.inv_two_var);
        ey[i] = e;
        py += e;
        qy = fmaf(e, d, qy);
    }
    const float sx = block_sum<4>(px, red), sy = block_sum<4>(py, red);
    const float mx = block_sum<4>(qx, red) / sx, my = block_sum<4>(qy, red) / sy;
    const float inv = 1.f / (sx * sy);
    const float* gsrc = gout + (size_t)bk * n;
    float a = 0.f, b = 0.f, t = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int r = i / g.w, c = i - r * g.w;
        const float gh = gsrc[i] * ex[c] * (ey[r] * inv);
        t += gh;
        a = fmaf(gh, (float)c - cx, a);
        b = fmaf(gh, (float)r - cy, b);
    }
    a = block_sum<4>(a, red);
    b = block_sum<4>(b, red);
    t = block_sum<4>(t, red);
    if (threadIdx.x == 0) {
        const float k2 = 2.f * g.inv_two_var;   // 1 / sigma^2
        gkp[bk * 2] = (a - mx * t) * k2 * g.sx;
        gkp[bk * 2 + 1] = (b - my * t) * k2 * g.sy;
    }
}

// ---- masked heat-map losses (MSE / KL / JS) -----------------------------------------------------------
// per-pixel term of the three supervised heat-map losses (reference losses/losses.py:293-423; the divergences are kornia's
// kl_div_loss_2d / js_div_loss_2d on t + 1e-10, p + 1e-10, summed per map) and its derivative with respect to p
__device__ __forceinline__ float hm_term(int kind, float t, float p) {
    if (kind == LP_HM_MSE) {
        const float d = t - p;
        return d * d;
    }
    const float te = t + 1e-10f, pe = p + 1e-10f;
    if (kind == LP_HM_KL) return te * (logf(te) - logf(pe));
    const float lm = logf(0.5f * (te + pe));
    return 0.5f * te * (logf(te) - lm) + 0.5f * pe * (logf(pe) - lm);
}

__device__ __forceinline__ float hm_dterm(int kind, float t, float p) {
    if (kind == LP_HM_MSE) return 2.f * (p - t);
    const float te = t + 1e-10f, pe = p + 1e-10f;
    if (kind == LP_HM_KL) return -te / pe;
    return 0.5f * (logf(pe) - logf(0.5f * (te + pe)));
}

// pass 1: per map, the sum of the per-pixel terms and a validity flag.  FROM_KP: the target is the Gaussian at kp
// (unimodal loss, MSE only).
template <bool FROM_KP>
__global__ __launch_bounds__(256) void hm_rowsq_kernel(const float* __restrict__ targ, const float* __restrict__ pred,
                                                       const float* __restrict__ kp, const float* __restrict__ conf,
                                                       float prob_threshold, GaussSpec g, int kind, float* __restrict__ rowsum,
                                                       int* __restrict__ valid) {
    __shared__ float red[4];
    const int bk = blockIdx.x, n = g.h * g.w;
    const float* p = pred + (size_t)bk * n;
    float part = 0.f;
    bool ok;
    if (FROM_KP) {
        float cx, cy;
        ok = gauss_centre(kp[bk * 2], kp[bk * 2 + 1], g, cx, cy) && (conf[bk] >= prob_threshold);
        if (ok) {  // block-uniform
            float gs = 0.f;
            for (int i = threadIdx.x; i < n; i += 256) gs += gauss_at(i, cx, cy, g);
            const float inv = 1.f / block_sum<4>(gs, red);
            for (int i = threadIdx.x; i < n; i += 256) {
                const float d = gauss_at(i, cx, cy, g) * inv - p[i];
                part = fmaf(d, d, part);
            }
        }
    } else {
        const float* t = targ + (size_t)bk * n;
        float nz = 0.f;
        for (int i = threadIdx.x; i < n; i += 256) {
            const float tv = t[i];
            part += hm_term(kind, tv, p[i]);
            nz += (tv != 0.f) ? 1.f : 0.f;
        }
        ok = block_sum<4>(nz, red) > 0.f;
    }
    const float total = block_sum<4>(part, red);
    if (threadIdx.x == 0) {
        rowsum[bk] = ok ? total : 0.f;
        valid[bk] = ok ? 1 : 0;
    }
}

// pass 2 (one workgroup): loss = sum(rowsum over valid) / n_valid ; mean of an empty set is NaN as in torch,
// except for the unimodal loss, which is defined as 0 when nothing is kept.
__global__ __launch_bounds__(256) void hm_finish_kernel(const float* __restrict__ rowsum, const int* __restrict__ valid, int rows,
                                                        int zero_if_empty, float* __restrict__ loss, float* __restrict__ nvalid_out) {
    __shared__ float red[4];
    float s = 0.f, c = 0.f;
    for (int i = threadIdx.x; i < rows; i += 256) {
        s += rowsum[i];
        c += (float)valid[i];
    }
    s = block_sum<4>(s, red);
    c = block_sum<4>(c, red);
    if (threadIdx.x == 0) {
        loss[0] = (c == 0.f && zero_if_empty) ? 0.f : s / c;
        nvalid_out[0] = c;
    }
}

// backward: grad_pred = gout * d term / d p / n_valid on valid maps (MSE: 2 (p - t)), 0 elsewhere
template <bool FROM_KP>
__global__ __launch_bounds__(256) void hm_grad_kernel(const float* __restrict__ targ, const float* __restrict__ pred,
                                                      const float* __restrict__ kp, GaussSpec g, int kind,
                                                      const int* __restrict__ valid, const float* __restrict__ nvalid,
                                                      const float* __restrict__ gout, float* __restrict__ gpred, int accumulate) {
    __shared__ float red[4];
    const int bk = blockIdx.x, n = g.h * g.w;
    float* gp = gpred + (size_t)bk * n;
    if (!valid[bk]) {
        if (!accumulate)
            for (int i = threadIdx.x; i < n; i += 256) gp[i] = 0.f;
        return;
    }
    const float scale = gout[0] / nvalid[0];
    const float* p = pred + (size_t)bk * n;
    if (FROM_KP) {
        float cx, cy;
        gauss_centre(kp[bk * 2], kp[bk * 2 + 1], g, cx, cy);
        float gs = 0.f;
        for (int i = threadIdx.x; i < n; i += 256) gs += gauss_at(i, cx, cy, g);
        const float inv = 1.f / block_sum<4>(gs, red);
        for (int i = threadIdx.x; i < n; i += 256) {
            const float v = scale * 2.f * (p[i] - gauss_at(i, cx, cy, g) * inv);
            gp[i] = accumulate ? gp[i] + v : v;
        }
    } else {
        const float* t = targ + (size_t)bk * n;
        for (int i = threadIdx.x; i < n; i += 256) {
            const float v = scale * hm_dterm(kind, t[i], p[i]);
            gp[i] = accumulate ? gp[i] + v : v;
        }
    }
}

// ---- spatial softmax (T = 1) over each (frame, keypoint) map, strided input -> dense NCHW output ------------
// in element (b, k, i) lives at in[b*sb + i*si + k*sk]  (NHWC with padded channels: sb=n*C, si=C, sk=1).
__global__ __launch_bounds__(256) void softmax2d_fwd_kernel(const float* __restrict__ in, long sb, long si, long sk, int K, int n,
                                                            float* __restrict__ out) {
    __shared__ float red[4];
    const int bk = blockIdx.x, b = bk / K, k = bk - b * K;
    const float* src = in + (size_t)b * sb + (size_t)k * sk;
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < n; i += 256) mx = fmaxf(mx, src[(size_t)i * si]);
    mx = block_max<4>(mx, red);
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += expf(src[(size_t)i * si] - mx);
    s = block_sum<4>(s, red);
    float* dst = out + (size_t)bk * n;
    for (int i = threadIdx.x; i < n; i += 256) dst[i] = expf(src[(size_t)i * si] - mx) / s;
}

// (Measured and dropped, round 6: four workgroups per frame - each taking the (max, sum) pairs / the dots over the WHOLE maps and writing a
// quarter of the pixels - because 192 frames leave a quarter of the CUs without a workgroup.  Slower both ways, 167 -> 391 us forward and
// 333 -> 409 us backward: the first pass is the expensive one (two exponentials per logit), and it is the one the slices repeat;
// profiles/r06o_softmax_kernels.txt, profiles/retired/r06_softmax_slices.patch.)
// Same soft-max for pixel-major logits (sk == 1: the K logits of a pixel are contiguous, as the head's last layer writes them).
// One 1024-lane workgroup per frame: every lane owns pixels i = lane, lane + 1024, ... and reads a pixel's K logits as 16-B
// pieces (the per-map kernel above reads the same cache lines K times with a 4-B stride); pass 1 keeps K online (max, sum)
// pairs per lane and merges them through LDS, pass 2 re-reads the (L2-resident) logits and writes each map with coalesced rows.
constexpr int kSmK = 32;  // maps per frame handled by the pixel-major kernel
__global__ __launch_bounds__(1024) void softmax2d_pixmajor_kernel(const float* __restrict__ in, long sb, long si, int K, int n,
                                                                  float* __restrict__ out) {
    __shared__ float red_m[16][kSmK], red_s[16][kSmK], fin_m[kSmK], fin_is[kSmK];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* src = in + (size_t)b * sb;
    const int k4 = (K + 3) >> 2;
    float m[kSmK], s[kSmK];
#pragma unroll
    for (int k = 0; k < kSmK; ++k) {
        m[k] = -INFINITY;
        s[k] = 0.f;
    }
    for (int i = tid; i < n; i += 1024) {
        const f32x4* row = reinterpret_cast<const f32x4*>(src + (size_t)i * si);
        // the pixel's pieces are all requested before the first is used (round 6: behind one `if (q < k4)` each they were k4 dependent memory
        // latencies per pixel, with 16 waves per CU to hide them); a piece index past k4 reads piece k4 - 1 again and is dropped
        f32x4 piece[kSmK / 4];
#pragma unroll
        for (int q = 0; q < kSmK / 4; ++q) piece[q] = row[q < k4 ? q : k4 - 1];
#pragma unroll
        for (int q = 0; q < kSmK / 4; ++q) {
            if (q < k4) {
                const f32x4 v = piece[q];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = q * 4 + e;
                    const float x = v[e];
                    const float mm = fmaxf(m[k], x);
                    s[k] = s[k] * __expf(m[k] - mm) + __expf(x - mm);  // exp(-inf - finite) = 0 on the first pixel
                    m[k] = mm;
                }
            }
        }
    }
    // merge across the wave, then across the 16 waves
#pragma unroll
    for (int k = 0; k < kSmK; ++k) {
        if (k < K) {
            float mk = m[k], sk_ = s[k];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const float m2 = __shfl_xor(mk, d, 64), s2 = __shfl_xor(sk_, d, 64);
                const float mm = fmaxf(mk, m2);
                sk_ = (mk == -INFINITY ? 0.f : sk_ * __expf(mk - mm)) + (m2 == -INFINITY ? 0.f : s2 * __expf(m2 - mm));
                mk = mm;
            }
            if (lane == 0) {
                red_m[wave][k] = mk;
                red_s[wave][k] = sk_;
            }
        }
    }
    __syncthreads();
    if (tid < K) {
        float mk = red_m[0][tid], sk_ = red_s[0][tid];
        for (int w = 1; w < 16; ++w) {
            const float m2 = red_m[w][tid], s2 = red_s[w][tid];
            const float mm = fmaxf(mk, m2);
            sk_ = (mk == -INFINITY ? 0.f : sk_ * __expf(mk - mm)) + (m2 == -INFINITY ? 0.f : s2 * __expf(m2 - mm));
            mk = mm;
        }
        fin_m[tid] = mk;
        fin_is[tid] = 1.f / sk_;
    }
    __syncthreads();
    float* dst = out + (size_t)b * K * n;
    for (int i = tid; i < n; i += 1024) {
        const f32x4* row = reinterpret_cast<const f32x4*>(src + (size_t)i * si);
        f32x4 piece[kSmK / 4];
#pragma unroll
        for (int q = 0; q < kSmK / 4; ++q) piece[q] = row[q < k4 ? q : k4 - 1];
#pragma unroll
        for (int q = 0; q < kSmK / 4; ++q) {
            if (q < k4) {
                const f32x4 v = piece[q];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = q * 4 + e;
                    if (k < K) dst[(size_t)k * n + i] = __expf(v[e] - fin_m[k]) * fin_is[k];
                }
            }
        }
    }
}

// dlogit_i = p_i (g_i - sum_j g_j p_j); written back as bf16 (it feeds the MFMA kernels) in the strided layout of the logits
__global__ __launch_bounds__(256) void softmax2d_bwd_kernel(const float* __restrict__ prob, const float* __restrict__ gprob, int K,
                                                            int n, unsigned short* __restrict__ gin, long sb, long si, long sk) {
    __shared__ float red[4];
    const int bk = blockIdx.x, b = bk / K, k = bk - b * K;
    const float* p = prob + (size_t)bk * n;
    const float* g = gprob + (size_t)bk * n;
    float dot = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) dot = fmaf(p[i], g[i], dot);
    dot = block_sum<4>(dot, red);
    unsigned short* dst = gin + (size_t)b * sb + (size_t)k * sk;
    for (int i = threadIdx.x; i < n; i += 256) dst[(size_t)i * si] = f32_to_bf16(p[i] * (g[i] - dot));
}

// Same backward for pixel-major gradients (sk == 1, channel pitch si a multiple of 8): one 1024-lane workgroup per frame.  Pass 1
// keeps the K dots sum_i p g per lane (coalesced reads of every map), merged through LDS; pass 2 writes each pixel's WHOLE channel
// row - the K gradients and zeros in the pad channels - as 16-B pieces, so the caller needs no zero fill and no 2-byte strided stores.
__global__ __launch_bounds__(1024) void softmax2d_bwd_pixmajor_kernel(const float* __restrict__ prob, const float* __restrict__ gprob, int K,
                                                                      int n, unsigned short* __restrict__ gin, long sb, long si) {
    __shared__ float red[16][kSmK], fin[kSmK];
    // pass 2's staging rows: a wave's 64 pixels x 8 chunks of 16 B, pitch 144 B (round 6, below)
    __shared__ __attribute__((aligned(16))) unsigned short stage[16][64][72];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* p = prob + (size_t)b * K * n;
    const float* g = gprob + (size_t)b * K * n;
    float dot[kSmK];
#pragma unroll
    for (int k = 0; k < kSmK; ++k) dot[k] = 0.f;
    // The maps are read EIGHT at a time, all sixteen loads requested before the first product (round 6).  With one `if (k < K)` around each
    // pair of loads the compiler could not move a load above the branch in front of it, so a lane waited for memory K times per pixel with two
    // loads in flight - and this kernel has 16 waves per CU to hide that with: 250 of its 350 us were this loop.  A map index past K reads map
    // K - 1 again (a valid address) and its product is dropped.
    for (int i = tid; i < n; i += 1024) {
        // (one walking pointer per tensor: indexed as p[k n + i] the unrolled loop keeps 2 x 32 map base addresses alive - more than the
        //  scalar file holds - and the kernel, capped at 128 registers by its 1024 threads, spilled 38)
        const float* pk = p + i;
        const float* gk = g + i;
#pragma unroll
        for (int k0 = 0; k0 < kSmK; k0 += 8) {
            if (k0 < K) {   // (workgroup-uniform)
                float pv[8], gv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const size_t o = (size_t)(k0 + e < K ? e : K - 1 - k0) * n;
                    pv[e] = pk[o];
                    gv[e] = gk[o];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (k0 + e < K) dot[k0 + e] = fmaf(pv[e], gv[e], dot[k0 + e]);
                pk += (size_t)8 * n;
                gk += (size_t)8 * n;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kSmK; ++k) {
        if (k < K) {
            const float d = wave_sum(dot[k]);
            if (lane == 0) red[wave][k] = d;
        }
    }
    __syncthreads();
    if (tid < K) {
        float d = 0.f;
        for (int w = 0; w < 16; ++w) d += red[w][tid];
        fin[tid] = d;
    }
    __syncthreads();
    unsigned short* dst = gin + (size_t)b * sb;
    const int chunks = (int)(si >> 3);
    // A lane owns a pixel, and a pixel's row is `chunks` x 16 B: stored straight from the lane, every store instruction touched 64 different
    // 128-B lines with 16 B each - 178 of this kernel's 285 us were those stores (timing builds, profiles/r06v_softmax_probe.txt).  With the
    // head's 64-channel rows (chunks == 8) a wave now stages its 64 pixels x 128 B in LDS and writes them back as eight fully coalesced 1-KB
    // pieces (lane -> pixel lane / 8 of the piece, chunk lane % 8).
    const bool via_lds = chunks == 8;
    for (int i0 = wave * 64; i0 < n; i0 += 1024) {
        const int i = i0 + lane;
        const bool live = i < n;
        const float* pk = p + (live ? i : n - 1);
        const float* gk = g + (live ? i : n - 1);
        for (int c = 0; c < chunks; ++c) {
            float o[8];
            if (c * 8 < K) {   // (workgroup-uniform) the chunk's maps, all requested before the first is used; past K: map K - 1 again, dropped
                float pv[8], gv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const size_t off = (size_t)(c * 8 + e < K ? e : K - 1 - c * 8) * n;
                    pv[e] = pk[off];
                    gv[e] = gk[off];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = c * 8 + e < K ? pv[e] * (gv[e] - fin[c * 8 + e]) : 0.f;
                pk += (size_t)8 * n;
                gk += (size_t)8 * n;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = 0.f;
            }
            if (via_lds) *reinterpret_cast<u16x8*>(&stage[wave][lane][c * 8]) = pack_bf16x8(o);
            else if (live) *reinterpret_cast<u16x8*>(dst + (size_t)i * si + c * 8) = pack_bf16x8(o);
        }
        if (via_lds) {
            __builtin_amdgcn_wave_barrier();   // (lanes exchange through the wave's own rows: lock step on the device)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int px = j * 8 + (lane >> 3), ii = i0 + px;
                const u16x8 v = *reinterpret_cast<const u16x8*>(&stage[wave][px][(lane & 7) * 8]);
                if (ii < n) *reinterpret_cast<u16x8*>(dst + (size_t)ii * si + (lane & 7) * 8) = v;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

static GaussSpec make_spec(int img_h, int img_w, int h, int w, float sigma) {
    GaussSpec g;
    g.sx = (float)w / (float)img_w;
    g.sy = (float)h / (float)img_h;
    g.inv_two_var = 1.f / (2.f * sigma * sigma);
    g.h = h;
    g.w = w;
    return g;
}

// ---- evaluate_heatmaps_at_location (data/heatmaps.py:90-142): sum of the (2r+1)^2 window around int64(loc), zero padded.
// One lane per (frame, keypoint); taps are added in the reference's order (row offset outer, column offset inner).
__global__ __launch_bounds__(256) void heatmap_confidence_kernel(const float* __restrict__ heat, const float* __restrict__ locs, int n_maps,
                                                                 int h, int w, int radius, float* __restrict__ out) {
    const int bk = blockIdx.x * 256 + threadIdx.x;
    if (bk >= n_maps) return;
    const float lx = locs[bk * 2], ly = locs[bk * 2 + 1];
    float acc = 0.f;
    if (lx == lx && ly == ly && fabsf(lx) < 1e9f && fabsf(ly) < 1e9f) {
        const int cx = (int)lx, cy = (int)ly;  // truncation toward zero, as Tensor.type(torch.int64)
        const float* m = heat + (size_t)bk * h * w;
        for (int dy = -radius; dy <= radius; ++dy) {
            const int y = cy + dy;
            for (int dx = -radius; dx <= radius; ++dx) {
                const int x = cx + dx;
                if (y >= 0 && y < h && x >= 0 && x < w) acc += m[y * w + x];
            }
        }
    }
    out[bk] = acc;
}

// ---- TemporalHeatmapLoss (losses/losses.py:706-869): per (t, k) a distance between the heat-maps of frames t and t+1 -----------
//   temporal_heatmap_mse: mean over pixels of (p_t - p_t+1)^2                                             (:815-819)
//   temporal_heatmap_kl : kornia kl_div_loss_2d(pred = p_t + 1e-10, target = p_t+1 + 1e-10, 'none')
//                         = sum target (log target - log pred)                                            (:820-826)
// then: 0 where conf_t or conf_t+1 < prob_threshold (:783-791), relu(d - epsilon_k) (:763), mean over ALL (S-1) K entries (:865).
// pass 1: one workgroup per (t, k), t < S-1 -> d[t][k]
__global__ __launch_bounds__(256) void temporal_hm_dist_kernel(const float* __restrict__ pred, int K, int n, int kind, float* __restrict__ d) {
    __shared__ float red[4];
    const int tk = blockIdx.x;  // t * K + k
    const float* a = pred + (size_t)tk * n;        // frame t
    const float* b = a + (size_t)K * n;            // frame t + 1
    float part = 0.f;
    if (kind == LP_HM_MSE) {
        for (int i = threadIdx.x; i < n; i += 256) {
            const float df = a[i] - b[i];
            part = fmaf(df, df, part);
        }
    } else {
        for (int i = threadIdx.x; i < n; i += 256) {
            const float pe = a[i] + 1e-10f, te = b[i] + 1e-10f;
            part += te * (logf(te) - logf(pe));
        }
    }
    const float tot = block_sum<4>(part, red);
    if (threadIdx.x == 0) d[tk] = kind == LP_HM_MSE ? tot / (float)n : tot;
}

// pass 2 (one workgroup): mask, epsilon, mean; act[t][k] = 1 where the rectified term is active (its gradient is non-zero)
__global__ __launch_bounds__(256) void temporal_hm_finish_kernel(const float* __restrict__ d, const float* __restrict__ conf,
                                                                 const float* __restrict__ eps, float thr, int S, int K,
                                                                 float* __restrict__ act, float* __restrict__ loss) {
    __shared__ float red[4];
    const int rows = (S - 1) * K;
    float s = 0.f;
    for (int i = threadIdx.x; i < rows; i += 256) {
        const int k = i % K;
        const bool ignore = (conf[i] < thr) || (conf[i + K] < thr);
        const float v = (ignore ? 0.f : d[i]) - eps[k];
        const bool on = v > 0.f;
        s += on ? v : 0.f;
        act[i] = (on && !ignore) ? 1.f : 0.f;
    }
    s = block_sum<4>(s, red);
    if (threadIdx.x == 0) loss[0] = s / (float)rows;  // S == 1: 0 / 0 = NaN, the mean of an empty tensor
}

// backward: the map of frame t takes part in the pairs (t-1, t) and (t, t+1)
__global__ __launch_bounds__(256) void temporal_hm_grad_kernel(const float* __restrict__ pred, const float* __restrict__ act, int S, int K, int n,
                                                               int kind, const float* __restrict__ gout, float* __restrict__ gpred,
                                                               int accumulate) {
    const int tk = blockIdx.x, t = tk / K;
    const float scale = gout[0] / (float)((S - 1) * K);
    const float w_next = (t < S - 1) ? act[tk] * scale : 0.f;      // pair (t, t+1): this map is the first argument
    const float w_prev = (t > 0) ? act[tk - K] * scale : 0.f;      // pair (t-1, t): this map is the second argument
    const float* p = pred + (size_t)tk * n;
    float* gp = gpred + (size_t)tk * n;
    const float inv_n = 1.f / (float)n;
    for (int i = threadIdx.x; i < n; i += 256) {
        float gv = 0.f;
        if (w_next != 0.f) {
            const float q = p[i + (size_t)K * n];
            gv += w_next * (kind == LP_HM_MSE ? 2.f * (p[i] - q) * inv_n : -(q + 1e-10f) / (p[i] + 1e-10f));
        }
        if (w_prev != 0.f) {
            const float q = p[(ptrdiff_t)i - (ptrdiff_t)K * n];
            gv += w_prev * (kind == LP_HM_MSE ? 2.f * (p[i] - q) * inv_n : logf(p[i] + 1e-10f) - logf(q + 1e-10f) + 1.f);
        }
        gp[i] = accumulate ? gp[i] + gv : gv;
    }
}

}  // namespace lp

extern "C" int lp_heatmap_gen(const float* keypoints, const int* visibility, int B, int K, int img_h, int img_w, int h, int w,
                              float sigma, float* out, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(keypoints && out && B >= 0 && K > 0 && h > 0 && w > 0 && img_h > 0 && img_w > 0 && sigma > 0.f);
    if (B == 0) return LP_OK;
    hipLaunchKernelGGL(heatmap_gen_kernel, dim3(B * K), dim3(256), 0, (hipStream_t)stream, keypoints, visibility,
                       make_spec(img_h, img_w, h, w, sigma), out);
    return launch_status();
}

extern "C" int lp_heatmap_gen_bwd(const float* keypoints, const int* visibility, int B, int K, int img_h, int img_w, int h, int w,
                                  float sigma, const float* grad_out, float* grad_keypoints, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(keypoints && grad_out && grad_keypoints && B >= 0 && K > 0 && h > 0 && w > 0 && img_h > 0 && img_w > 0 && sigma > 0.f);
    if (h > kGaussAxisMax || w > kGaussAxisMax) return LP_ERR_UNSUPPORTED;
    if (B == 0) return LP_OK;
    hipLaunchKernelGGL(heatmap_gen_bwd_kernel, dim3(B * K), dim3(256), 0, (hipStream_t)stream, keypoints, visibility,
                       make_spec(img_h, img_w, h, w, sigma), grad_out, grad_keypoints);
    return launch_status();
}

extern "C" int lp_heatmap_confidence(const float* heat, const float* locs, int B, int K, int h, int w, int radius, float* out,
                                     lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(heat && locs && out && B >= 0 && K > 0 && h > 0 && w > 0 && radius >= 0);
    if (B == 0) return LP_OK;
    hipLaunchKernelGGL(heatmap_confidence_kernel, dim3((B * K + 255) / 256), dim3(256), 0, (hipStream_t)stream, heat, locs, B * K, h, w,
                       radius, out);
    return launch_status();
}

// workspace layout: float d[(S-1)*K] | float act[(S-1)*K]
extern "C" size_t lp_temporal_heatmap_workspace_bytes(int S, int K) { return (size_t)(S > 1 ? S - 1 : 1) * K * 8 + 16; }

extern "C" int lp_temporal_heatmap_fwd(int kind, const float* pred, const float* conf, int S, int K, int h, int w, const float* epsilon,
                                       float prob_threshold, float* loss, void* workspace, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(pred && conf && epsilon && loss && workspace && S > 0 && K > 0 && h > 0 && w > 0);
    LP_REQUIRE(kind == LP_HM_MSE || kind == LP_HM_KL);
    float* d = (float*)workspace;
    float* act = d + (size_t)(S > 1 ? S - 1 : 1) * K;
    hipStream_t st = (hipStream_t)stream;
    if (S > 1) hipLaunchKernelGGL(temporal_hm_dist_kernel, dim3((S - 1) * K), dim3(256), 0, st, pred, K, h * w, kind, d);
    hipLaunchKernelGGL(temporal_hm_finish_kernel, dim3(1), dim3(256), 0, st, (const float*)d, conf, epsilon, prob_threshold, S, K, act, loss);
    return launch_status();
}

extern "C" int lp_temporal_heatmap_bwd(int kind, const float* pred, int S, int K, int h, int w, const void* workspace, const float* gout,
                                       float* gpred, int accumulate, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(pred && workspace && gout && gpred && S > 0 && K > 0 && h > 0 && w > 0);
    LP_REQUIRE(kind == LP_HM_MSE || kind == LP_HM_KL);
    const float* act = (const float*)workspace + (size_t)(S > 1 ? S - 1 : 1) * K;
    hipLaunchKernelGGL(temporal_hm_grad_kernel, dim3(S * K), dim3(256), 0, (hipStream_t)stream, pred, act, S, K, h * w, kind, gout, gpred,
                       accumulate);
    return launch_status();
}

extern "C" size_t lp_heatmap_mse_workspace_bytes(int B, int K) { return (size_t)B * K * 8 + 16; }

// workspace layout: float rowsum[B*K] | int valid[B*K] | float nvalid | pad
extern "C" int lp_heatmap_loss_fwd(int kind, const float* targ, const float* pred, int B, int K, int h, int w, float* loss,
                                   void* workspace, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(targ && pred && loss && workspace && B > 0 && K > 0 && h > 0 && w > 0);
    LP_REQUIRE(kind == LP_HM_MSE || kind == LP_HM_KL || kind == LP_HM_JS);
    float* rowsum = (float*)workspace;
    int* valid = (int*)(rowsum + (size_t)B * K);
    float* nvalid = (float*)(valid + (size_t)B * K);
    GaussSpec g = make_spec(1, 1, h, w, 1.f);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL((hm_rowsq_kernel<false>), dim3(B * K), dim3(256), 0, st, targ, pred, (const float*)nullptr,
                       (const float*)nullptr, 0.f, g, kind, rowsum, valid);
    hipLaunchKernelGGL(hm_finish_kernel, dim3(1), dim3(256), 0, st, (const float*)rowsum, (const int*)valid, B * K, 0, loss, nvalid);
    return launch_status();
}

extern "C" int lp_heatmap_mse_fwd(const float* targ, const float* pred, int B, int K, int h, int w, float* loss, void* workspace,
                                  lp_stream_t stream) {
    return lp_heatmap_loss_fwd(LP_HM_MSE, targ, pred, B, K, h, w, loss, workspace, stream);
}

extern "C" int lp_heatmap_loss_bwd(int kind, const float* targ, const float* pred, int B, int K, int h, int w, const void* workspace,
                                   const float* gout, float* gpred, int accumulate, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(targ && pred && workspace && gout && gpred && B > 0 && K > 0 && h > 0 && w > 0);
    LP_REQUIRE(kind == LP_HM_MSE || kind == LP_HM_KL || kind == LP_HM_JS);
    const float* rowsum = (const float*)workspace;
    const int* valid = (const int*)(rowsum + (size_t)B * K);
    const float* nvalid = (const float*)(valid + (size_t)B * K);
    GaussSpec g = make_spec(1, 1, h, w, 1.f);
    hipLaunchKernelGGL((hm_grad_kernel<false>), dim3(B * K), dim3(256), 0, (hipStream_t)stream, targ, pred, (const float*)nullptr,
                       g, kind, valid, nvalid, gout, gpred, accumulate);
    return launch_status();
}

extern "C" int lp_heatmap_mse_bwd(const float* targ, const float* pred, int B, int K, int h, int w, const void* workspace,
                                  const float* gout, float* gpred, int accumulate, lp_stream_t stream) {
    return lp_heatmap_loss_bwd(LP_HM_MSE, targ, pred, B, K, h, w, workspace, gout, gpred, accumulate, stream);
}

extern "C" int lp_unimodal_mse_fwd(const float* kp_aug, const float* pred, const float* conf, int S, int K, int img_h, int img_w,
                                   int h, int w, float sigma, float prob_threshold, float* loss, void* workspace,
                                   lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(kp_aug && pred && conf && loss && workspace && S > 0 && K > 0 && h > 0 && w > 0 && sigma > 0.f);
    float* rowsum = (float*)workspace;
    int* valid = (int*)(rowsum + (size_t)S * K);
    float* nvalid = (float*)(valid + (size_t)S * K);
    GaussSpec g = make_spec(img_h, img_w, h, w, sigma);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL((hm_rowsq_kernel<true>), dim3(S * K), dim3(256), 0, st, (const float*)nullptr, pred, kp_aug, conf,
                       prob_threshold, g, (int)LP_HM_MSE, rowsum, valid);
    hipLaunchKernelGGL(hm_finish_kernel, dim3(1), dim3(256), 0, st, (const float*)rowsum, (const int*)valid, S * K, 1, loss, nvalid);
    return launch_status();
}

extern "C" int lp_unimodal_mse_bwd(const float* kp_aug, const float* pred, int S, int K, int img_h, int img_w, int h, int w,
                                   float sigma, const void* workspace, const float* gout, float* gpred, int accumulate,
                                   lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(kp_aug && pred && workspace && gout && gpred && S > 0 && K > 0 && h > 0 && w > 0 && sigma > 0.f);
    const float* rowsum = (const float*)workspace;
    const int* valid = (const int*)(rowsum + (size_t)S * K);
    const float* nvalid = (const float*)(valid + (size_t)S * K);
    GaussSpec g = make_spec(img_h, img_w, h, w, sigma);
    hipLaunchKernelGGL((hm_grad_kernel<true>), dim3(S * K), dim3(256), 0, (hipStream_t)stream, (const float*)nullptr, pred, kp_aug, g,
                       (int)LP_HM_MSE, valid, nvalid, gout, gpred, accumulate);
    return launch_status();
}

extern "C" int lp_softmax2d_fwd(const float* in, long stride_b, long stride_i, long stride_k, int B, int K, int n, float* out,
                                lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(in && out && B >= 0 && K > 0 && n > 0);
    if (B == 0) return LP_OK;
    // pixel-major logits (the head's layout): one workgroup per frame reading whole pixels; needs 16-B readable rows
    if (stride_k == 1 && K <= kSmK && stride_i % 4 == 0 && stride_i >= ((K + 3) & ~3) && stride_b % 4 == 0 && ((uintptr_t)in & 15) == 0)
        hipLaunchKernelGGL(softmax2d_pixmajor_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, in, stride_b, stride_i, K, n, out);
    else
        hipLaunchKernelGGL(softmax2d_fwd_kernel, dim3(B * K), dim3(256), 0, (hipStream_t)stream, in, stride_b, stride_i, stride_k, K, n, out);
    return launch_status();
}

extern "C" int lp_softmax2d_bwd(const float* prob, const float* gprob, int B, int K, int n, void* gin_bf16, long stride_b, long stride_i,
                                long stride_k, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(prob && gprob && gin_bf16 && B >= 0 && K > 0 && n > 0);
    if (B == 0) return LP_OK;
    // pixel-major gradient rows (the head's layout): whole channel rows per pixel, pad channels [K, stride_i) written as zeros
    if (stride_k == 1 && K <= kSmK && stride_i % 8 == 0 && stride_i >= K && stride_b % 8 == 0 && ((uintptr_t)gin_bf16 & 15) == 0)
        hipLaunchKernelGGL(softmax2d_bwd_pixmajor_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, prob, gprob, K, n,
                           (unsigned short*)gin_bf16, stride_b, stride_i);
    else
        hipLaunchKernelGGL(softmax2d_bwd_kernel, dim3(B * K), dim3(256), 0, (hipStream_t)stream, prob, gprob, K, n, (unsigned short*)gin_bf16,
                           stride_b, stride_i, stride_k);
    return launch_status();
}
