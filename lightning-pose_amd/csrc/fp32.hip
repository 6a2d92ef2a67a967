// fp32 VALIDATION path of the heatmap tracker's network: the same layers as conv.hip / bn.hip, computed in fp32 end to end on the
// fp32 matrix-core instruction (v_mfma_f32_32x32x2_f32), so that a whole training step can be compared with the reference - which
// trains in fp32 only (lightning_pose/train.py:411-428 passes no `precision=`) - at the 1e-4 tolerance BASELINE.json's north_star
// states for fp32.  The bf16-mixed kernels are the product path and the measured one; nothing here is tuned: operands are fetched
// with scalar loads straight from global memory (no LDS staging), one wave per 32 x 32 output tile.  What matters is that every
// value goes through the same data layout (NHWC activations, [Co][R][S][Ci] weights in the flat parameter buffer), the same
// BatchNorm / pooling / head arithmetic and the same host wiring as the product, in the reference's precision.
//
// Replaces, like conv.hip / bn.hip: the cuDNN / ATen calls behind `self.backbone(images)` (models/base.py:398, torchvision
// Bottleneck semantics in SURVEY.md Appendix A) and `HeatmapHead.forward` (models/heads/heatmap.py:203-212).
#include "lp_common.h"

namespace lp {

struct F32Geom {
    int B, Hi, Wi, Ci;  // input  tensor (NHWC, dense)
    int Ho, Wo, Co;     // output tensor (NHWC, dense)
    int R, S, stride, pad;
    int KH, KW, CiS;    // storage dims of the weight [Co][KH][KW][CiS] (KH >= R, KW >= S, CiS >= Ci: the stem is stored [64][8][8][4])
};

// out[m][n] = sum_k A(m, k) W(n, k) (+ bias[n]) (+ addend[m][n])
//   MODE 0 (forward)        m = (b, ho, wo)  n = co  k = (r, s, ci)   A = x[b][ho*st - pad + r][wo*st - pad + s][ci]
//   MODE 1 (data gradient)  m = (b, hi, wi)  n = ci  k = (r, s, co)   A = dy[b][(hi + pad - r)/st][(wi + pad - s)/st][co] where divisible
// MFMA 32x32x2 f32: lane l supplies A[row l % 32][k = l / 32] and B[k = l / 32][column l % 32].
template <int MODE>
__global__ __launch_bounds__(256) void f32_conv_kernel(const float* __restrict__ X, const float* __restrict__ W, F32Geom g, int M, int N,
                                                       int tiles_n, int ntiles, const float* __restrict__ bias,
                                                       const float* addend, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= ntiles) return;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int r32 = lane & 31, kk = lane >> 5;
    const int m = tm * 32 + r32, n = tn * 32 + r32;
    const bool mv = m < M, nv = n < N;
    const int oh = MODE == 0 ? g.Ho : g.Hi, ow = MODE == 0 ? g.Wo : g.Wi;  // spatial size of the OUTPUT of this launch
    const int mm = mv ? m : 0;
    const int b = mm / (oh * ow), rem = mm - b * oh * ow;
    const int y = rem / ow, x = rem - y * ow;
    const int ck = MODE == 0 ? g.Ci : g.Co;             // contraction channels per tap
    const size_t wtap = (size_t)g.CiS;                  // weight stride of one (r, s) step
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int r = 0; r < g.R; ++r)
        for (int s = 0; s < g.S; ++s) {
            const float* pa = nullptr;
            if (mv) {
                if (MODE == 0) {
                    const int sy = y * g.stride - g.pad + r, sx = x * g.stride - g.pad + s;
                    if (sy >= 0 && sy < g.Hi && sx >= 0 && sx < g.Wi) pa = X + ((size_t)(b * g.Hi + sy) * g.Wi + sx) * g.Ci;
                } else {
                    const int th = y + g.pad - r, tw = x + g.pad - s;
                    if (th >= 0 && tw >= 0 && th % g.stride == 0 && tw % g.stride == 0) {
                        const int sy = th / g.stride, sx = tw / g.stride;
                        if (sy < g.Ho && sx < g.Wo) pa = X + ((size_t)(b * g.Ho + sy) * g.Wo + sx) * g.Co;
                    }
                }
            }
            // forward: W[n][r][s][c] (c contiguous); data gradient: W[c][r][s][n] (stride KH*KW*CiS over c)
            const float* pw = nullptr;
            size_t wstep = 1;
            if (nv) {
                if (MODE == 0) {
                    pw = W + (((size_t)n * g.KH + r) * g.KW + s) * wtap;
                } else {
                    pw = W + ((size_t)r * g.KW + s) * wtap + n;
                    wstep = (size_t)g.KH * g.KW * wtap;
                }
            }
            for (int c0 = 0; c0 < ck; c0 += 8) {
                float av[4], bv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int c = c0 + 2 * u + kk;
                    av[u] = (pa != nullptr && c < ck) ? pa[c] : 0.f;
                    bv[u] = (pw != nullptr && c < ck) ? pw[(size_t)c * wstep] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
            }
        }
    const int col = tn * 32 + (lane & 31);
    if (col >= N) return;
    const float bs = bias ? bias[col] : 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = tm * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (row < M) {
            const size_t o = (size_t)row * N + col;
            float v = acc[e] + bs;
            if (addend) v += addend[o];
            out[o] = v;
        }
    }
}

// dW[n][r][s][ci] += sum_m x_gather[m][(r, s, ci)] * dy[m][n]: D[j][n], rows j = (r, s, ci) from the gathered activations, columns n from
// dy, contraction over the pixels m (split over blockIdx.y; partial tiles are added with fp32 atomics)
__global__ __launch_bounds__(256) void f32_wgrad_kernel(const float* __restrict__ X, const float* __restrict__ DY, F32Geom g, int M, int Kw,
                                                        int tiles_n, int ntiles, int m_per_split, float* __restrict__ dW) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= ntiles) return;
    const int tj = tile / tiles_n, tn = tile - tj * tiles_n;
    const int r32 = lane & 31, kk = lane >> 5;
    const int j = tj * 32 + r32, n = tn * 32 + r32;
    const bool jv = j < Kw, nv = n < g.Co;
    const int jj = jv ? j : 0;
    const int tap = jj / g.Ci, ci = jj - tap * g.Ci;
    const int tr = tap / g.S, ts = tap - tr * g.S;
    const int hw = g.Ho * g.Wo;
    const int m_begin = blockIdx.y * m_per_split, m_end = min(M, m_begin + m_per_split);
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int m0 = m_begin; m0 < m_end; m0 += 8) {
        float av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int m = m0 + 2 * u + kk;
            av[u] = bv[u] = 0.f;
            if (m < m_end) {
                if (jv) {
                    const int b = m / hw, rem = m - b * hw;
                    const int ho = rem / g.Wo, wo = rem - ho * g.Wo;
                    const int hi = ho * g.stride - g.pad + tr, wi = wo * g.stride - g.pad + ts;
                    if (hi >= 0 && hi < g.Hi && wi >= 0 && wi < g.Wi) av[u] = X[((size_t)(b * g.Hi + hi) * g.Wi + wi) * g.Ci + ci];
                }
                if (nv) bv[u] = DY[(size_t)m * g.Co + n];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
    }
    const int col = tn * 32 + (lane & 31);
    if (col >= g.Co) return;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = tj * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (row < Kw) {
            const int tp = row / g.Ci, c = row - tp * g.Ci;
            const int r = tp / g.S, s = tp - r * g.S;
            atomicAdd(&dW[(((size_t)col * g.KH + r) * g.KW + s) * g.CiS + c], acc[e]);
        }
    }
}

// ---- BatchNorm (training mode), ReLU, residual: one thread per element, channel fastest ------------------------------------------
// sums[0][c] += sum x, sums[1][c] += sum x^2 over the rows (thread = channel x row stripe; fp32 atomics)
__global__ __launch_bounds__(256) void f32_bn_stats_kernel(const float* __restrict__ X, int M, int C, float* __restrict__ sums) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    if (c >= C) return;
    float s0 = 0.f, s1 = 0.f;
    for (int r = blockIdx.y * 4 + rl; r < M; r += gridDim.y * 4) {
        const float v = X[(size_t)r * C + c];
        s0 += v;
        s1 = fmaf(v, v, s1);
    }
    atomicAdd(&sums[c], s0);
    atomicAdd(&sums[C + c], s1);
}

// the same sums without atomics: every (row stripe, row lane) writes its partial pair to part[stripe * 4 + lane][2][C] and
// f32_bn_stats_fold_kernel adds them up in index order - the validation executor's forward is then reproducible bit for bit from run to run
// (its keypoints are compared with the reference at 3e-3 px, and soft-argmax multiplies by T = 1000: the summation order showed)
__global__ __launch_bounds__(256) void f32_bn_stats_part_kernel(const float* __restrict__ X, int M, int C, float* __restrict__ part) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    if (c >= C) return;
    float s0 = 0.f, s1 = 0.f;
    for (int r = blockIdx.y * 4 + rl; r < M; r += gridDim.y * 4) {
        const float v = X[(size_t)r * C + c];
        s0 += v;
        s1 = fmaf(v, v, s1);
    }
    float* dst = part + (size_t)(blockIdx.y * 4 + rl) * 2 * C;
    dst[c] = s0;
    dst[C + c] = s1;
}

__global__ __launch_bounds__(256) void f32_bn_stats_fold_kernel(const float* __restrict__ part, int nparts, int C, float* __restrict__ sums) {
    const int i = blockIdx.x * 256 + threadIdx.x;   // over 2 C
    if (i >= 2 * C) return;
    float t = 0.f;
    for (int p = 0; p < nparts; ++p) t += part[(size_t)p * 2 * C + i];
    sums[i] += t;
}

__global__ __launch_bounds__(256) void f32_bn_apply_kernel(const float* __restrict__ X, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ residual, int relu,
                                                           size_t total, int C, float* __restrict__ Y) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        float o = (X[i] - mean[c]) * invstd[c] * gamma[c] + beta[c];
        if (residual) o += residual[i];
        if (relu) o = fmaxf(o, 0.f);
        Y[i] = o;
    }
}

// sums[0][c] += sum dz, sums[1][c] += sum dz * xhat, dz = dy where y_out > 0 (y_out may be NULL: no ReLU); d beta / d gamma likewise
__global__ __launch_bounds__(256) void f32_bn_bwd_reduce_kernel(const float* __restrict__ DY, const float* __restrict__ Yout,
                                                                const float* __restrict__ X, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, int M, int C, float* __restrict__ sums,
                                                                float* __restrict__ dbeta, float* __restrict__ dgamma) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    if (c >= C) return;
    const float mu = mean[c], is = invstd[c];
    float s0 = 0.f, s1 = 0.f;
    for (int r = blockIdx.y * 4 + rl; r < M; r += gridDim.y * 4) {
        const size_t i = (size_t)r * C + c;
        float d = DY[i];
        if (Yout && !(Yout[i] > 0.f)) d = 0.f;
        s0 += d;
        s1 = fmaf(d, (X[i] - mu) * is, s1);
    }
    atomicAdd(&sums[c], s0);
    atomicAdd(&sums[C + c], s1);
    if (dbeta) atomicAdd(&dbeta[c], s0);
    if (dgamma) atomicAdd(&dgamma[c], s1);
}

__global__ __launch_bounds__(256) void f32_bn_bwd_apply_kernel(const float* __restrict__ DY, const float* __restrict__ Yout,
                                                               const float* __restrict__ X, const float* __restrict__ mean,
                                                               const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                               const float* __restrict__ sums, float inv_count, size_t total, int C,
                                                               float* __restrict__ DX, float* __restrict__ DRES) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        float d = DY[i];
        if (Yout && !(Yout[i] > 0.f)) d = 0.f;
        const float xh = (X[i] - mean[c]) * invstd[c];
        DX[i] = gamma[c] * invstd[c] * (d - sums[c] * inv_count - xh * sums[C + c] * inv_count);
        if (DRES) DRES[i] = d;
    }
}

// 3x3 / stride 2 / pad 1 max-pool, NHWC fp32; arg-max tap recorded (first maximum in row-major scan, ATen's tie rule)
__global__ __launch_bounds__(256) void f32_maxpool_fwd_kernel(const float* __restrict__ X, int B, int Hi, int Wi, int C, int Ho, int Wo,
                                                              float* __restrict__ Y, unsigned char* __restrict__ IDX) {
    const size_t total = (size_t)B * Ho * Wo * C;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (size_t)gridDim.x * 256) {
        const int c = (int)(q % C);
        size_t p = q / C;
        const int wo = (int)(p % Wo);
        p /= Wo;
        const int ho = (int)(p % Ho), b = (int)(p / Ho);
        float m = -INFINITY;
        unsigned char arg = 0;
        for (int kh = 0; kh < 3; ++kh) {
            const int hi = ho * 2 - 1 + kh;
            if (hi < 0 || hi >= Hi) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int wi = wo * 2 - 1 + kw;
                if (wi < 0 || wi >= Wi) continue;
                const float v = X[(((size_t)b * Hi + hi) * Wi + wi) * C + c];
                if (v > m) {
                    m = v;
                    arg = (unsigned char)(kh * 3 + kw);
                }
            }
        }
        Y[q] = m;
        IDX[q] = arg;
    }
}

__global__ __launch_bounds__(256) void f32_maxpool_bwd_kernel(const unsigned char* __restrict__ IDX, const float* __restrict__ DY, int B,
                                                              int Hi, int Wi, int C, int Ho, int Wo, float* __restrict__ DX) {
    const size_t total = (size_t)B * Hi * Wi * C;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (size_t)gridDim.x * 256) {
        const int c = (int)(q % C);
        size_t p = q / C;
        const int wi = (int)(p % Wi);
        p /= Wi;
        const int hi = (int)(p % Hi), b = (int)(p / Hi);
        float gsum = 0.f;
        const int ho_lo = hi / 2, ho_hi = min(Ho - 1, (hi + 1) / 2);
        const int wo_lo = wi / 2, wo_hi = min(Wo - 1, (wi + 1) / 2);
        for (int ho = ho_lo; ho <= ho_hi; ++ho)
            for (int wo = wo_lo; wo <= wo_hi; ++wo) {
                const unsigned mine = (unsigned)((hi - (ho * 2 - 1)) * 3 + (wi - (wo * 2 - 1)));
                const size_t o = (((size_t)b * Ho + ho) * Wo + wo) * C + c;
                if (IDX[o] == mine) gsum += DY[o];
            }
        DX[q] = gsum;
    }
}

// images (B,3,H,W) NCHW -> (B,H,W,4), channel 3 = 0
__global__ __launch_bounds__(256) void f32_nchw3_to_nhwc4_kernel(const float* __restrict__ X, int B, int HW, float* __restrict__ Y) {
    const size_t total = (size_t)B * HW;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (size_t)gridDim.x * 256) {
        const size_t b = q / HW, p = q % HW;
        const float* src = X + b * 3 * HW + p;
        Y[q * 4 + 0] = src[0];
        Y[q * 4 + 1] = src[HW];
        Y[q * 4 + 2] = src[2 * (size_t)HW];
        Y[q * 4 + 3] = 0.f;
    }
}

// PixelShuffle(2) on NHWC: out[b][2y+i][2x+j][c] = in[b][y][x][4c + 2i + j] (high-resolution channel pitch ld >= Cout); inverse for the gradient
__global__ __launch_bounds__(256) void f32_pixel_shuffle_kernel(const float* __restrict__ IN, int B, int h, int w, int Cout, int ld, int inverse,
                                                                float* __restrict__ OUT) {
    const int cin = Cout * 4;
    const size_t total = (size_t)B * h * w * cin;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (size_t)gridDim.x * 256) {
        const int cl = (int)(q % cin);
        size_t p = q / cin;
        const int x = (int)(p % w);
        p /= w;
        const int y = (int)(p % h), b = (int)(p / h);
        const int c = cl >> 2, i = (cl >> 1) & 1, j = cl & 1;
        const size_t o = ((((size_t)b * 2 * h + 2 * y + i) * 2 * w) + 2 * x + j) * ld + c;
        if (inverse) OUT[q] = IN[o];
        else OUT[o] = IN[q];
    }
}

// soft-max backward with fp32 output in the logits' strided layout: dlogit_i = p_i (g_i - sum_j g_j p_j)
__global__ __launch_bounds__(256) void f32_softmax2d_bwd_kernel(const float* __restrict__ prob, const float* __restrict__ gprob, int K, int n,
                                                                float* __restrict__ gin, long sb, long si, long sk) {
    __shared__ float red[4];
    const int bk = blockIdx.x, b = bk / K, k = bk - b * K;
    const float* p = prob + (size_t)bk * n;
    const float* g = gprob + (size_t)bk * n;
    float dot = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) dot = fmaf(p[i], g[i], dot);
    dot = block_sum<4>(dot, red);
    float* dst = gin + (size_t)b * sb + (size_t)k * sk;
    for (int i = threadIdx.x; i < n; i += 256) dst[(size_t)i * si] = p[i] * (g[i] - dot);
}

static int f32_grid(size_t n) {
    size_t blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    return blocks < 1 ? 1 : (int)blocks;
}

static bool f32_geom_ok(const lp_conv_geom* c, int KH, int KW, int CiS) {
    return c && c->B > 0 && c->Hi > 0 && c->Wi > 0 && c->Ci > 0 && c->Ho > 0 && c->Wo > 0 && c->Co > 0 && c->R > 0 && c->S > 0 &&
           c->stride > 0 && c->pad >= 0 && KH >= c->R && KW >= c->S && CiS >= c->Ci;
}

static F32Geom to_f32_geom(const lp_conv_geom* c, int KH, int KW, int CiS) {
    F32Geom g{c->B, c->Hi, c->Wi, c->Ci, c->Ho, c->Wo, c->Co, c->R, c->S, c->stride, c->pad, KH, KW, CiS};
    return g;
}

}  // namespace lp

extern "C" int lp_f32_conv_fwd(const float* x, const float* w, const lp_conv_geom* geom, int KH, int KW, int CiS, const float* bias,
                               const float* addend, float* out, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(x && w && out && f32_geom_ok(geom, KH, KW, CiS));
    const F32Geom g = to_f32_geom(geom, KH, KW, CiS);
    const long long M = (long long)g.B * g.Ho * g.Wo;
    if (M >= (1LL << 31)) return LP_ERR_UNSUPPORTED;
    const int N = g.Co, tiles_n = (N + 31) / 32, ntiles = (int)((M + 31) / 32) * tiles_n;
    hipLaunchKernelGGL((f32_conv_kernel<0>), dim3((ntiles + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, w, g, (int)M, N, tiles_n, ntiles, bias,
                       addend, out);
    return launch_status();
}

extern "C" int lp_f32_conv_dgrad(const float* dy, const float* w, const lp_conv_geom* geom, int KH, int KW, int CiS, const float* bias,
                                 const float* addend, float* dx, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(dy && w && dx && f32_geom_ok(geom, KH, KW, CiS));
    const F32Geom g = to_f32_geom(geom, KH, KW, CiS);
    const long long M = (long long)g.B * g.Hi * g.Wi;
    if (M >= (1LL << 31)) return LP_ERR_UNSUPPORTED;
    const int N = g.Ci, tiles_n = (N + 31) / 32, ntiles = (int)((M + 31) / 32) * tiles_n;
    hipLaunchKernelGGL((f32_conv_kernel<1>), dim3((ntiles + 3) / 4), dim3(256), 0, (hipStream_t)stream, dy, w, g, (int)M, N, tiles_n, ntiles, bias,
                       addend, dx);
    return launch_status();
}

extern "C" int lp_f32_conv_wgrad(const float* x, const float* dy, const lp_conv_geom* geom, int KH, int KW, int CiS, float* dw,
                                 lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(x && dy && dw && f32_geom_ok(geom, KH, KW, CiS));
    const F32Geom g = to_f32_geom(geom, KH, KW, CiS);
    const long long M = (long long)g.B * g.Ho * g.Wo;
    if (M >= (1LL << 31)) return LP_ERR_UNSUPPORTED;
    const int Kw = g.R * g.S * g.Ci, tiles_n = (g.Co + 31) / 32, ntiles = ((Kw + 31) / 32) * tiles_n;
    // enough pixel slices to fill the chip (~2048 waves), each at least 256 pixels deep
    int split = (2048 * 4 + ntiles - 1) / ntiles;
    const int max_split = (int)((M + 255) / 256);
    if (split > max_split) split = max_split;
    if (split < 1) split = 1;
    if (split > 65535) split = 65535;
    const int per = (int)(((M + split - 1) / split + 7) / 8 * 8);
    split = (int)((M + per - 1) / per);
    hipLaunchKernelGGL(f32_wgrad_kernel, dim3((ntiles + 3) / 4, split), dim3(256), 0, (hipStream_t)stream, x, dy, g, (int)M, Kw, tiles_n, ntiles,
                       per, dw);
    return launch_status();
}

extern "C" int lp_f32_bn_stats(const float* x, int M, int C, float* sums, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(x && sums && M > 0 && C > 0);
    int gy = M / 64;
    gy = gy < 1 ? 1 : (gy > 256 ? 256 : gy);
    hipLaunchKernelGGL(f32_bn_stats_kernel, dim3((C + 63) / 64, gy), dim3(256), 0, (hipStream_t)stream, x, M, C, sums);
    return launch_status();
}

static int f32_stats_stripes(int M) {
    int gy = M / 64;
    return gy < 1 ? 1 : (gy > 256 ? 256 : gy);
}

extern "C" size_t lp_f32_bn_stats_workspace_bytes(int M, int C) {
    return M > 0 && C > 0 ? (size_t)f32_stats_stripes(M) * 4 * 2 * (size_t)C * sizeof(float) : 0;
}

// sums += [sum x, sum x^2] as lp_f32_bn_stats, through per-stripe partial sums in `workspace` added in a fixed order: reproducible
extern "C" int lp_f32_bn_stats_ordered(const float* x, int M, int C, float* sums, void* workspace, size_t workspace_bytes, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(x && sums && workspace && M > 0 && C > 0 && workspace_bytes >= lp_f32_bn_stats_workspace_bytes(M, C));
    const int gy = f32_stats_stripes(M);
    hipLaunchKernelGGL(f32_bn_stats_part_kernel, dim3((C + 63) / 64, gy), dim3(256), 0, (hipStream_t)stream, x, M, C, (float*)workspace);
    hipLaunchKernelGGL(f32_bn_stats_fold_kernel, dim3((2 * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, gy * 4, C,
                       sums);
    return launch_status();
}

extern "C" int lp_f32_bn_apply(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                               const float* residual, int relu, int M, int C, float* y, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(x && mean && invstd && gamma && beta && y && M > 0 && C > 0);
    const size_t total = (size_t)M * C;
    hipLaunchKernelGGL(f32_bn_apply_kernel, dim3(f32_grid(total)), dim3(256), 0, (hipStream_t)stream, x, mean, invstd, gamma, beta, residual, relu,
                       total, C, y);
    return launch_status();
}

extern "C" int lp_f32_bn_bwd_reduce(const float* dy, const float* y_out, const float* x, const float* mean, const float* invstd, int M, int C,
                                    float* sums, float* dbeta_acc, float* dgamma_acc, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(dy && x && mean && invstd && sums && M > 0 && C > 0);
    int gy = M / 64;
    gy = gy < 1 ? 1 : (gy > 256 ? 256 : gy);
    hipLaunchKernelGGL(f32_bn_bwd_reduce_kernel, dim3((C + 63) / 64, gy), dim3(256), 0, (hipStream_t)stream, dy, y_out, x, mean, invstd, M, C, sums,
                       dbeta_acc, dgamma_acc);
    return launch_status();
}

extern "C" int lp_f32_bn_bwd_apply(const float* dy, const float* y_out, const float* x, const float* mean, const float* invstd,
                                   const float* gamma, const float* sums, float count, int M, int C, float* dx, float* dres,
                                   lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(dy && x && mean && invstd && gamma && sums && dx && M > 0 && C > 0 && count > 0.f);
    const size_t total = (size_t)M * C;
    hipLaunchKernelGGL(f32_bn_bwd_apply_kernel, dim3(f32_grid(total)), dim3(256), 0, (hipStream_t)stream, dy, y_out, x, mean, invstd, gamma, sums,
                       1.f / count, total, C, dx, dres);
    return launch_status();
}

extern "C" int lp_f32_maxpool_fwd(const float* x, int B, int Hi, int Wi, int C, float* y, void* argmax_u8, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(x && y && argmax_u8 && B > 0 && Hi > 0 && Wi > 0 && C > 0);
    const int Ho = (Hi - 1) / 2 + 1, Wo = (Wi - 1) / 2 + 1;
    hipLaunchKernelGGL(f32_maxpool_fwd_kernel, dim3(f32_grid((size_t)B * Ho * Wo * C)), dim3(256), 0, (hipStream_t)stream, x, B, Hi, Wi, C, Ho, Wo, y,
                       (unsigned char*)argmax_u8);
    return launch_status();
}

extern "C" int lp_f32_maxpool_bwd(const void* argmax_u8, const float* dy, int B, int Hi, int Wi, int C, float* dx, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(argmax_u8 && dy && dx && B > 0 && Hi > 0 && Wi > 0 && C > 0);
    const int Ho = (Hi - 1) / 2 + 1, Wo = (Wi - 1) / 2 + 1;
    hipLaunchKernelGGL(f32_maxpool_bwd_kernel, dim3(f32_grid((size_t)B * Hi * Wi * C)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned char*)argmax_u8, dy, B, Hi, Wi, C, Ho, Wo, dx);
    return launch_status();
}

extern "C" int lp_f32_images_to_nhwc4(const float* images, int B, int H, int W, float* out, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(images && out && B > 0 && H > 0 && W > 0);
    hipLaunchKernelGGL(f32_nchw3_to_nhwc4_kernel, dim3(f32_grid((size_t)B * H * W)), dim3(256), 0, (hipStream_t)stream, images, B, H * W, out);
    return launch_status();
}

extern "C" int lp_f32_pixel_shuffle(const float* in, int B, int h, int w, int c_out, int ld, int inverse, float* out, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(in && out && B > 0 && h > 0 && w > 0 && c_out > 0 && ld >= c_out);
    hipLaunchKernelGGL(f32_pixel_shuffle_kernel, dim3(f32_grid((size_t)B * h * w * c_out * 4)), dim3(256), 0, (hipStream_t)stream, in, B, h, w, c_out,
                       ld, inverse, out);
    return launch_status();
}

extern "C" int lp_f32_softmax2d_bwd(const float* prob, const float* gprob, int B, int K, int n, float* gin, long stride_b, long stride_i,
                                    long stride_k, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(prob && gprob && gin && B >= 0 && K > 0 && n > 0);
    if (B == 0) return LP_OK;
    hipLaunchKernelGGL(f32_softmax2d_bwd_kernel, dim3(B * K), dim3(256), 0, (hipStream_t)stream, prob, gprob, K, n, gin, stride_b, stride_i,
                       stride_k);
    return launch_status();
}
