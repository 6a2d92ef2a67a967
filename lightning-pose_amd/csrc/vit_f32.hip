// fp32 VALIDATION forms of the ViT-S/16 glue and of its attention (gfx950; the Linear layers run on lp_f32_conv_* as 1x1 convolutions).
//
// The reference trains in fp32 only (lightning_pose/train.py:411-428 passes no `precision=`), so config C4 (VisionEncoder over the
// HuggingFace ViTModel, lightning_pose/models/backbones/vit.py:16-49) is held to BASELINE.json's 1e-4 through these kernels, exactly as
// csrc/fp32.hip does for the ResNet-50 trunk.  Nothing is fused and nothing is tuned: one wave per row, scalar loads, every tensor fp32.
// The bf16-mixed kernels of vit.hip / attn.hip are the product and the measured path.
#include "lp_common.h"

namespace lp {

// images (B,3,H,W) -> patch rows [B * gh * gw][3 P P], k = (c, ky, kx) as Conv2d's weight.flatten(1)
__global__ __launch_bounds__(256) void f32_vit_patchify_kernel(const float* __restrict__ img, int B, int H, int W, int P,
                                                               float* __restrict__ out) {
    const int gw = W / P, gh = H / P, kdim = 3 * P * P;
    const size_t total = (size_t)B * gh * gw * kdim;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (size_t)gridDim.x * 256) {
        const int k = (int)(q % kdim);
        size_t p = q / kdim;
        const int px = (int)(p % gw);
        p /= gw;
        const int py = (int)(p % gh), b = (int)(p / gh);
        const int c = k / (P * P), ky = (k - c * P * P) / P, kx = k % P;
        out[q] = img[(((size_t)b * 3 + c) * H + py * P + ky) * W + px * P + kx];
    }
}

// x[b][0] = cls + pos[0];  x[b][1 + p] = patch[b][p] + pos[1 + p]
__global__ __launch_bounds__(256) void f32_vit_tokens_fwd_kernel(const float* __restrict__ patch, const float* __restrict__ cls,
                                                                 const float* __restrict__ pos, int B, int Np, int D, float* __restrict__ x) {
    const int T = Np + 1;
    const size_t total = (size_t)B * T * D;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (size_t)gridDim.x * 256) {
        const int d = (int)(q % D);
        const size_t row = q / D;
        const int t = (int)(row % T), b = (int)(row / T);
        const float v = t == 0 ? cls[d] : patch[((size_t)b * Np + t - 1) * D + d];
        x[q] = v + pos[(size_t)t * D + d];
    }
}

// dpatch[b][p] = dx[b][1 + p];  dpos[t] = sum_b dx[b][t]
__global__ __launch_bounds__(256) void f32_vit_tokens_bwd_kernel(const float* __restrict__ dx, int B, int Np, int D, float* __restrict__ dpatch,
                                                                 float* __restrict__ dpos) {
    const int T = Np + 1, total = T * D;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < total; q += gridDim.x * 256) {
        const int d = q % D, t = q / D;
        float s = 0.f;
        for (int b = 0; b < B; ++b) {
            const float v = dx[((size_t)b * T + t) * D + d];
            s += v;
            if (t > 0) dpatch[((size_t)b * Np + t - 1) * D + d] = v;
        }
        dpos[q] = s;
    }
}

// LayerNorm over D, one wave per row; x_out = x (+ delta); y = (x_out - mean) rstd gamma + beta; rows with row % drop_T == 0 dropped from y
__global__ __launch_bounds__(256) void f32_layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ delta,
                                                                float* __restrict__ x_out, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float eps, int M, int D, int drop_T,
                                                                float* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        const float* xr = x + (size_t)row * D;
        float s = 0.f;
        for (int d = lane; d < D; d += 64) {
            float v = xr[d];
            if (delta != nullptr) {
                v += delta[(size_t)row * D + d];
                x_out[(size_t)row * D + d] = v;
            }
            s += v;
        }
        const float mu = wave_sum(s) / (float)D;
        const float* xs = delta != nullptr ? x_out + (size_t)row * D : xr;   // (each lane re-reads what it wrote itself)
        float q = 0.f;
        for (int d = lane; d < D; d += 64) {
            const float c = xs[d] - mu;
            q = fmaf(c, c, q);
        }
        const float rs = 1.f / sqrtf(wave_sum(q) / (float)D + eps);
        if (lane == 0) {
            mean[row] = mu;
            rstd[row] = rs;
        }
        if (drop_T > 0 && row % drop_T == 0) continue;
        const size_t orow = drop_T > 0 ? (size_t)(row - row / drop_T - 1) : (size_t)row;
        for (int d = lane; d < D; d += 64) y[orow * D + d] = fmaf((xs[d] - mu) * rs, gamma[d], beta[d]);
    }
}

// dx_acc += LayerNorm backward of dy (rows compacted as in the forward when drop_T > 0); dgamma / dbeta accumulated with atomics
__global__ __launch_bounds__(256) void f32_layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ gamma, int M, int D, int drop_T,
                                                                float* __restrict__ dx_acc, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        if (drop_T > 0 && row % drop_T == 0) continue;
        const size_t orow = drop_T > 0 ? (size_t)(row - row / drop_T - 1) : (size_t)row;
        const float mu = mean[row], rs = rstd[row];
        const float* xr = x + (size_t)row * D;
        const float* gr = dy + orow * D;
        float a = 0.f, b = 0.f;
        for (int d = lane; d < D; d += 64) {
            const float g = gr[d] * gamma[d], xh = (xr[d] - mu) * rs;
            a += g;
            b = fmaf(g, xh, b);
        }
        a = wave_sum(a) / (float)D;
        b = wave_sum(b) / (float)D;
        for (int d = lane; d < D; d += 64) {
            const float xh = (xr[d] - mu) * rs;
            dx_acc[(size_t)row * D + d] += rs * (gr[d] * gamma[d] - a - xh * b);
            atomicAdd(&dgamma[d], gr[d] * xh);
            atomicAdd(&dbeta[d], gr[d]);
        }
    }
}

// exact GELU (erf form, torch.nn.functional.gelu default - ViTConfig.hidden_act = "gelu")
__global__ __launch_bounds__(256) void f32_gelu_fwd_kernel(const float* __restrict__ x, size_t n, float* __restrict__ y) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = x[i];
        y[i] = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    }
}
__global__ __launch_bounds__(256) void f32_gelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, size_t n,
                                                           float* __restrict__ dx) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = x[i];
        const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752440f));
        const float pdf = 0.39894228040143267794f * expf(-0.5f * v * v);
        dx[i] = dy[i] * (cdf + v * pdf);
    }
}

// ---- attention, head dimension 64: one wave per (image, head, query row) ----------------------------------------------------------
// q / k / v are column blocks of the fused qkv tensor (row pitch ld); p [B][nh][T][T] keeps the probabilities for the backward pass
__global__ __launch_bounds__(256) void f32_attn_fwd_kernel(const float* __restrict__ qkv, int ld, int koff, int voff, int B, int nh, int T,
                                                           float scale, float* __restrict__ p, float* __restrict__ o, int ldo) {
    __shared__ float sq[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long rows = (long)B * nh * T;
    // (every wave of a workgroup runs the same number of rounds: the barriers below stay uniform)
    for (long r0 = (long)blockIdx.x * 4; r0 < rows; r0 += (long)gridDim.x * 4) {
        const long r = r0 + wave;
        const bool live = r < rows;
        const int i = live ? (int)(r % T) : 0;
        const long bh = live ? r / T : 0;
        const int h = (int)(bh % nh), b = (int)(bh / nh);
        const float* base = qkv + (size_t)b * T * ld + h * 64;
        __syncthreads();
        sq[wave][lane] = live ? base[(size_t)i * ld + lane] : 0.f;
        __syncthreads();
        if (!live) continue;
        float* pr = p + (size_t)r * T;
        float mx = -INFINITY;
        for (int j = lane; j < T; j += 64) {
            const float* kr = base + (size_t)j * ld + koff;
            float s = 0.f;
            for (int d = 0; d < 64; ++d) s = fmaf(sq[wave][d], kr[d], s);
            s *= scale;
            pr[j] = s;
            mx = fmaxf(mx, s);
        }
        mx = wave_max(mx);
        float sum = 0.f;
        for (int j = lane; j < T; j += 64) {
            const float e = expf(pr[j] - mx);
            pr[j] = e;
            sum += e;
        }
        const float inv = 1.f / wave_sum(sum);
        for (int j = lane; j < T; j += 64) pr[j] *= inv;
        __threadfence_block();   // the probabilities of the other lanes are read below
        __builtin_amdgcn_wave_barrier();
        float acc = 0.f;
        for (int j = 0; j < T; ++j) acc = fmaf(pr[j], base[(size_t)j * ld + voff + lane], acc);
        o[((size_t)b * T + i) * ldo + h * 64 + lane] = acc;
    }
}

// per query row: dP = dO V^T, dS = P o (dP - rowsum(P o dP)) (stored over ds), dQ = scale dS K
__global__ __launch_bounds__(256) void f32_attn_bwd_q_kernel(const float* __restrict__ qkv, int ld, int koff, int voff,
                                                             const float* __restrict__ dout, int ldo, const float* __restrict__ p, int B,
                                                             int nh, int T, float scale, float* __restrict__ ds, float* __restrict__ dqkv,
                                                             int ldd) {
    __shared__ float sdo[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long rows = (long)B * nh * T;
    for (long r0 = (long)blockIdx.x * 4; r0 < rows; r0 += (long)gridDim.x * 4) {
        const long r = r0 + wave;
        const bool live = r < rows;
        const int i = live ? (int)(r % T) : 0;
        const long bh = live ? r / T : 0;
        const int h = (int)(bh % nh), b = (int)(bh / nh);
        const float* base = qkv + (size_t)b * T * ld + h * 64;
        __syncthreads();
        sdo[wave][lane] = live ? dout[((size_t)b * T + i) * ldo + h * 64 + lane] : 0.f;
        __syncthreads();
        if (!live) continue;
        const float* pr = p + (size_t)r * T;
        float* dr = ds + (size_t)r * T;
        float dot = 0.f;
        for (int j = lane; j < T; j += 64) {
            const float* vr = base + (size_t)j * ld + voff;
            float s = 0.f;
            for (int d = 0; d < 64; ++d) s = fmaf(sdo[wave][d], vr[d], s);
            dr[j] = s;
            dot = fmaf(pr[j], s, dot);
        }
        dot = wave_sum(dot);
        for (int j = lane; j < T; j += 64) dr[j] = pr[j] * (dr[j] - dot);
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();
        float acc = 0.f;
        for (int j = 0; j < T; ++j) acc = fmaf(dr[j], base[(size_t)j * ld + koff + lane], acc);
        dqkv[((size_t)b * T + i) * ldd + h * 64 + lane] = acc * scale;
    }
}

// per key row: dV = P^T dO, dK = scale dS^T Q (lane = head dimension)
__global__ __launch_bounds__(256) void f32_attn_bwd_kv_kernel(const float* __restrict__ qkv, int ld, const float* __restrict__ dout, int ldo,
                                                              const float* __restrict__ p, const float* __restrict__ ds, int B, int nh,
                                                              int T, float scale, float* __restrict__ dqkv, int ldd, int koff, int voff) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long rows = (long)B * nh * T;
    for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
        const int j = (int)(r % T);
        const long bh = r / T;
        const int h = (int)(bh % nh), b = (int)(bh / nh);
        const float* qb = qkv + (size_t)b * T * ld + h * 64 + lane;
        const float* ob = dout + (size_t)b * T * ldo + h * 64 + lane;
        const float* pc = p + (size_t)bh * T * T + j;
        const float* dc = ds + (size_t)bh * T * T + j;
        float dv = 0.f, dk = 0.f;
        for (int i = 0; i < T; ++i) {
            dv = fmaf(pc[(size_t)i * T], ob[(size_t)i * ldo], dv);
            dk = fmaf(dc[(size_t)i * T], qb[(size_t)i * ld], dk);
        }
        float* dst = dqkv + ((size_t)b * T + j) * ldd + h * 64 + lane;
        dst[voff] = dv;
        dst[koff] = dk * scale;
    }
}

static int f32_grid(size_t work_items) {
    size_t blocks = (work_items + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    return blocks < 1 ? 1 : (int)blocks;
}
static int f32_row_grid(long rows) {
    long blocks = (rows + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    return blocks < 1 ? 1 : (int)blocks;
}

}  // namespace lp

extern "C" int lp_f32_vit_patchify(const float* images, int B, int H, int W, int patch, float* out, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(images && out && B > 0 && H > 0 && W > 0 && patch > 0);
    if (H % patch != 0 || W % patch != 0) return LP_ERR_UNSUPPORTED;
    const size_t total = (size_t)B * (H / patch) * (W / patch) * 3 * patch * patch;
    hipLaunchKernelGGL(f32_vit_patchify_kernel, dim3(f32_grid(total)), dim3(256), 0, (hipStream_t)stream, images, B, H, W, patch, out);
    return launch_status();
}

extern "C" int lp_f32_vit_tokens_fwd(const float* patch, const float* cls, const float* pos, int B, int Np, int D, float* x,
                                     lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(patch && cls && pos && x && B > 0 && Np > 0 && D > 0);
    hipLaunchKernelGGL(f32_vit_tokens_fwd_kernel, dim3(f32_grid((size_t)B * (Np + 1) * D)), dim3(256), 0, (hipStream_t)stream, patch, cls, pos, B, Np, D, x);
    return launch_status();
}

extern "C" int lp_f32_vit_tokens_bwd(const float* dx, int B, int Np, int D, float* dpatch, float* dpos, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(dx && dpatch && dpos && B > 0 && Np > 0 && D > 0);
    hipLaunchKernelGGL(f32_vit_tokens_bwd_kernel, dim3(f32_grid((size_t)(Np + 1) * D)), dim3(256), 0, (hipStream_t)stream, dx, B, Np, D, dpatch, dpos);
    return launch_status();
}

extern "C" int lp_f32_layernorm_fwd(const float* x, const float* delta, float* x_out, const float* gamma, const float* beta, float eps, int M,
                                    int D, int drop_T, float* y, float* mean, float* rstd, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(x && gamma && beta && y && mean && rstd && M > 0 && D > 0 && drop_T >= 0 && (delta == nullptr || x_out != nullptr));
    hipLaunchKernelGGL(f32_layernorm_fwd_kernel, dim3(f32_row_grid(M)), dim3(256), 0, (hipStream_t)stream, x, delta, x_out, gamma, beta, eps, M, D, drop_T, y, mean, rstd);
    return launch_status();
}

extern "C" int lp_f32_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, int M, int D,
                                    int drop_T, float* dx_acc, float* dgamma_acc, float* dbeta_acc, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(dy && x && mean && rstd && gamma && dx_acc && dgamma_acc && dbeta_acc && M > 0 && D > 0 && drop_T >= 0);
    hipLaunchKernelGGL(f32_layernorm_bwd_kernel, dim3(f32_row_grid(M)), dim3(256), 0, (hipStream_t)stream, dy, x, mean, rstd, gamma, M, D, drop_T, dx_acc, dgamma_acc,
                                                                                dbeta_acc);
    return launch_status();
}

extern "C" int lp_f32_gelu_fwd(const float* x, size_t n, float* y, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(x && y && n > 0);
    hipLaunchKernelGGL(f32_gelu_fwd_kernel, dim3(f32_grid(n)), dim3(256), 0, (hipStream_t)stream, x, n, y);
    return launch_status();
}

extern "C" int lp_f32_gelu_bwd(const float* x, const float* dy, size_t n, float* dx, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(x && dy && dx && n > 0);
    hipLaunchKernelGGL(f32_gelu_bwd_kernel, dim3(f32_grid(n)), dim3(256), 0, (hipStream_t)stream, x, dy, n, dx);
    return launch_status();
}

extern "C" int lp_f32_attn_fwd(const float* qkv, int ld, int k_off, int v_off, int B, int nh, int T, float scale, float* p, float* o, int ldo,
                               lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(qkv && p && o && B > 0 && nh > 0 && T > 0 && ld >= nh * 64 && ldo >= nh * 64);
    hipLaunchKernelGGL(f32_attn_fwd_kernel, dim3(f32_row_grid((long)B * nh * T)), dim3(256), 0, (hipStream_t)stream, qkv, ld, k_off, v_off, B, nh, T, scale, p, o, ldo);
    return launch_status();
}

extern "C" int lp_f32_attn_bwd(const float* qkv, int ld, int k_off, int v_off, const float* dout, int ldo, const float* p, int B, int nh, int T,
                               float scale, float* ds_workspace, float* dqkv, int ldd, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(qkv && dout && p && ds_workspace && dqkv && B > 0 && nh > 0 && T > 0 && ld >= nh * 64 && ldo >= nh * 64 && ldd >= nh * 64);
    const int grid = f32_row_grid((long)B * nh * T);
    hipLaunchKernelGGL(f32_attn_bwd_q_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, qkv, ld, k_off, v_off, dout, ldo, p, B, nh, T, scale, ds_workspace, dqkv, ldd);
    hipLaunchKernelGGL(f32_attn_bwd_kv_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, qkv, ld, dout, ldo, p, ds_workspace, B, nh, T, scale, dqkv, ldd, k_off, v_off);
    return launch_status();
}
