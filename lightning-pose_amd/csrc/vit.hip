// Glue of the ViT-S/16 backbone between its GEMMs (which run on lp_gemm_nt / lp_conv_wgrad): patch extraction, token
// assembly with interpolated position embeddings, LayerNorm, GELU, attention soft-max, batched transposes.  gfx950.
//
// Replaces the HuggingFace `ViTModel` called by `VisionEncoder.forward` (lightning_pose/models/backbones/vit.py:16-49,
// selected by backbone = "vits_dino", models/backbones/factory.py:188-190):
//   ViTEmbeddings (patch projection 16x16/16, [CLS], bicubic-interpolated position embeddings), 12 x ViTLayer
//   (LayerNorm -> 6-head self-attention -> residual -> LayerNorm -> MLP with exact GELU -> residual), final LayerNorm.
// Policy (bf16-mixed): the residual stream, LayerNorm statistics and all reductions are fp32;
// GEMM operands are bf16.  Every kernel here is an HBM-bound stream or a per-row reduction.
#include "lp_common.h"

namespace lp {

__device__ __forceinline__ void unpack8v(const u16x8& v, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = bf16_to_f32(v[i]);
}

// ---- images (B,3,H,W) fp32 -> patch rows [B * (H/P) * (W/P)][3*P*P] bf16, k = (c, ky, kx) as Conv2d's weight.flatten(1) ---
__global__ __launch_bounds__(256) void vit_patchify_kernel(const float* __restrict__ img, int B, int H, int W, int P,
                                                           unsigned short* __restrict__ out) {
    const int gw = W / P, gh = H / P, kdim = 3 * P * P, chunks = kdim >> 3;
    const size_t total = (size_t)B * gh * gw * chunks;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (size_t)gridDim.x * 256) {
        const int ch = (int)(q % chunks);
        size_t p = q / chunks;
        const int px = (int)(p % gw);
        p /= gw;
        const int py = (int)(p % gh), b = (int)(p / gh);
        const int k = ch * 8, c = k / (P * P), ky = (k - c * P * P) / P, kx = k % P;  // 8 consecutive kx (P % 8 == 0)
        const float* src = img + (((size_t)b * 3 + c) * H + py * P + ky) * W + px * P + kx;
        float f[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = src[i];
        *reinterpret_cast<u16x8*>(out + q * 8) = pack_bf16x8(f);
    }
}

// ---- tokens: x[b][0] = cls + pos[0];  x[b][1+p] = patch[b][p] + pos[1+p]   (fp32 residual stream) ------------------
__global__ __launch_bounds__(256) void vit_tokens_fwd_kernel(const unsigned short* __restrict__ patch, const float* __restrict__ cls,
                                                             const float* __restrict__ pos, int B, int Np, int D,
                                                             float* __restrict__ x) {
    const int T = Np + 1, chunks = D >> 3;
    const size_t total = (size_t)B * T * chunks;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (size_t)gridDim.x * 256) {
        const int ch = (int)(q % chunks);
        const size_t row = q / chunks;
        const int t = (int)(row % T), b = (int)(row / T);
        float v[8];
        if (t == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = cls[ch * 8 + i];
        } else {
            unpack8v(*reinterpret_cast<const u16x8*>(patch + ((size_t)b * Np + t - 1) * D + ch * 8), v);
        }
        float* dst = x + row * D + ch * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) dst[i] = v[i] + pos[t * D + ch * 8 + i];
    }
}

// backward: dpatch[b][p] = bf16(dx[b][1+p]);  dpos[t] = sum_b dx[b][t]  (d cls = dpos[0])
__global__ __launch_bounds__(256) void vit_tokens_bwd_kernel(const float* __restrict__ dx, int B, int Np, int D,
                                                             unsigned short* __restrict__ dpatch, float* __restrict__ dpos) {
    const int T = Np + 1, chunks = D >> 3;
    const int total = T * chunks;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < total; q += gridDim.x * 256) {
        const int ch = q % chunks, t = q / chunks;
        float s[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) s[i] = 0.f;
        for (int b = 0; b < B; ++b) {
            const float* src = dx + ((size_t)b * T + t) * D + ch * 8;
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                v[i] = src[i];
                s[i] += v[i];
            }
            if (t > 0) *reinterpret_cast<u16x8*>(dpatch + ((size_t)b * Np + t - 1) * D + ch * 8) = pack_bf16x8(v);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) dpos[t * D + ch * 8 + i] = s[i];
    }
}

// ---- Y[r][d] (+)= sum_q Wm[r][q] X[q][d]  (or Wm^T): the position-embedding interpolation and its adjoint (tiny) ------
__global__ __launch_bounds__(256) void small_matmul_kernel(const float* __restrict__ Wm, const float* __restrict__ X, int R, int Q, int D,
                                                           int transpose_w, int accumulate, float* __restrict__ Y) {
    const int rows = transpose_w ? Q : R, inner = transpose_w ? R : Q;
    const int total = rows * D;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int d = i % D, r = i / D;
        float a = 0.f;
        for (int k = 0; k < inner; ++k) a = fmaf(transpose_w ? Wm[k * Q + r] : Wm[r * Q + k], X[k * D + d], a);
        Y[i] = accumulate ? Y[i] + a : a;
    }
}

// ---- LayerNorm over D (D % 4 == 0, D <= 1024), optional residual add in front ---------------------------------------------------------
// x_out = x (+ delta);  y = (x_out - mean) * rstd * gamma + beta.  `drop_T` > 0: rows with (row % drop_T) == 0 ([CLS]) are
// not written to y and the others are compacted (the feature map the head consumes).
// HALF a wave per row (round 6): a lane owns float4 pieces (piece index hl + 32 * pass, hl = lane & 31) - 16-B accesses on the fp32 streams,
// 8-B on the bf16 ones - so ViT-S's 384 columns are exactly 3 passes of 32 lanes (one wave per row covered them in 1.5 passes of 64: a quarter
// of the lanes idle) and a wave has TWO rows' loads in flight before the first of its two dependent reductions.  NP = passes that cover D.
constexpr int kLnMaxD = 1024;
__device__ __forceinline__ void unpack4(const u16x4& v, float (&f)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = bf16_to_f32(v[i]);
}
__device__ __forceinline__ u16x4 pack4(const float (&f)[4]) {
    const unsigned lo = pack_bf16x2(f[0], f[1]), hi = pack_bf16x2(f[2], f[3]);
    u16x4 v = {(unsigned short)(lo & 0xffffu), (unsigned short)(lo >> 16), (unsigned short)(hi & 0xffffu), (unsigned short)(hi >> 16)};
    return v;
}
__device__ __forceinline__ float half_wave_sum(float v) {   // over the 32 lanes of this lane's half
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

template <int NP>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const unsigned short* __restrict__ delta,
                                                            float* __restrict__ x_out, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps, int M, int D, int drop_T,
                                                            unsigned short* __restrict__ y, float* __restrict__ mean,
                                                            float* __restrict__ rstd) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, hl = lane & 31;
    const int pieces = D >> 2;
    for (int row0 = (blockIdx.x * 4 + wave) * 2; row0 < M; row0 += gridDim.x * 8) {
        const int row = row0 + half;
        const bool live = row < M;
        float v[NP][4];
        float s = 0.f;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int pc = hl + 32 * p;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[p][e] = 0.f;
            if (live && pc < pieces) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)row * D + pc * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[p][e] = xv[e];
                if (delta != nullptr) {
                    float d[4];
                    unpack4(*reinterpret_cast<const u16x4*>(delta + (size_t)row * D + pc * 4), d);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[p][e] += d[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) s += v[p][e];
            }
        }
        const float mu = half_wave_sum(s) / (float)D;
        float ss = 0.f;
#pragma unroll
        for (int p = 0; p < NP; ++p)
            if (hl + 32 * p < pieces) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = v[p][e] - mu;
                    ss = fmaf(d, d, ss);
                }
            }
        const float rs = 1.f / sqrtf(half_wave_sum(ss) / (float)D + eps);
        if (!live) continue;   // (after the shuffles: both halves of a wave take part in them)
        if (hl == 0) {
            mean[row] = mu;
            rstd[row] = rs;
        }
        int yrow = row;
        bool wy = true;
        if (drop_T > 0) {
            const int b = row / drop_T, t = row - b * drop_T;
            wy = t > 0;
            yrow = b * (drop_T - 1) + t - 1;
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int pc = hl + 32 * p;
            if (pc < pieces) {
                if (x_out != nullptr) {
                    const f32x4 o = {v[p][0], v[p][1], v[p][2], v[p][3]};
                    *reinterpret_cast<f32x4*>(x_out + (size_t)row * D + pc * 4) = o;
                }
                if (wy) {
                    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + pc * 4), bt = *reinterpret_cast<const f32x4*>(beta + pc * 4);
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = fmaf((v[p][e] - mu) * rs, g[e], bt[e]);
                    *reinterpret_cast<u16x4*>(y + (size_t)yrow * D + pc * 4) = pack4(o);
                }
            }
        }
    }
}

// dx += rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma;  per-workgroup partial d gamma / d beta -> atomics
// COLSUM (round 3): also the column sums of the bf16 stream gradient it writes - that tensor is the dy of the NEXT Linear backward, and its
// column sums are that layer's bias gradient: taken here, the weight gradient needs no bias pass and may run on the pipelined kernel.
// Half a wave per row, as the forward kernel; NP = passes of 32 lanes x 4 columns that cover D (3 for ViT-S, 6 for ViT-B).
template <bool COLSUM, int NP>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const unsigned short* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, int M, int D, int drop_T,
                                                            float* __restrict__ dx, unsigned short* __restrict__ dx_bf16,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ colsum) {
    __shared__ float red[COLSUM ? 3 : 2][8][128];  // [gamma|beta|column sum][wave, half][column of the current pass]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, hl = lane & 31;
    const int pieces = D >> 2;
    float ag[NP][4], ab[NP][4], gm[NP][4], ac[COLSUM ? NP : 1][4];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int pc = hl + 32 * p;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ag[p][e] = ab[p][e] = 0.f;
            ac[COLSUM ? p : 0][e] = 0.f;
            gm[p][e] = pc < pieces ? gamma[pc * 4 + e] : 0.f;
        }
    }
    for (int row0 = (blockIdx.x * 4 + wave) * 2; row0 < M; row0 += gridDim.x * 8) {
        const int row = row0 + half;
        const bool live = row < M;
        int yrow = row;
        bool has = live;
        if (drop_T > 0 && live) {
            const int b = row / drop_T, t = row - b * drop_T;
            has = t > 0;
            yrow = b * (drop_T - 1) + t - 1;
        }
        const float mu = live ? mean[row] : 0.f, rs = live ? rstd[row] : 0.f;
        float g[NP][4], xh[NP][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int pc = hl + 32 * p;
#pragma unroll
            for (int e = 0; e < 4; ++e) g[p][e] = xh[p][e] = 0.f;
            if (live && pc < pieces) {
                float d[4] = {0.f, 0.f, 0.f, 0.f};
                if (has) unpack4(*reinterpret_cast<const u16x4*>(dy + (size_t)yrow * D + pc * 4), d);
                const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)row * D + pc * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[p][e] = (xv[e] - mu) * rs;
                    ag[p][e] = fmaf(d[e], xh[p][e], ag[p][e]);
                    ab[p][e] += d[e];
                    g[p][e] = d[e] * gm[p][e];
                    s1 += g[p][e];
                    s2 = fmaf(g[p][e], xh[p][e], s2);
                }
            }
        }
        s1 = half_wave_sum(s1) / (float)D;
        s2 = half_wave_sum(s2) / (float)D;
        if (!live) continue;   // (after the shuffles)
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int pc = hl + 32 * p;
            if (pc < pieces) {
                f32x4* dst = reinterpret_cast<f32x4*>(dx + (size_t)row * D + pc * 4);
                f32x4 o = *dst;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] += rs * (g[p][e] - s1 - xh[p][e] * s2);
                *dst = o;
                if (dx_bf16 != nullptr) {  // the updated residual-stream gradient as the next GEMM's operand (saves a cast pass)
                    typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
                    const u32x2_t w = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
                    *reinterpret_cast<u32x2_t*>(dx_bf16 + (size_t)row * D + pc * 4) = w;
                    if (COLSUM) {   // (of the fp32 values: the reference's bias gradient sums the unrounded dy)
#pragma unroll
                        for (int e = 0; e < 4; ++e) ac[COLSUM ? p : 0][e] += o[e];
                    }
                }
            }
        }
    }
    // column sums of this workgroup: 4 waves x 2 halves -> LDS -> one atomic per column, one 128-column pass at a time
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        if (32 * p >= pieces) break;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[0][wave * 2 + half][hl * 4 + e] = ag[p][e];
            red[1][wave * 2 + half][hl * 4 + e] = ab[p][e];
            if (COLSUM) red[COLSUM ? 2 : 0][wave * 2 + half][hl * 4 + e] = ac[COLSUM ? p : 0][e];
        }
        __syncthreads();
        const int cl = threadIdx.x;            // column within this pass
        const int c = 128 * p + cl;
        if (cl < 128 && c < D) {
            auto total = [&](int k) {
                return ((red[k][0][cl] + red[k][1][cl]) + (red[k][2][cl] + red[k][3][cl])) + ((red[k][4][cl] + red[k][5][cl]) + (red[k][6][cl] + red[k][7][cl]));
            };
            atomicAdd(&dgamma[c], total(0));
            atomicAdd(&dbeta[c], total(1));
            if (COLSUM) atomicAdd(&colsum[c], total(COLSUM ? 2 : 0));
        }
    }
}

// ---- GELU (exact, erf) on bf16 streams --------------------------------------------------------------------------------
// (gelu8 / gelu8_bwd: lp_common.h - the Linear layers' store passes use the same two functions, conv_pipe.h: kEkGeluFwd / kEkGeluBwd)

__global__ __launch_bounds__(256) void gelu_fwd_kernel(const unsigned short* __restrict__ x, size_t n_chunks, unsigned short* __restrict__ y) {
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n_chunks; q += (size_t)gridDim.x * 256) {
        float v[8];
        unpack8v(*reinterpret_cast<const u16x8*>(x + q * 8), v);
        gelu8(v);
        *reinterpret_cast<u16x8*>(y + q * 8) = pack_bf16x8(v);
    }
}

__global__ __launch_bounds__(256) void gelu_bwd_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ dy,
                                                       size_t n_chunks, unsigned short* __restrict__ dx) {
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n_chunks; q += (size_t)gridDim.x * 256) {
        float v[8], d[8];
        unpack8v(*reinterpret_cast<const u16x8*>(x + q * 8), v);
        unpack8v(*reinterpret_cast<const u16x8*>(dy + q * 8), d);
        gelu8_bwd(d, v);
        *reinterpret_cast<u16x8*>(dx + q * 8) = pack_bf16x8(d);
    }
}

// GELU backward of a [rows][cols] tensor that also leaves the column sums of what it writes (= the bias gradient of the Linear layer in front
// of the activation).  grid = (column groups of 64 chunks, row blocks); a thread keeps ONE 8-column chunk and walks rows 4 apart.
__global__ __launch_bounds__(256) void gelu_bwd_colsum_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ dy, int rows,
                                                              int chunks, unsigned short* __restrict__ dx, float* __restrict__ colsum) {
    __shared__ float red[4][64][8];
    const int lane = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int ch = blockIdx.x * 64 + lane;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    if (ch < chunks) {
        for (int r = blockIdx.y * 4 + rl; r < rows; r += gridDim.y * 4) {
            const size_t q = (size_t)r * chunks + ch;
            float v[8], d[8];
            unpack8v(*reinterpret_cast<const u16x8*>(x + q * 8), v);
            unpack8v(*reinterpret_cast<const u16x8*>(dy + q * 8), d);
            gelu8_bwd(d, v);
            *reinterpret_cast<u16x8*>(dx + q * 8) = pack_bf16x8(d);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += d[i];   // (fp32: the reference's bias gradient sums the unrounded dy)
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) red[rl][lane][i] = acc[i];
    __syncthreads();
    for (int t = threadIdx.x; t < 64 * 8; t += 256) {
        const int l = t >> 3, i = t & 7;
        const int c = (blockIdx.x * 64 + l) * 8 + i;
        if (blockIdx.x * 64 + l < chunks) atomicAdd(&colsum[c], (red[0][l][i] + red[1][l][i]) + (red[2][l][i] + red[3][l][i]));
    }
}

// ---- attention soft-max over the n valid columns of bf16 rows of pitch ld (ld % 8 == 0, ld <= 1024), in place; pad columns are
// zeroed.  One wave per row; a lane owns 16-B chunks (chunk index lane + 64 * pass), so a 640-wide row is two accesses per lane.
constexpr int kSmPass = 2;
__global__ __launch_bounds__(256) void softmax_rows_fwd_kernel(unsigned short* __restrict__ s, int rows, int n, int ld, float scale) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunks = ld >> 3;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        unsigned short* p = s + (size_t)row * ld;
        float v[kSmPass][8];
        float mx = -INFINITY;
#pragma unroll
        for (int q = 0; q < kSmPass; ++q) {
            const int ch = lane + 64 * q;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[q][e] = -INFINITY;
            if (ch < chunks) {
                float f[8];
                unpack8v(*reinterpret_cast<const u16x8*>(p + ch * 8), f);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (ch * 8 + e < n) {
                        v[q][e] = f[e] * scale;
                        mx = fmaxf(mx, v[q][e]);
                    }
            }
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < kSmPass; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[q][e] = __expf(v[q][e] - mx);  // exp(-inf) = 0 for the pad / inactive slots
                sum += v[q][e];
            }
        const float inv = 1.f / wave_sum(sum);
#pragma unroll
        for (int q = 0; q < kSmPass; ++q) {
            const int ch = lane + 64 * q;
            if (ch < chunks) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = v[q][e] * inv;
                *reinterpret_cast<u16x8*>(p + ch * 8) = pack_bf16x8(o);
            }
        }
    }
}

// ds = scale * p * (dp - sum_j dp_j p_j), in place over dp; pad columns zeroed (p is zero there)
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const unsigned short* __restrict__ prob, unsigned short* __restrict__ dp,
                                                               int rows, int n, int ld, float scale) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunks = ld >> 3;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const unsigned short* p = prob + (size_t)row * ld;
        unsigned short* d = dp + (size_t)row * ld;
        float pv[kSmPass][8], dv[kSmPass][8];
        float dot = 0.f;
#pragma unroll
        for (int q = 0; q < kSmPass; ++q) {
            const int ch = lane + 64 * q;
#pragma unroll
            for (int e = 0; e < 8; ++e) pv[q][e] = dv[q][e] = 0.f;
            if (ch < chunks) {
                float a[8], b_[8];
                unpack8v(*reinterpret_cast<const u16x8*>(p + ch * 8), a);
                unpack8v(*reinterpret_cast<const u16x8*>(d + ch * 8), b_);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (ch * 8 + e < n) {
                        pv[q][e] = a[e];
                        dv[q][e] = b_[e];
                        dot = fmaf(a[e], b_[e], dot);
                    }
            }
        }
        dot = wave_sum(dot);
#pragma unroll
        for (int q = 0; q < kSmPass; ++q) {
            const int ch = lane + 64 * q;
            if (ch < chunks) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = scale * pv[q][e] * (dv[q][e] - dot);
                *reinterpret_cast<u16x8*>(d + ch * 8) = pack_bf16x8(o);
            }
        }
    }
}

// ---- D[row][h] = sum_d a[row][h*64 + d] * b[row][h*64 + d]  (attention backward: rowsum(dO o O) per head, the soft-max backward's
// row term).  One wave per row: lane l owns the 16-B chunk l (+ 64 per pass; 8 chunks per 64-wide head), an 8-lane butterfly finishes a head.
__global__ __launch_bounds__(256) void attn_rowdot_kernel(const unsigned short* __restrict__ a, const unsigned short* __restrict__ b, int rows,
                                                          int nh, int ld, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunks = nh * 8;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        for (int c0 = 0; c0 < chunks; c0 += 64) {  // 8 heads per pass (ViT-S: one pass, ViT-B: two); wave-uniform trip count
            const int c = c0 + lane;
            float dot = 0.f;
            if (c < chunks) {
                float x[8], y[8];
                unpack8v(*reinterpret_cast<const u16x8*>(a + (size_t)row * ld + c * 8), x);
                unpack8v(*reinterpret_cast<const u16x8*>(b + (size_t)row * ld + c * 8), y);
#pragma unroll
                for (int e = 0; e < 8; ++e) dot = fmaf(x[e], y[e], dot);
            }
#pragma unroll
            for (int m = 4; m >= 1; m >>= 1) dot += __shfl_xor(dot, m, 64);
            if ((lane & 7) == 0 && c < chunks) out[(size_t)row * nh + (c >> 3)] = dot;
        }
    }
}

// ---- batched 2-D transpose of bf16 matrices: out[z][c][r] = in[z][r][c], r < R, c < Cc; out columns [R, ldo) zeroed ----------
// 64 x 64 tiles through LDS; both global sides move 16 B per lane when the pitches / offsets allow it (they do for every
// caller here: pitches and batch strides are multiples of 8), 2 B otherwise.
__global__ __launch_bounds__(256) void transpose_batched_kernel(const unsigned short* __restrict__ in, int R, int Cc, int ldi, long long in_b,
                                                                long long in_h, unsigned short* __restrict__ out, int ldo, long long out_b,
                                                                long long out_h, int nh, int vec) {
    __shared__ unsigned short tile[64][72];  // row pitch 144 B: 16-B aligned rows, conflict-light column reads
    const int z = blockIdx.z, zb = z / nh, zh = z - zb * nh;
    const unsigned short* src = in + zb * in_b + zh * in_h;
    unsigned short* dst = out + zb * out_b + zh * out_h;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;  // the output pad region is covered by r tiles up to ldo
    if (vec) {
        // load: thread -> (row i, 8-column chunk j); 2 passes of 32 rows
        const int j = threadIdx.x & 7, i0 = threadIdx.x >> 3;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int i = i0 + 32 * p, r = r0 + i, c = c0 + j * 8;
            u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (r < R) {
                if (c + 8 <= Cc) {
                    v = *reinterpret_cast<const u16x8*>(src + (size_t)r * ldi + c);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (c + e < Cc) v[e] = src[(size_t)r * ldi + c + e];
                }
            }
            *reinterpret_cast<u16x8*>(&tile[i][j * 8]) = v;
        }
        __syncthreads();
        // store: thread -> (output row = input column ci, 8-row chunk jr)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int ci = (threadIdx.x >> 3) + 32 * p, jr = threadIdx.x & 7;
            const int c = c0 + ci, r = r0 + jr * 8;
            if (c < Cc && r < ldo) {
                u16x8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = tile[jr * 8 + e][ci];
                if (r + 8 <= ldo) {
                    *reinterpret_cast<u16x8*>(dst + (size_t)c * ldo + r) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (r + e < ldo) dst[(size_t)c * ldo + r + e] = v[e];
                }
            }
        }
        return;
    }
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < Cc) ? src[(size_t)r * ldi + c] : (unsigned short)0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < Cc && r < ldo) dst[(size_t)c * ldo + r] = tile[tx][i];
    }
}

static int vit_grid(size_t work_items) {
    size_t blocks = (work_items + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    return blocks < 1 ? 1 : (int)blocks;
}

}  // namespace lp

extern "C" int lp_vit_patchify(const float* images, int B, int H, int W, int patch, void* out_bf16, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(images && out_bf16 && B > 0 && H > 0 && W > 0 && patch > 0);
    if (patch % 8 != 0 || H % patch != 0 || W % patch != 0) return LP_ERR_UNSUPPORTED;
    const size_t work = (size_t)B * (H / patch) * (W / patch) * (3 * patch * patch / 8);
    hipLaunchKernelGGL(vit_patchify_kernel, dim3(vit_grid(work)), dim3(256), 0, (hipStream_t)stream, images, B, H, W, patch,
                       (unsigned short*)out_bf16);
    return launch_status();
}

extern "C" int lp_vit_tokens_fwd(const void* patch_bf16, const float* cls, const float* pos, int B, int Np, int D, float* x,
                                 lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(patch_bf16 && cls && pos && x && B > 0 && Np > 0 && D > 0);
    if (D % 8 != 0) return LP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(vit_tokens_fwd_kernel, dim3(vit_grid((size_t)B * (Np + 1) * (D / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)patch_bf16, cls, pos, B, Np, D, x);
    return launch_status();
}

extern "C" int lp_vit_tokens_bwd(const float* dx, int B, int Np, int D, void* dpatch_bf16, float* dpos, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(dx && dpatch_bf16 && dpos && B > 0 && Np > 0 && D > 0);
    if (D % 8 != 0) return LP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(vit_tokens_bwd_kernel, dim3(vit_grid((size_t)(Np + 1) * (D / 8))), dim3(256), 0, (hipStream_t)stream, dx, B, Np, D,
                       (unsigned short*)dpatch_bf16, dpos);
    return launch_status();
}

extern "C" int lp_small_matmul(const float* w, const float* x, int R, int Q, int D, int transpose_w, int accumulate, float* y,
                               lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(w && x && y && R > 0 && Q > 0 && D > 0);
    const int rows = transpose_w ? Q : R;
    hipLaunchKernelGGL(small_matmul_kernel, dim3(vit_grid((size_t)rows * D)), dim3(256), 0, (hipStream_t)stream, w, x, R, Q, D, transpose_w,
                       accumulate, y);
    return launch_status();
}

extern "C" int lp_layernorm_fwd(const float* x, const void* delta_bf16, float* x_out, const float* gamma, const float* beta, float eps,
                                int M, int D, int drop_T, void* y_bf16, float* mean, float* rstd, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(x && gamma && beta && y_bf16 && mean && rstd && M > 0 && D > 0 && drop_T >= 0 && (delta_bf16 == nullptr || x_out != nullptr));
    if (D > kLnMaxD || D % 4 != 0) return LP_ERR_UNSUPPORTED;
    int blocks = (M + 7) / 8;
    if (blocks > 256 * 8) blocks = 256 * 8;
#define LP_LN_FWD(NP_)                                                                                                                      \
    hipLaunchKernelGGL((layernorm_fwd_kernel<NP_>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (const unsigned short*)delta_bf16, \
                       x_out, gamma, beta, eps, M, D, drop_T, (unsigned short*)y_bf16, mean, rstd)
    switch ((D + 127) / 128) {
    case 1: LP_LN_FWD(1); break;
    case 2: LP_LN_FWD(2); break;
    case 3: LP_LN_FWD(3); break;
    case 4: LP_LN_FWD(4); break;
    case 5: case 6: LP_LN_FWD(6); break;
    default: LP_LN_FWD(8); break;
    }
#undef LP_LN_FWD
    return launch_status();
}

static int layernorm_bwd_impl(const void* dy_bf16, const float* x, const float* mean, const float* rstd, const float* gamma, int M, int D,
                              int drop_T, float* dx_acc, void* dx_bf16, float* dgamma_acc, float* dbeta_acc, lp_stream_t stream,
                              float* colsum_acc = nullptr) {
    using namespace lp;
    LP_REQUIRE(dy_bf16 && x && mean && rstd && gamma && dx_acc && dgamma_acc && dbeta_acc && M > 0 && D > 0 && drop_T >= 0);
    if (D > kLnMaxD || D % 4 != 0) return LP_ERR_UNSUPPORTED;
    int blocks = (M + 7) / 8;
    if (blocks > 2048) blocks = 2048;  // 8 waves per SIMD; also bounds the d gamma / d beta atomics per column
#define LP_LN_BWD(CS_, NP_)                                                                                                                   \
    hipLaunchKernelGGL((layernorm_bwd_kernel<CS_, NP_>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)dy_bf16, x, mean, \
                       rstd, gamma, M, D, drop_T, dx_acc, (unsigned short*)dx_bf16, dgamma_acc, dbeta_acc, colsum_acc)
#define LP_LN_BWD_NP(CS_)                        \
    switch ((D + 127) / 128) {                   \
    case 1: LP_LN_BWD(CS_, 1); break;            \
    case 2: LP_LN_BWD(CS_, 2); break;            \
    case 3: LP_LN_BWD(CS_, 3); break;            \
    case 4: LP_LN_BWD(CS_, 4); break;            \
    case 5: case 6: LP_LN_BWD(CS_, 6); break;    \
    default: LP_LN_BWD(CS_, 8); break;           \
    }
    if (colsum_acc != nullptr) {
        LP_LN_BWD_NP(true)
    } else {
        LP_LN_BWD_NP(false)
    }
#undef LP_LN_BWD_NP
#undef LP_LN_BWD
    return launch_status();
}

extern "C" int lp_layernorm_bwd(const void* dy_bf16, const float* x, const float* mean, const float* rstd, const float* gamma, int M, int D,
                                int drop_T, float* dx_acc, float* dgamma_acc, float* dbeta_acc, lp_stream_t stream) {
    return layernorm_bwd_impl(dy_bf16, x, mean, rstd, gamma, M, D, drop_T, dx_acc, nullptr, dgamma_acc, dbeta_acc, stream);
}

// same, and the updated dx_acc also leaves rounded to bf16 (the operand of the next Linear layer's backward)
extern "C" int lp_layernorm_bwd_bf16(const void* dy_bf16, const float* x, const float* mean, const float* rstd, const float* gamma, int M, int D,
                                     int drop_T, float* dx_acc, void* dx_bf16, float* dgamma_acc, float* dbeta_acc, lp_stream_t stream) {
    LP_REQUIRE(dx_bf16);
    return layernorm_bwd_impl(dy_bf16, x, mean, rstd, gamma, M, D, drop_T, dx_acc, dx_bf16, dgamma_acc, dbeta_acc, stream);
}

// ... and the column sums of dx_bf16 accumulated into colsum_acc[D]: the bias gradient of the Linear layer whose dy that tensor is
extern "C" int lp_layernorm_bwd_bf16_colsum(const void* dy_bf16, const float* x, const float* mean, const float* rstd, const float* gamma, int M,
                                            int D, int drop_T, float* dx_acc, void* dx_bf16, float* dgamma_acc, float* dbeta_acc,
                                            float* colsum_acc, lp_stream_t stream) {
    LP_REQUIRE(dx_bf16 && colsum_acc);
    return layernorm_bwd_impl(dy_bf16, x, mean, rstd, gamma, M, D, drop_T, dx_acc, dx_bf16, dgamma_acc, dbeta_acc, stream, colsum_acc);
}

extern "C" int lp_gelu_fwd(const void* x_bf16, size_t n, void* y_bf16, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(x_bf16 && y_bf16 && n > 0);
    if (n % 8 != 0) return LP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3(vit_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x_bf16, n / 8,
                       (unsigned short*)y_bf16);
    return launch_status();
}

extern "C" int lp_gelu_bwd(const void* x_bf16, const void* dy_bf16, size_t n, void* dx_bf16, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(x_bf16 && dy_bf16 && dx_bf16 && n > 0);
    if (n % 8 != 0) return LP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(vit_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x_bf16,
                       (const unsigned short*)dy_bf16, n / 8, (unsigned short*)dx_bf16);
    return launch_status();
}

extern "C" int lp_gelu_bwd_colsum(const void* x_bf16, const void* dy_bf16, int rows, int cols, void* dx_bf16, float* colsum_acc, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(x_bf16 && dy_bf16 && dx_bf16 && colsum_acc && rows > 0 && cols > 0);
    if (cols % 8 != 0) return LP_ERR_UNSUPPORTED;
    const int chunks = cols / 8, gx = (chunks + 63) / 64;
    int gy = (rows + 3) / 4;
    const int cap = (256 * 8 + gx - 1) / gx;   // ~8 workgroups per CU; bounds the atomics per column
    if (gy > cap) gy = cap;
    hipLaunchKernelGGL(gelu_bwd_colsum_kernel, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x_bf16,
                       (const unsigned short*)dy_bf16, rows, chunks, (unsigned short*)dx_bf16, colsum_acc);
    return launch_status();
}

extern "C" int lp_softmax_rows_fwd(void* s_bf16, int rows, int n, int ld, float scale, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(s_bf16 && rows > 0 && n > 0 && ld >= n);
    if (ld > 512 * kSmPass || ld % 8 != 0) return LP_ERR_UNSUPPORTED;
    int blocks = (rows + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(softmax_rows_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (unsigned short*)s_bf16, rows, n, ld, scale);
    return launch_status();
}

extern "C" int lp_softmax_rows_bwd(const void* p_bf16, void* dp_bf16, int rows, int n, int ld, float scale, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(p_bf16 && dp_bf16 && rows > 0 && n > 0 && ld >= n);
    if (ld > 512 * kSmPass || ld % 8 != 0) return LP_ERR_UNSUPPORTED;
    int blocks = (rows + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)p_bf16,
                       (unsigned short*)dp_bf16, rows, n, ld, scale);
    return launch_status();
}

extern "C" int lp_attn_rowdot(const void* a_bf16, const void* b_bf16, int rows, int nh, int ld, float* out, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(a_bf16 && b_bf16 && out && rows > 0 && nh > 0 && ld >= nh * 64);
    if (ld % 8 != 0) return LP_ERR_UNSUPPORTED;  // head dimension 64; one wave per row, 8 heads per pass
    int blocks = (rows + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(attn_rowdot_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)a_bf16,
                       (const unsigned short*)b_bf16, rows, nh, ld, out);
    return launch_status();
}

extern "C" int lp_transpose_batched(const void* in_bf16, int R, int Cc, int ldi, long long in_b, long long in_h, void* out_bf16, int ldo,
                                    long long out_b, long long out_h, int nb, int nh, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(in_bf16 && out_bf16 && R > 0 && Cc > 0 && ldi >= Cc && ldo >= R && nb > 0 && nh > 0);
    if ((long long)nb * nh > 65535) return LP_ERR_UNSUPPORTED;
    dim3 grid((Cc + 63) / 64, (ldo + 63) / 64, nb * nh);
    const int vec = (ldi % 8 == 0 && ldo % 8 == 0 && in_b % 8 == 0 && in_h % 8 == 0 && out_b % 8 == 0 && out_h % 8 == 0 &&
                     ((uintptr_t)in_bf16 % 16 == 0) && ((uintptr_t)out_bf16 % 16 == 0)) ? 1 : 0;
    hipLaunchKernelGGL(transpose_batched_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short*)in_bf16, R, Cc, ldi, in_b, in_h,
                       (unsigned short*)out_bf16, ldo, out_b, out_h, nh, vec);
    return launch_status();
}
