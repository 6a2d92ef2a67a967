// HBM-bound glue of the ResNet-50 trunk on NHWC bf16 activations: training-mode BatchNorm (statistics, apply +
// ReLU + residual, backward), 3x3/2 max-pool, input layout conversion and the head's PixelShuffle.  gfx950.
//
// Replaces the ATen/cuDNN BatchNorm2d, ReLU, MaxPool2d and PixelShuffle calls inside `self.backbone(images)` /
// `HeatmapHead.forward` (lightning_pose/models/base.py:398, models/heads/heatmap.py:44,208; torchvision Bottleneck
// semantics in SURVEY.md Appendix A).  Statistics are per call (= per forward pass: the labeled and the unlabeled
// batch are normalised separately, models/base.py:682-695) and are exposed as raw [sum, sum of squares] so that
// SyncBatchNorm (train.py:427) is one small all-reduce on that buffer before lp_bn_finalize.
//
// Every kernel moves 16 B per lane (8 bf16 channels), keeps per-channel scale/shift in registers and reduces in
// fp32.  Roofline: HBM; algorithmic bytes are listed per kernel in DESIGN.md.
#include <stdlib.h>

#include "lp_common.h"

namespace lp {

__device__ __forceinline__ void unpack8(const u16x8& v, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = bf16_to_f32(v[i]);
}

__device__ __forceinline__ u16x8 pack8(const float (&f)[8]) { return pack_bf16x8(f); }

// streaming 16-B load of data that is dead afterwards (keeps it from displacing reusable lines in L2 / MALL)
__device__ __forceinline__ u16x8 load_stream8(const unsigned short* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(p));
#else
    return *reinterpret_cast<const u16x8*>(p);
#endif
}

// ---- per-channel sums over rows: sums[0][c] += sum a, sums[1][c] += sum a*b -------------------------------------
// MODE 0: plain statistics (sum x, sum x^2)
// MODE 1: BN backward reductions: dz = relu-masked dy;  sum dz, sum dz * xhat
// Grid-stride over rows with register accumulators (4 independent 16-B loads in flight per lane) and ONE LDS reduction per workgroup, which
// leaves its 2 C partial sums in ITS row of `rows` ([gridDim.x][2][C] fp32); rows_reduce_kernel adds the rows in a fixed order and adds the
// totals into the fixed-point sums.  (Up to 1024 workgroups finish together here: adding their sums straight into the totals - fp32
// atomics until round 3, fx_add in a first version of round 4 - serialises ~26 ns per atomic and address, 27 / 52 us of a 110 us launch;
// the convolution store passes, <= 256 workgroups finishing at different times, do use fx_add.  profiles/r04m_*)
template <int MODE>
__global__ __launch_bounds__(256) void colreduce_kernel(const unsigned short* __restrict__ A, const unsigned short* __restrict__ Yout,
                                                        const unsigned short* __restrict__ X, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, int M, int C,
                                                        float* __restrict__ rows) {
    __shared__ float red[2][256][8];
    float* const row = rows + (size_t)blockIdx.x * 2 * C;
    const int chunks = C >> 3;                       // 16-B chunks per row
    const int cpb = chunks < 256 ? chunks : 256;     // chunks handled per block pass
    const int lanes_r = 256 / cpb;                   // row lanes
    const int ch = threadIdx.x % cpb, rl = threadIdx.x / cpb;
    for (int cbase = blockIdx.y * cpb; cbase < chunks; cbase += gridDim.y * cpb) {
        const int c = (cbase + ch) * 8;
        float s0[8], s1[8], mu[8], is[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) s0[i] = s1[i] = 0.f;
        const bool cv = (cbase + ch) < chunks && rl < lanes_r;
        if (MODE == 1 && cv) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                mu[i] = mean[c + i];
                is[i] = invstd[c + i];
            }
        }
        if (cv) {
            const int rstep = gridDim.x * lanes_r;
            for (int r = blockIdx.x * lanes_r + rl; r < M; r += rstep) {
                const size_t off = (size_t)r * C + c;
                float a[8];
                unpack8(*reinterpret_cast<const u16x8*>(A + off), a);
                if (MODE == 0) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        s0[i] += a[i];
                        s1[i] = fmaf(a[i], a[i], s1[i]);
                    }
                } else {
                    float x[8];
                    unpack8(*reinterpret_cast<const u16x8*>(X + off), x);
                    if (Yout != nullptr) {
                        const u16x8 yo = *reinterpret_cast<const u16x8*>(Yout + off);
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                            if (!bf16_positive(yo[i])) a[i] = 0.f;  // relu'(y): y <= 0
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        s0[i] += a[i];
                        s1[i] = fmaf(a[i], (x[i] - mu[i]) * is[i], s1[i]);
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            red[0][threadIdx.x][i] = s0[i];
            red[1][threadIdx.x][i] = s1[i];
        }
        __syncthreads();
        // every thread finishes one (chunk, component) pair: 256 threads >= cpb * 8 pairs only when cpb <= 32; loop otherwise
        for (int pair = threadIdx.x; pair < cpb * 8; pair += 256) {
            const int pc = pair >> 3, pi = pair & 7;
            if (cbase + pc < chunks) {
                float t0 = 0.f, t1 = 0.f;
                for (int q = 0; q < lanes_r; ++q) {
                    t0 += red[0][q * cpb + pc][pi];
                    t1 += red[1][q * cpb + pc][pi];
                }
                const int cc = (cbase + pc) * 8 + pi;
                row[cc] = t0;
                row[C + cc] = t1;
            }
        }
        __syncthreads();
    }
}

// sums[comp][c] += the total over the partial-sum rows ([nrows][2][C] fp32) of (comp, c), added in an order that depends on nothing but
// nrows: 32 interleaved chains (rows g, g + 32, ...) per element, joined by a fixed pairwise tree.  One 256-thread workgroup covers the 16
// channels [c0, c0 + 16) of both components with float4 loads, 8 in flight per thread (the rows were written all over the chip: every
// load is an L2 miss, and this launch sits between a reduction and the BatchNorm pass that waits for it).
__global__ __launch_bounds__(256) void rows_reduce_kernel(const float* __restrict__ rows, int nrows, int C, lp_fxsum* __restrict__ sums) {
    __shared__ float red[32][32];
    const int c0 = blockIdx.x * 16;
    const int pq = (int)threadIdx.x % 8, rg = (int)threadIdx.x / 8;   // pq: (component, quad of channels); rg: chain
    const float* src = rows + (pq >> 2) * C + c0 + (pq & 3) * 4;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    if (c0 + (pq & 3) * 4 < C) {   // (C is a multiple of 8: the last workgroup may cover 8 channels only)
        constexpr int U = 8;
        for (int r = rg; r < nrows; r += 32 * U) {
            f32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int ru = r + u * 32;
                v[u] = ru < nrows ? *reinterpret_cast<const f32x4*>(src + (size_t)ru * 2 * C) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < U; ++u) t0 += v[u][0], t1 += v[u][1], t2 += v[u][2], t3 += v[u][3];
        }
    }
    red[rg][pq * 4 + 0] = t0, red[rg][pq * 4 + 1] = t1, red[rg][pq * 4 + 2] = t2, red[rg][pq * 4 + 3] = t3;
    __syncthreads();
    if ((int)threadIdx.x < 32) {
        float u[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) u[i] = red[i][threadIdx.x];
#pragma unroll
        for (int w = 16; w >= 1; w >>= 1)
#pragma unroll
            for (int i = 0; i < w; ++i) u[i] += u[i + w];
        const int comp = (int)threadIdx.x >> 4, c = c0 + ((int)threadIdx.x & 15);
        if (c < C) fx_add(&sums[(size_t)comp * C + c], u[0]);
    }
}

__device__ __forceinline__ float sum_value(const float* p) { return *p; }          // (fp32 sums: the fp32 validation executor's)
__device__ __forceinline__ float sum_value(const lp_fxsum* p) { return fx_value(p); }

// mean / invstd from [sum, sumsq]; running statistics updated with torch's momentum rule (unbiased variance)
// `nseg` segments ([seg][2][C] sums -> [seg][C] moments), running statistics updated segment by segment in order
struct BnMoments {
    float mu, var, invstd;
};
__device__ __forceinline__ BnMoments bn_moments(float s1, float s2, float count, float eps) {
    BnMoments m;
    m.mu = s1 / count;
    float var = s2 / count - m.mu * m.mu;
    m.var = var < 0.f ? 0.f : var;   // (not fmaxf: a NaN - a poisoned sum - must reach invstd and the running variance too)
    m.invstd = 1.f / sqrtf(m.var + eps);
    return m;
}
__device__ __forceinline__ void bn_running_update(const BnMoments& m, float count, float momentum, float* running_mean, float* running_var) {
    const float unbiased = count > 1.f ? m.var * count / (count - 1.f) : m.var;
    *running_mean = (1.f - momentum) * *running_mean + momentum * m.mu;
    *running_var = (1.f - momentum) * *running_var + momentum * unbiased;
}
template <typename SumT>
__global__ void bn_finalize_kernel(const SumT* __restrict__ sums, float count0, float count1, int nseg, int C, float eps, float momentum,
                                   float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ running_mean,
                                   float* __restrict__ running_var) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    for (int sg = 0; sg < nseg; ++sg) {
        const float count = sg == 0 ? count0 : count1;
        const SumT* sm = sums + (size_t)sg * 2 * C;
        const BnMoments m = bn_moments(sum_value(&sm[c]), sum_value(&sm[C + c]), count, eps);
        mean[sg * C + c] = m.mu;
        invstd[sg * C + c] = m.invstd;
        if (running_mean != nullptr) bn_running_update(m, count, momentum, &running_mean[c], &running_var[c]);
    }
}

// y = relu?( (x - mean) * invstd * gamma + beta (+ residual) )
// The grid stride is a multiple of the row length (host guarantees it), so a lane keeps ONE channel chunk for its whole walk:
// its 8 (mean, scale, shift) triples live in registers and the loop is nothing but 16-B streams, 4 chunks in flight per lane.
// RBN = true (round 5, lp_bn_apply_seg_rbn): `residual` is the PRE-normalisation tensor of the block's projection shortcut and `rb` its
// BatchNorm's terms - the shortcut is normalised here, rounded to bf16 exactly as its own lp_bn_apply pass stored it, and added; that pass (a
// write and a read of a block-output-sized tensor) is gone.  Results bit-identical to lp_bn_apply(shortcut) -> lp_bn_apply(..., residual).
struct BnResidualBn {
    const float *mean, *invstd, *gamma, *beta;   // mean / invstd: [segments][C]
};
// (launch bounds: the plain walk needs 5 waves per SIMD = at most 96 registers - its grid is ONE resident round of 5 workgroups per CU)
// Round 6 measured folding lp_bn_finalize into this launch (every lane converting its 8 channels' fixed-point sums, the first C / 8 lanes
// storing the moments and the running statistics): bit-identical and 2.4 % SLOWER per step (42.15 -> 43.12 ms, three alternating pairs) - the
// 327 k lanes re-read 256 B of sums each where they read 64 B of finished moments, a third more load traffic than the launch's own stream;
// retired with its measurement (profiles/retired/r06_bn_apply_fin.patch, profiles/r06b_step_ab.txt, r06c_step_ab.txt).
// LO = true (round 6, lp_bn_apply_seg_lo: the "fp32 residual stream" policy, LP_RESIDUAL_FP32=1 - NOT the benchmarked default): a block output
// o is kept as a bf16 PAIR, hi = bf16(o) (what every convolution reads, as before) and lo = bf16(o - hi), and the next identity block adds
// hi + lo (16 mantissa bits) instead of hi; a projection shortcut normalised in this pass (RBN) is added unrounded.  Own instantiations (4 waves
// per SIMD: the second residual stream is 16 more registers), so the default walk is untouched.  What it buys and costs: DESIGN.md section 3.
struct BnResidualLo {
    const unsigned short* residual_lo;   // [M][C] bf16 or nullptr
    unsigned short* y_lo;                // [M][C] bf16 or nullptr
};
template <bool RBN, bool LO = false>
__global__ __launch_bounds__(256, (RBN || LO) ? 4 : 5) void bn_apply_kernel(const unsigned short* __restrict__ X, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const unsigned short* __restrict__ residual,
                                                       int relu, size_t n_total, int C, unsigned short* __restrict__ Y,
                                                       unsigned char* __restrict__ bits, size_t seg_chunk, BnResidualBn rb,
                                                       BnResidualLo rl = BnResidualLo{}) {
    // Two BatchNorm segments in one launch (seg_chunk > 0: chunks [0, seg_chunk) use mean / invstd row 0, the rest row 1; the boundary is
    // a whole number of rows): the walk runs once per segment with that segment's terms in registers; a lane keeps its channel chunk
    // because every start is congruent to its global index modulo the stride.
    const int chunks = C >> 3;
    const size_t stride = (size_t)gridDim.x * 256;
    const size_t q0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int c = (int)(q0 % chunks) * 8;
    const int nseg = seg_chunk > 0 ? 2 : 1;
    for (int sg = 0; sg < nseg; ++sg) {
    const size_t lo = sg == 0 ? 0 : seg_chunk, n_chunks = (nseg == 2 && sg == 0) ? seg_chunk : n_total;
    size_t q = q0 >= lo ? q0 : q0 + (lo - q0 + stride - 1) / stride * stride;
    float mu[8], sc[8], be[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        mu[i] = mean[sg * C + c + i];
        sc[i] = invstd[sg * C + c + i] * gamma[c + i];
        be[i] = beta[c + i];
    }
    float mud[RBN ? 8 : 1], scd[RBN ? 8 : 1], bed[RBN ? 8 : 1];
    if (RBN) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            mud[RBN ? i : 0] = rb.mean[sg * C + c + i];
            scd[RBN ? i : 0] = rb.invstd[sg * C + c + i] * rb.gamma[c + i];
            bed[RBN ? i : 0] = rb.beta[c + i];
        }
    }
    constexpr int U = 4;
    for (; q < n_chunks; q += U * stride) {
        constexpr bool kRLo = LO && !RBN;   // (a projection shortcut has no lo word: it is added unrounded)
        u16x8 xv[U], rv[U], rlv[kRLo ? U : 1];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (q + u * stride < n_chunks) xv[u] = load_stream8(X + (q + u * stride) * 8);
        if (residual != nullptr) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (q + u * stride < n_chunks) rv[u] = *reinterpret_cast<const u16x8*>(residual + (q + u * stride) * 8);
        }
        if (kRLo && rl.residual_lo != nullptr) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (q + u * stride < n_chunks) rlv[kRLo ? u : 0] = load_stream8(rl.residual_lo + (q + u * stride) * 8);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (q + u * stride >= n_chunks) break;
            float x[8], o[8];
            unpack8(xv[u], x);
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = fmaf(x[i] - mu[i], sc[i], be[i]);
            if (residual != nullptr) {
                float r[8];
                unpack8(rv[u], r);
                if (RBN) {   // the shortcut's own lp_bn_apply (no ReLU), rounded to bf16 as that pass stored it (LO: added unrounded)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float nd = fmaf(r[i] - mud[RBN ? i : 0], scd[RBN ? i : 0], bed[RBN ? i : 0]);
                        r[i] = LO ? nd : bf16_to_f32(f32_to_bf16(nd));
                    }
                }
                if (kRLo && rl.residual_lo != nullptr) {
                    float rl8[8];
                    unpack8(rlv[kRLo ? u : 0], rl8);
#pragma unroll
                    for (int i = 0; i < 8; ++i) r[i] += rl8[i];
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] += r[i];
            }
            if (relu) {
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = fmaxf(o[i], 0.f);
            }
            const u16x8 hi8 = pack8(o);
            *reinterpret_cast<u16x8*>(Y + (q + u * stride) * 8) = hi8;
            if (LO && rl.y_lo != nullptr) {   // what bf16 dropped of this output, itself in bf16
                float h8[8], l8[8];
                unpack8(hi8, h8);
#pragma unroll
                for (int i = 0; i < 8; ++i) l8[i] = o[i] - h8[i];
                *reinterpret_cast<u16x8*>(rl.y_lo + (q + u * stride) * 8) = pack8(l8);
            }
            if (bits != nullptr) {  // 1-bit ReLU mask: o > 2^-134 is exactly "the stored bf16 is > 0"
                unsigned m = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) m |= (o[i] > 0x1p-134f ? 1u : 0u) << i;
                bits[q + u * stride] = (unsigned char)m;
            }
        }
    }
    }
}

// d beta[c] += sum over the segments of local[seg][0][c], d gamma[c] += ... [1][c]: the parameter gradients of a BatchNorm ARE its two backward
// sums (of THIS rank: `local` is the buffer before any SyncBatchNorm exchange).  Done by the kernel that consumes the sums, one thread of the
// whole grid per channel (plain read-modify-write, a single pass for any grid of C / 256 workgroups or more: a workgroup that did all C
// channels alone started its walk ~10 us late and the launch ended that much later, profiles/r04p_*) - now that the sums no longer travel
// through per-address fp32 atomics.
__device__ __forceinline__ void bn_param_grads(const lp_fxsum* __restrict__ local, int nseg, int C, float* __restrict__ dbeta,
                                               float* __restrict__ dgamma) {
    if (local == nullptr) return;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x) {
        float b = 0.f, g = 0.f;
        for (int sg = 0; sg < nseg; ++sg) {
            b += fx_value(&local[(size_t)sg * 2 * C + c]);
            g += fx_value(&local[(size_t)sg * 2 * C + C + c]);
        }
        if (dbeta != nullptr) dbeta[c] += b;
        if (dgamma != nullptr) dgamma[c] += g;
    }
}

// dx = gamma * invstd * (dz - sum(dz)/N - xhat * sum(dz*xhat)/N);  dz = relu-masked dy; optionally dz is also
// written out (gradient of the residual branch).  Same walk as bn_apply_kernel: per-channel terms in registers.
// `sums`: the (possibly all-reduced) totals the correction terms use, nullptr = no batch-statistics terms (eval-mode BatchNorm: a fixed
// affine map); `local` / dbeta / dgamma: see bn_param_grads.
constexpr int kBnBwdMaxC = 2048;   // widest BatchNorm bn_bwd_apply_kernel takes (ResNet-50's layer4): 32 KB of LDS for its correction terms
// (launch bounds: 5 waves per SIMD = 96 VGPRs, as before the fixed-point sums; the allocator spills two values, outside the chunk loop;
// without the bound: 102 VGPRs = 4 waves, the same speed - profiles/r04o_*)
// TERMS = true (round 5): the correction terms arrive as plain floats `terms` [segment][2][C] (bn_bwd_terms_kernel below converted them and
// added the parameter gradients, one tiny launch in front of this one): no LDS table, no barrier, no 64-bit conversions in the streaming
// kernel - its 1280 workgroups (ONE resident round: every one of them used to sit in that prologue at the same time, with HBM idle;
// ~10 us of a 122 us launch, profiles/r04r_bn_bwd_prologue.txt) start streaming after one L2 round trip.  TERMS = false keeps the
// self-contained form for callers without a workspace (lp_bn_bwd_apply(..., terms_ws = NULL)).
// DS = true (round 5, lp_bn_bwd_apply_seg_ds): the masked gradient this pass reads is ALSO the input of the block's projection-shortcut
// BatchNorm backward, whose two reductions [sum dz, sum dz * xhat_d] used to cost a pass of their own over dz and that BatchNorm's
// pre-normalisation tensor zd (lp_bn_bwd_reduce).  Here the walk reads zd as a fourth stream and leaves the workgroup's partial sums per segment
// in its row of `ds_rows` ([segment][gridDim.x][2][C] fp32, colreduce_kernel's row format: rows_reduce_kernel adds them in a fixed order) -
// dz is read once for both BatchNorms.  Its own instantiation (3 waves per SIMD: the accumulators and zd's constants are 40 more registers).
#ifndef LP_BN_DS_WAVES
#define LP_BN_DS_WAVES 3   // waves per SIMD of the DS instantiation: 133 registers, no scratch (4 = capped at 128, five spilled: the same speed, profiles/r05x_ds_waves.txt)
#endif
struct BnBwdDs {
    const unsigned short* zd;          // [M][C] bf16
    const float *mean_d, *invstd_d;    // [segments][C]
    float* rows;                       // [segments][gridDim.x][2][C]
};
template <bool TERMS, bool DS = false>
__global__ __launch_bounds__(256, DS ? LP_BN_DS_WAVES : 5) void bn_bwd_apply_kernel(const unsigned short* __restrict__ DY, const unsigned short* __restrict__ Yout,
                                                           const unsigned short* __restrict__ X, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           const lp_fxsum* __restrict__ sums, float inv_count, size_t n_total, int C,
                                                           unsigned short* __restrict__ DX, unsigned short* __restrict__ DRES,
                                                           size_t seg_chunk, float inv_count1, const lp_fxsum* __restrict__ local,
                                                           float* __restrict__ dbeta, float* __restrict__ dgamma,
                                                           const float* __restrict__ terms, BnBwdDs ds = BnBwdDs{}) {
    // (segments as in bn_apply_kernel: mean / invstd rows of C, sums rows of 2 C, one 1 / count per segment)
    const int chunks = C >> 3;
    const size_t stride = (size_t)gridDim.x * 256;
    const size_t q0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int c = (int)(q0 % chunks) * 8;
    const int nseg = seg_chunk > 0 ? 2 : 1;
    // The correction terms sum / count of the whole launch ([segment][2][C], C <= kBnBwdMaxC: host-checked) are converted from fixed point
    // ONCE per workgroup, a few per thread, into LDS.  Converted per lane (16 values per segment, four registers each while in flight) the
    // kernel needed 130 instead of 94 VGPRs = 3 instead of 5 waves per SIMD and the step lost 2.6 ms (profiles/r04m_*).  nullptr: zeros.
    __shared__ float kterm[TERMS ? 1 : 2 * 2 * kBnBwdMaxC];
    __shared__ float dsred[DS ? 2 : 1][DS ? 256 : 1][8];
    if (!TERMS) {
        bn_param_grads(local, nseg, C, dbeta, dgamma);
        __builtin_amdgcn_sched_barrier(0);   // (keeps the parameter-gradient code's registers out of the walk's allocation)
    }
    // segment 0's per-channel terms are requested BEFORE the conversion and its barrier, so the workgroup waits one memory latency, not two
    float mu[8], is[8], ga[8], k0[8], k1[8];
    if (!TERMS) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            mu[i] = mean[c + i];
            is[i] = invstd[c + i];
            ga[i] = gamma[c + i];
        }
        constexpr int UC = 8;   // conversions in flight per thread (C = 2048, two segments: 32 per thread = 4 round trips to L2)
        const int n = nseg * 2 * C;
        for (int e0 = threadIdx.x; e0 < n; e0 += 256 * UC) {
            long long hi[UC], lo[UC];
#pragma unroll
            for (int u = 0; u < UC; ++u) {
                const int e = e0 + 256 * u;
                hi[u] = (sums != nullptr && e < n) ? sums[e].hi : 0;
                lo[u] = (sums != nullptr && e < n) ? sums[e].lo : 0;
            }
#pragma unroll
            for (int u = 0; u < UC; ++u) {
                const int e = e0 + 256 * u;
                const lp_fxsum v{hi[u], lo[u]};
                if (e < n) kterm[e] = fx_value(&v) * (e < 2 * C ? inv_count : inv_count1);
            }
        }
        __syncthreads();
    }
    for (int sg = 0; sg < nseg; ++sg) {
    const size_t lo = sg == 0 ? 0 : seg_chunk, n_chunks = (nseg == 2 && sg == 0) ? seg_chunk : n_total;
    size_t q = q0 >= lo ? q0 : q0 + (lo - q0 + stride - 1) / stride * stride;
    const float* kt = TERMS ? terms : kterm;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (TERMS || sg > 0) {
            mu[i] = mean[sg * C + c + i];
            is[i] = invstd[sg * C + c + i];
        }
        ga[i] = gamma[c + i] * is[i];
        k0[i] = kt[sg * 2 * C + c + i];
        k1[i] = kt[sg * 2 * C + C + c + i] * is[i];   // (xhat k1 = (x - mean) (invstd k1): one register array less in the walk)
    }
    float mud[DS ? 8 : 1], isd[DS ? 8 : 1], sd0[DS ? 8 : 1], sd1[DS ? 8 : 1];
    if (DS) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            mud[DS ? i : 0] = ds.mean_d[sg * C + c + i];
            isd[DS ? i : 0] = ds.invstd_d[sg * C + c + i];
            sd0[DS ? i : 0] = sd1[DS ? i : 0] = 0.f;
        }
    }
    constexpr int U = 2;
    for (; q < n_chunks; q += U * stride) {
        u16x8 dv[U], xv[U], yv[U], zdv[DS ? U : 1];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (q + u * stride < n_chunks) {
                dv[u] = load_stream8(DY + (q + u * stride) * 8);
                xv[u] = load_stream8(X + (q + u * stride) * 8);
                if (Yout != nullptr) yv[u] = *reinterpret_cast<const u16x8*>(Yout + (q + u * stride) * 8);
                if (DS) zdv[DS ? u : 0] = load_stream8(ds.zd + (q + u * stride) * 8);
            }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (q + u * stride >= n_chunks) break;
            float dz[8], x[8], o[8];
            unpack8(dv[u], dz);
            unpack8(xv[u], x);
            if (Yout != nullptr) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (!bf16_positive(yv[u][i])) dz[i] = 0.f;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                o[i] = ga[i] * (dz[i] - k0[i] - (x[i] - mu[i]) * k1[i]);
            }
            *reinterpret_cast<u16x8*>(DX + (q + u * stride) * 8) = pack8(o);
            if (DRES != nullptr) *reinterpret_cast<u16x8*>(DRES + (q + u * stride) * 8) = pack8(dz);
            if (DS) {   // colreduce_kernel<1>'s terms, on the values the pass already holds
                float zd[8];
                unpack8(zdv[DS ? u : 0], zd);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    sd0[DS ? i : 0] += dz[i];
                    sd1[DS ? i : 0] = fmaf(dz[i], (zd[i] - mud[DS ? i : 0]) * isd[DS ? i : 0], sd1[DS ? i : 0]);
                }
            }
        }
    }
    if (DS) {   // the workgroup's partial sums of this segment -> its row (threads tid, tid + chunks, ... hold the same channels: 256 % chunks == 0)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            dsred[0][threadIdx.x][i] = sd0[DS ? i : 0];
            dsred[DS ? 1 : 0][threadIdx.x][i] = sd1[DS ? i : 0];
        }
        __syncthreads();
        const int lanes_r = 256 / chunks;
        float* const row = ds.rows + ((size_t)sg * gridDim.x + blockIdx.x) * 2 * C;
        for (int pair = threadIdx.x; pair < chunks * 8; pair += 256) {
            const int pc = pair >> 3, pi = pair & 7;
            float t0 = 0.f, t1 = 0.f;
            for (int r = 0; r < lanes_r; ++r) {
                t0 += dsred[0][r * chunks + pc][pi];
                t1 += dsred[DS ? 1 : 0][r * chunks + pc][pi];
            }
            // (thread tid holds chunk (blockIdx.x * 256 + tid) % chunks = tid % chunks)
            row[pc * 8 + pi] = t0;
            row[C + pc * 8 + pi] = t1;
        }
        __syncthreads();
    }
    }
}

// The launch in front of bn_bwd_apply_kernel<true>: terms[seg][2][C] = sum / count as floats (zeros without `sums`: eval-mode BatchNorm),
// and the BatchNorm's parameter gradients (bn_param_grads) - one thread per value.
__global__ __launch_bounds__(256) void bn_bwd_terms_kernel(const lp_fxsum* __restrict__ sums, float inv_count0, float inv_count1, int C, int nseg,
                                                           float* __restrict__ terms, const lp_fxsum* __restrict__ local, float* __restrict__ dbeta,
                                                           float* __restrict__ dgamma) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < nseg * 2 * C) terms[e] = sums != nullptr ? fx_value(&sums[e]) * (e < 2 * C ? inv_count0 : inv_count1) : 0.f;
    bn_param_grads(local, nseg, C, dbeta, dgamma);
}

// ---- 3x3 stride-2 pad-1 max-pool on NHWC bf16 --------------------------------------------------------------
// forward also records which tap (kh*3+kw, first maximum in row-major scan with strict >, as ATen's
// max_pool2d_with_indices) won, one byte per output element, so backward is a pure gather
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const unsigned short* __restrict__ X, int B, int Hi, int Wi, int C, int Ho,
                                                          int Wo, unsigned short* __restrict__ Y, unsigned char* __restrict__ IDX) {
    const int chunks = C >> 3;
    const size_t total = (size_t)B * Ho * Wo * chunks;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (size_t)gridDim.x * 256) {
        const int ch = (int)(q % chunks);
        size_t p = q / chunks;
        const int wo = (int)(p % Wo);
        p /= Wo;
        const int ho = (int)(p % Ho), b = (int)(p / Ho);
        float m[8];
        unsigned char arg[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            m[i] = -INFINITY;
            arg[i] = 0;
        }
        for (int kh = 0; kh < 3; ++kh) {
            const int hi = ho * 2 - 1 + kh;
            if (hi < 0 || hi >= Hi) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int wi = wo * 2 - 1 + kw;
                if (wi < 0 || wi >= Wi) continue;
                float x[8];
                unpack8(*reinterpret_cast<const u16x8*>(X + (((size_t)b * Hi + hi) * Wi + wi) * C + ch * 8), x);
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (x[i] > m[i]) {
                        m[i] = x[i];
                        arg[i] = (unsigned char)(kh * 3 + kw);
                    }
            }
        }
        *reinterpret_cast<u16x8*>(Y + q * 8) = pack8(m);
        uint2 packed;
        packed.x = arg[0] | (arg[1] << 8) | (arg[2] << 16) | ((unsigned)arg[3] << 24);
        packed.y = arg[4] | (arg[5] << 8) | (arg[6] << 16) | ((unsigned)arg[7] << 24);
        *reinterpret_cast<uint2*>(IDX + q * 8) = packed;
    }
}

// gradient reaching input pixel (b, hi, wi), channels ch*8..+7, from the (<= 4) windows whose recorded arg-max is this pixel
__device__ __forceinline__ void pool_gather(const unsigned char* __restrict__ IDX, const unsigned short* __restrict__ DY, int b, int hi,
                                            int wi, int ch, int C, int Ho, int Wo, float (&g)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) g[i] = 0.f;
    const int ho_lo = hi / 2, ho_hi = min(Ho - 1, (hi + 1) / 2);
    const int wo_lo = wi / 2, wo_hi = min(Wo - 1, (wi + 1) / 2);
    for (int ho = ho_lo; ho <= ho_hi; ++ho)
        for (int wo = wo_lo; wo <= wo_hi; ++wo) {
            const unsigned mine = (unsigned)((hi - (ho * 2 - 1)) * 3 + (wi - (wo * 2 - 1)));
            const size_t o = (((size_t)b * Ho + ho) * Wo + wo) * C + ch * 8;
            const uint2 packed = *reinterpret_cast<const uint2*>(IDX + o);
            float d[8];
            unpack8(*reinterpret_cast<const u16x8*>(DY + o), d);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned a = ((i < 4 ? packed.x : packed.y) >> (8 * (i & 3))) & 0xffu;
                if (a == mine) g[i] += d[i];
            }
        }
}

// gather form: each input pixel collects dy from the (<= 4) windows whose recorded arg-max is this pixel
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const unsigned char* __restrict__ IDX, const unsigned short* __restrict__ DY,
                                                          int B, int Hi, int Wi, int C, int Ho, int Wo,
                                                          unsigned short* __restrict__ DX) {
    const int chunks = C >> 3;
    const size_t total = (size_t)B * Hi * Wi * chunks;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (size_t)gridDim.x * 256) {
        const int ch = (int)(q % chunks);
        size_t p = q / chunks;
        const int wi = (int)(p % Wi);
        p /= Wi;
        const int hi = (int)(p % Hi), b = (int)(p / Hi);
        float g[8];
        pool_gather(IDX, DY, b, hi, wi, ch, C, Ho, Wo, g);
        *reinterpret_cast<u16x8*>(DX + q * 8) = pack8(g);
    }
}

// ---- stem: BatchNorm + ReLU + max-pool in one pass --------------------------------------------------------------------------
// The stem's activation (B x 192 x 192 x 64 at 384^2: 0.9 GB per 192 frames) is consumed by the max-pool only, so it is never
// written: the pooled value is max over the window of bf16(relu(bn(z))) - each tap rounded exactly as lp_bn_apply would have
// stored it, so values AND arg-max ties equal the unfused lp_bn_apply -> lp_maxpool_fwd - and the backward kernels rebuild the
// activation's gradient on the fly from the pooled gradient, the arg-max bytes and z (ReLU gate: o > 2^-134, see bn_apply_kernel).
// (Measured and dropped, round 6 - this kernel runs at 2.8 - 3.0 TB/s of its bytes and neither memory latency nor L2 locality is why: requesting all
// nine taps before the first is used (clamped addresses + validity flags instead of the border branches: 92 VGPRs) 137.4 -> 136.9 us at 64 frames,
// 291 -> 294 at 128; an XCD-aware row order, so that the input row two neighbouring output rows share is fetched into one L2, 139.0 -> 136.9 and
// 296 -> 303; the step unchanged both times (profiles/r06p_*, r06q_*, retired/r06_pool_fwd_*.patch).  What is left is its arithmetic: every input value
// is normalised, rounded and compared in each of the 2.25 windows it belongs to, ~650 VALU instructions per 16 output bytes.)
__global__ __launch_bounds__(256) void bn_relu_maxpool_fwd_kernel(const unsigned short* __restrict__ Z, const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, int B, int Hi, int Wi, int C, int Ho,
                                                                  int Wo, unsigned short* __restrict__ Y, unsigned char* __restrict__ IDX) {
    // Row walk: a workgroup takes whole output rows (b, ho); a thread keeps ONE channel chunk (chunks = C/8 divides 256) and strides over
    // the row's pixels - no per-element index arithmetic beyond one multiply-add (the flat-index form spent three 64-bit divisions per
    // 16 bytes: 5 - 8 % of these kernels' time).
    const int chunks = C >> 3, ppi = 256 / chunks;
    const int ch = threadIdx.x % chunks, pl = threadIdx.x / chunks;
    float mu[8], sc[8], be[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        mu[i] = mean[ch * 8 + i];
        sc[i] = invstd[ch * 8 + i] * gamma[ch * 8 + i];
        be[i] = beta[ch * 8 + i];
    }
    const int rows = B * Ho;
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const int b = row / Ho, ho = row - b * Ho;
        for (int wo = pl; wo < Wo; wo += ppi) {
            float m[8];
            unsigned char arg[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                m[i] = -INFINITY;
                arg[i] = 0;
            }
            for (int kh = 0; kh < 3; ++kh) {
                const int hi = ho * 2 - 1 + kh;
                if (hi < 0 || hi >= Hi) continue;
                for (int kw = 0; kw < 3; ++kw) {
                    const int wi = wo * 2 - 1 + kw;
                    if (wi < 0 || wi >= Wi) continue;
                    float x[8];
                    unpack8(*reinterpret_cast<const u16x8*>(Z + (((size_t)b * Hi + hi) * Wi + wi) * C + ch * 8), x);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float a = bf16_to_f32(f32_to_bf16(fmaxf(fmaf(x[i] - mu[i], sc[i], be[i]), 0.f)));
                        if (a > m[i]) {
                            m[i] = a;
                            arg[i] = (unsigned char)(kh * 3 + kw);
                        }
                    }
                }
            }
            const size_t q = ((size_t)row * Wo + wo) * chunks + ch;
            *reinterpret_cast<u16x8*>(Y + q * 8) = pack8(m);
            uint2 packed;
            packed.x = arg[0] | (arg[1] << 8) | (arg[2] << 16) | ((unsigned)arg[3] << 24);
            packed.y = arg[4] | (arg[5] << 8) | (arg[6] << 16) | ((unsigned)arg[7] << 24);
            *reinterpret_cast<uint2*>(IDX + q * 8) = packed;
        }
    }
}

// [sum g, sum g * xhat] with g = relu-gated gradient of the (never stored) stem activation, gathered from the pooled gradient.
// chunks = C/8 must divide 256: thread -> (chunk tid % chunks, row lane tid / chunks), as colreduce_kernel.
__global__ __launch_bounds__(256) void bn_pool_bwd_reduce_kernel(const unsigned char* __restrict__ IDX, const unsigned short* __restrict__ DY,
                                                                 const unsigned short* __restrict__ Z, const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, int B, int Hi, int Wi, int C, int Ho, int Wo,
                                                                 lp_fxsum* __restrict__ sums) {
    __shared__ float red[2][256][8];
    const int chunks = C >> 3, lanes_r = 256 / chunks;
    const int ch = threadIdx.x % chunks, rl = threadIdx.x / chunks;
    float mu[8], is[8], sc[8], be[8], s0[8], s1[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        mu[i] = mean[ch * 8 + i];
        is[i] = invstd[ch * 8 + i];
        sc[i] = is[i] * gamma[ch * 8 + i];
        be[i] = beta[ch * 8 + i];
        s0[i] = s1[i] = 0.f;
    }
    const int rows = B * Hi;   // row walk as in bn_relu_maxpool_fwd_kernel: (b, hi) per workgroup step, pixels of the row over the row lanes
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const int b = row / Hi, hi = row - b * Hi;
        for (int wi = rl; wi < Wi; wi += lanes_r) {
            const size_t p = (size_t)row * Wi + wi;
            float z[8], g[8];
            unpack8(load_stream8(Z + p * C + ch * 8), z);
            pool_gather(IDX, DY, b, hi, wi, ch, C, Ho, Wo, g);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float zc = z[i] - mu[i];
                if (!(fmaf(zc, sc[i], be[i]) > 0x1p-134f)) g[i] = 0.f;
                s0[i] += g[i];
                s1[i] = fmaf(g[i], zc * is[i], s1[i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        red[0][threadIdx.x][i] = s0[i];
        red[1][threadIdx.x][i] = s1[i];
    }
    __syncthreads();
    for (int pair = threadIdx.x; pair < chunks * 8; pair += 256) {
        const int pc = pair >> 3, pi = pair & 7;
        float t0 = 0.f, t1 = 0.f;
        for (int q = 0; q < lanes_r; ++q) {
            t0 += red[0][q * chunks + pc][pi];
            t1 += red[1][q * chunks + pc][pi];
        }
        const int cc = pc * 8 + pi;
        fx_add(&sums[cc], t0);
        fx_add(&sums[C + cc], t1);
    }
}

// dz = gamma * invstd * (g - sum(g)/N - xhat * sum(g * xhat)/N), g as above
__global__ __launch_bounds__(256) void bn_pool_bwd_apply_kernel(const unsigned char* __restrict__ IDX, const unsigned short* __restrict__ DY,
                                                                const unsigned short* __restrict__ Z, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, const lp_fxsum* __restrict__ sums,
                                                                float inv_count, int B, int Hi, int Wi, int C, int Ho, int Wo,
                                                                unsigned short* __restrict__ DX, const lp_fxsum* __restrict__ local,
                                                                float* __restrict__ dbeta, float* __restrict__ dgamma) {
    bn_param_grads(local, 1, C, dbeta, dgamma);
    const int chunks = C >> 3, ppi = 256 / chunks;   // row walk as in bn_relu_maxpool_fwd_kernel
    const int ch = threadIdx.x % chunks, pl = threadIdx.x / chunks;
    float mu[8], is[8], sc[8], be[8], ga[8], k0[8], k1[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = ch * 8 + i;
        mu[i] = mean[c];
        is[i] = invstd[c];
        ga[i] = gamma[c] * is[i];
        sc[i] = ga[i];
        be[i] = beta[c];
        k0[i] = sums != nullptr ? fx_value(&sums[c]) * inv_count : 0.f;
        k1[i] = sums != nullptr ? fx_value(&sums[C + c]) * inv_count : 0.f;
    }
    const int rows = B * Hi;
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const int b = row / Hi, hi = row - b * Hi;
        for (int wi = pl; wi < Wi; wi += ppi) {
            const size_t q = ((size_t)row * Wi + wi) * chunks + ch;
            float z[8], g[8], o[8];
            unpack8(load_stream8(Z + q * 8), z);
            pool_gather(IDX, DY, b, hi, wi, ch, C, Ho, Wo, g);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float zc = z[i] - mu[i];
                if (!(fmaf(zc, sc[i], be[i]) > 0x1p-134f)) g[i] = 0.f;
                o[i] = ga[i] * (g[i] - k0[i] - zc * is[i] * k1[i]);
            }
            *reinterpret_cast<u16x8*>(DX + q * 8) = pack8(o);
        }
    }
}

// ---- round 3: the two backward passes with the pooled gradient and the arg-max bytes staged in LDS ---------------------------------
// bn_pool_bwd_reduce_kernel / _apply_kernel gather dy and the arg-max byte of up to 4 windows per input pixel straight from global memory:
// nine dependent-address loads per 16 B of z, 2.0 / 3.1 TB/s at any grid size (profiles/archive/r03ab_pool.txt).  Here a workgroup walks a band of
// output rows of one image; per output row ho it handles the input rows 2 ho and 2 ho + 1, whose windows lie in the output rows ho and ho + 1:
// those two rows of dy (96 x 128 B) and of arg-max bytes (96 x 64 B) sit in LDS (row ho + 1 is loaded while row ho is still there: each is
// read from memory once per band), and the gather reads LDS.  z is fetched LP_POOL_ZU = 2 chunks ahead per thread: with 6 in flight the
// kernel needed 195 VGPRs (2 waves per SIMD) and ran at 2.3 / 3.4 TB/s, with 2 it needs 115 and runs at 3.2 / 4.6 (profiles/archive/r03ag_pool_zu.txt:
// reduce 418 -> 260 us, apply 452 -> 314 us per 128 frames against the gather-from-memory kernels).  C = 64 only
// (the stem); other shapes keep the kernels above.  APPLY = false: the two reductions; true: dz.
constexpr int kPbW = 96;   // widest pooled row staged (Wo <= 96: 384-px frames)
#ifndef LP_POOL_ZU
#define LP_POOL_ZU 2
#endif
template <bool APPLY>
__global__ __launch_bounds__(256) void bn_pool_bwd_v2_kernel(const unsigned char* __restrict__ IDX, const unsigned short* __restrict__ DY,
                                                             const unsigned short* __restrict__ Z, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const lp_fxsum* __restrict__ sums_in, float inv_count,
                                                             int B, int Hi, int Wi, int Ho, int Wo, int band, lp_fxsum* __restrict__ sums,
                                                             unsigned short* __restrict__ DX, const lp_fxsum* __restrict__ local,
                                                             float* __restrict__ dbeta, float* __restrict__ dgamma) {
    constexpr int C = 64, chunks = 8;
    if (APPLY) bn_param_grads(local, 1, C, dbeta, dgamma);
    constexpr int kZU = LP_POOL_ZU;   // z chunks in flight per thread
    __shared__ __attribute__((aligned(16))) unsigned short sdy[2][kPbW * C];
    __shared__ __attribute__((aligned(16))) unsigned char sidx[2][kPbW * C];
    __shared__ float red[APPLY ? 1 : 2][APPLY ? 1 : 256][8];
    const int ch = threadIdx.x % chunks, pl = threadIdx.x / chunks;   // 32 pixel lanes
    float mu[8], is[8], sc[8], be[8], k0[8], k1[8], s0[8], s1[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = ch * 8 + i;
        mu[i] = mean[c];
        is[i] = invstd[c];
        sc[i] = is[i] * gamma[c];
        be[i] = beta[c];
        k0[i] = (APPLY && sums_in != nullptr) ? fx_value(&sums_in[c]) * inv_count : 0.f;
        k1[i] = (APPLY && sums_in != nullptr) ? fx_value(&sums_in[C + c]) * inv_count : 0.f;
        s0[i] = s1[i] = 0.f;
    }
    const int bands = (Ho + band - 1) / band;
    auto stage_row = [&](int b, int ho, int slot) {   // pooled row ho of image b -> LDS slot (rows past the image: never read)
        if (ho >= Ho) return;
        const size_t base = ((size_t)b * Ho + ho) * Wo * C;
        for (int q = threadIdx.x; q < Wo * chunks; q += 256) {
            *reinterpret_cast<u16x8*>(&sdy[slot][q * 8]) = *reinterpret_cast<const u16x8*>(DY + base + (size_t)q * 8);
            *reinterpret_cast<uint2*>(&sidx[slot][q * 8]) = *reinterpret_cast<const uint2*>(IDX + base + (size_t)q * 8);
        }
    };
    for (int wgi = blockIdx.x; wgi < B * bands; wgi += gridDim.x) {
        const int b = wgi / bands, ho0 = (wgi - b * bands) * band;
        const int ho1 = ho0 + band < Ho ? ho0 + band : Ho;
        __syncthreads();   // the previous band's rows are dead
        stage_row(b, ho0, ho0 & 1);
        for (int ho = ho0; ho < ho1; ++ho) {
            stage_row(b, ho + 1, (ho + 1) & 1);
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int hi = 2 * ho + r;
                if (hi >= Hi) break;
                const size_t rowbase = ((size_t)b * Hi + hi) * Wi;
                for (int wi0 = 0; wi0 < Wi; wi0 += 32 * kZU) {
                    u16x8 zv[kZU];
#pragma unroll
                    for (int k = 0; k < kZU; ++k) {
                        const int wi = wi0 + pl + 32 * k;
                        if (wi < Wi) zv[k] = load_stream8(Z + (rowbase + wi) * C + ch * 8);
                    }
#pragma unroll
                    for (int k = 0; k < kZU; ++k) {
                        const int wi = wi0 + pl + 32 * k;
                        if (wi >= Wi) break;
                        float z[8], g[8];
                        unpack8(zv[k], z);
#pragma unroll
                        for (int i = 0; i < 8; ++i) g[i] = 0.f;
                        // the (<= 4) windows that contain this pixel (a loop over the valid ones: reading all four unconditionally was slower,
                        // ~350 instructions per 16 B of z, profiles/archive/r03ac_pool_v2.txt)
                        const int hlo = ho, hhi = (r == 1 && ho + 1 < Ho) ? ho + 1 : ho;   // rows hi / 2 .. min(Ho - 1, (hi + 1) / 2)
                        const int wlo = wi >> 1, whi = ((wi + 1) >> 1) < Wo ? ((wi + 1) >> 1) : Wo - 1;
                        for (int hh = hlo; hh <= hhi; ++hh)
                            for (int ww = wlo; ww <= whi; ++ww) {
                                const unsigned mine = (unsigned)((hi - (hh * 2 - 1)) * 3 + (wi - (ww * 2 - 1)));
                                const int o = (ww * chunks + ch) * 8;
                                const uint2 packed = *reinterpret_cast<const uint2*>(&sidx[hh & 1][o]);
                                float d[8];
                                unpack8(*reinterpret_cast<const u16x8*>(&sdy[hh & 1][o]), d);
#pragma unroll
                                for (int i = 0; i < 8; ++i) {
                                    const unsigned a = ((i < 4 ? packed.x : packed.y) >> (8 * (i & 3))) & 0xffu;
                                    if (a == mine) g[i] += d[i];
                                }
                            }
                        if (APPLY) {
                            float o8[8];
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                const float zc = z[i] - mu[i];
                                if (!(fmaf(zc, sc[i], be[i]) > 0x1p-134f)) g[i] = 0.f;
                                o8[i] = sc[i] * (g[i] - k0[i] - zc * is[i] * k1[i]);
                            }
                            *reinterpret_cast<u16x8*>(DX + (rowbase + wi) * C + ch * 8) = pack8(o8);
                        } else {
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                const float zc = z[i] - mu[i];
                                if (!(fmaf(zc, sc[i], be[i]) > 0x1p-134f)) g[i] = 0.f;
                                s0[i] += g[i];
                                s1[i] = fmaf(g[i], zc * is[i], s1[i]);
                            }
                        }
                    }
                }
            }
            __syncthreads();   // row ho is dead: the next step's staging may overwrite its slot
        }
    }
    if (!APPLY) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            red[0][threadIdx.x][i] = s0[i];
            red[APPLY ? 0 : 1][threadIdx.x][i] = s1[i];
        }
        __syncthreads();
        if (threadIdx.x < C) {
            const int pc = threadIdx.x >> 3, pi = threadIdx.x & 7;
            float t0 = 0.f, t1 = 0.f;
            for (int q = 0; q < 32; ++q) {
                t0 += red[0][q * chunks + pc][pi];
                t1 += red[APPLY ? 0 : 1][q * chunks + pc][pi];
            }
            fx_add(&sums[threadIdx.x], t0);
            fx_add(&sums[C + threadIdx.x], t1);
        }
    }
}

// output rows per workgroup: 6 -> 16 bands per 96-row map, 1024 / 2048 workgroups for 64 / 128 frames = whole rounds (3, 4, 8, 12 measured:
// profiles/archive/r03ad_pool_band.txt)
static int pool_v2_band() { return 6; }

static bool pool_v2_ok(int C, int Hi, int Wi, int Ho, int Wo) {
    if (lp_switches().pool_v2 == 0) return false;
    return C == 64 && Wo <= kPbW && Hi <= 2 * Ho && Wi <= 2 * Wo;
}

// ---- images (B,3,H,W) fp32 NCHW -> (B,H,W,4) bf16, channel 3 = 0 ----------------------------------------------
__global__ __launch_bounds__(256) void nchw3_to_nhwc4_kernel(const float* __restrict__ X, int B, int HW, unsigned short* __restrict__ Y) {
    const size_t total = (size_t)B * HW;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (size_t)gridDim.x * 256) {
        const size_t b = q / HW, p = q % HW;
        const float* src = X + b * 3 * HW + p;
        u16x4 v = {f32_to_bf16(src[0]), f32_to_bf16(src[HW]), f32_to_bf16(src[2 * (size_t)HW]), 0};
        *reinterpret_cast<u16x4*>(Y + q * 4) = v;
    }
}

// ---- PixelShuffle(2) on NHWC: out[b][2y+i][2x+j][c] = in[b][y][x][4c + 2i + j] ;  inverse for the gradient ------------
// the high-resolution tensor has channel pitch `ld` >= Cout (pad channels are never touched)
template <bool INVERSE>
__global__ __launch_bounds__(256) void pixel_shuffle_kernel(const unsigned short* __restrict__ IN, int B, int h, int w, int Cout, int ld,
                                                            unsigned short* __restrict__ OUT) {
    // thread = one low-res pixel x 8 consecutive LOW-res channels (16 B) ; they map to 2 output channels x 4 positions
    const int cin = Cout * 4, chunks = cin >> 3;
    const size_t total = (size_t)B * h * w * chunks;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (size_t)gridDim.x * 256) {
        const int ch = (int)(q % chunks);
        size_t p = q / chunks;
        const int x = (int)(p % w);
        p /= w;
        const int y = (int)(p % h), b = (int)(p / h);
        u16x8 v;
        if (!INVERSE) v = *reinterpret_cast<const u16x8*>(IN + q * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int cl = ch * 8 + e;           // low-res channel = 4c + 2i + j
            const int c = cl >> 2, i = (cl >> 1) & 1, j = cl & 1;
            const size_t o = ((((size_t)b * 2 * h + 2 * y + i) * 2 * w) + 2 * x + j) * ld + c;
            if (INVERSE) v[e] = IN[o];
            else OUT[o] = v[e];
        }
        if (INVERSE) *reinterpret_cast<u16x8*>(OUT + q * 8) = v;
    }
}

// workgroups for a column reduction: enough to fill the chip (4 per CU), each owning >= 8 row passes
static int colreduce_blocks(int M, int C) {
    const int chunks = C / 8;
    const int lanes_r = chunks < 256 ? 256 / chunks : 1;
    long passes = ((long)M + lanes_r - 1) / lanes_r;
    long blocks = (passes + 7) / 8;
    if (blocks > 1024) blocks = 1024;
    return blocks < 1 ? 1 : (int)blocks;
}

// grid for the BatchNorm elementwise walks: the stride (grid * 256 lanes) must be a multiple of the row length in chunks so a
// lane stays on one channel chunk; rows of C/8 chunks with C/8 a divisor of 256 (every ResNet width) need no adjustment
static int bn_grid(size_t n_chunks, int chunks, int per_cu = 5) {
    size_t blocks = (n_chunks + 255) / 256;
    // ONE resident round: bn_apply_kernel / bn_bwd_apply_kernel run 5 waves per SIMD = 5 workgroups per CU.  (Until round 4 the cap was
    // 2048 = 1.6 rounds, the second one 60 % full: 1280 gave bn_apply 110.7 -> 102.2 us and bn_bwd_apply 129.0 -> 121.8 us per launch,
    // 4260 -> 4324 frames/s over three A/B pairs; 1024 - four per CU - loses it again, 1536 gives half of it.  profiles/r04p_*, r04q_*)
    if (blocks > (size_t)256 * per_cu) blocks = (size_t)256 * per_cu;   // (per_cu < 5: LP_BN_BWD_WGS_PER_CU, A/B of the backward walk's share of a CU beside the weight-gradient stream)
    if (blocks < 1) blocks = 1;
    if (256 % chunks != 0) {  // make grid * 256 a multiple of `chunks`: round the grid up to a multiple of chunks / gcd(chunks, 256)
        int a = chunks, b = 256;
        while (b) {
            const int t = a % b;
            a = b;
            b = t;
        }
        const size_t m = (size_t)(chunks / a);
        blocks = (blocks + m - 1) / m * m;
    }
    return (int)blocks;
}

// workgroups for the stem's row-walk kernels: one row per step, grid-stride beyond 16 workgroups per CU (4 for the reduction, whose
// workgroups each end in 2 C atomics)
static int pool_row_blocks(int rows, int per_cu = 16) {
    const int cap = 256 * per_cu;
    return rows < 1 ? 1 : (rows > cap ? cap : rows);
}

static int grid_for(size_t work_items) {
    size_t blocks = (work_items + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;  // grid-stride beyond 16 workgroups per CU
    return blocks < 1 ? 1 : (int)blocks;
}

// ---- inference: fold a BatchNorm's running statistics into the preceding convolution -------------------------------------
// y = gamma (conv(x, W) - mean) / sqrt(var + eps) + beta = conv(x, a W) + (beta - a mean),  a = gamma / sqrt(var + eps) per output
// channel.  One workgroup per output channel; w rows are [Co][per_co] fp32 masters, the folded copy is the bf16 GEMM operand.
__global__ __launch_bounds__(256) void bn_fold_kernel(const float* __restrict__ w, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ rmean,
                                                      const float* __restrict__ rvar, float eps, int per_co,
                                                      unsigned short* __restrict__ w_out, float* __restrict__ bias_out) {
    const int co = blockIdx.x;
    const float a = gamma[co] / sqrtf(rvar[co] + eps);  // IEEE divide + sqrt (what F.batch_norm(training=False) folds to), not the 1-ulp rsqrtf
    const float* src = w + (size_t)co * per_co;
    unsigned short* dst = w_out + (size_t)co * per_co;
    for (int j = threadIdx.x; j < per_co; j += 256) dst[j] = f32_to_bf16(src[j] * a);
    if (threadIdx.x == 0) bias_out[co] = beta[co] - rmean[co] * a;
}

}  // namespace lp

// workspace of the two stand-alone reductions below: one [2][C] fp32 row of partial sums per workgroup
extern "C" size_t lp_bn_reduce_workspace_bytes(int M, int C) {
    if (M <= 0 || C <= 0) return 0;
    return (size_t)lp::colreduce_blocks(M, C) * 2 * C * sizeof(float);
}

extern "C" int lp_bn_stats(const void* x, int M, int C, lp_fxsum* sums, void* workspace, size_t workspace_bytes, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(x && sums && workspace && M > 0 && C > 0);
    if (C % 8 != 0) return LP_ERR_UNSUPPORTED;
    dim3 grid(colreduce_blocks(M, C), 1);
    LP_REQUIRE(workspace_bytes >= (size_t)grid.x * 2 * C * sizeof(float));
    hipLaunchKernelGGL((colreduce_kernel<0>), grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x,
                       (const unsigned short*)nullptr, (const unsigned short*)nullptr, (const float*)nullptr, (const float*)nullptr, M,
                       C, (float*)workspace);
    hipLaunchKernelGGL(rows_reduce_kernel, dim3((C + 15) / 16), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, (int)grid.x, C, sums);
    return launch_status();
}

// dst[i] += the value of sums[i] (a fixed-point total as fp32): what a caller that wants plain floats does with lp_bn_stats' output
// (the head's bias gradients are column sums of the output gradient)
__global__ void fx_accumulate_kernel(const lp_fxsum* __restrict__ sums, int n, float* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += lp::fx_value(&sums[i]);
}

extern "C" int lp_fxsum_accumulate(const lp_fxsum* sums, int n, float* dst, lp_stream_t stream) {
    LP_REQUIRE(sums && dst && n > 0);
    hipLaunchKernelGGL(fx_accumulate_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, n, dst);
    return lp::launch_status();
}

extern "C" int lp_bn_fold(const float* w, const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                          float eps, int Co, int per_co, void* w_bf16, float* bias, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(w && gamma && beta && running_mean && running_var && w_bf16 && bias && Co > 0 && per_co > 0 && eps > 0.f);
    hipLaunchKernelGGL(bn_fold_kernel, dim3(Co), dim3(256), 0, (hipStream_t)stream, w, gamma, beta, running_mean, running_var, eps, per_co,
                       (unsigned short*)w_bf16, bias);
    return launch_status();
}

extern "C" int lp_bn_finalize(const lp_fxsum* sums, float count, int C, float eps, float momentum, float* mean, float* invstd,
                              float* running_mean, float* running_var, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(sums && mean && invstd && C > 0 && count > 0.f);
    hipLaunchKernelGGL(bn_finalize_kernel<lp_fxsum>, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, count, count, 1, C, eps,
                       momentum, mean, invstd, running_mean, running_var);
    return launch_status();
}

extern "C" int lp_bn_finalize2(const lp_fxsum* sums, float count0, float count1, int C, float eps, float momentum, float* mean, float* invstd,
                               float* running_mean, float* running_var, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(sums && mean && invstd && C > 0 && count0 > 0.f && count1 > 0.f);
    hipLaunchKernelGGL(bn_finalize_kernel<lp_fxsum>, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, count0, count1, 2, C, eps,
                       momentum, mean, invstd, running_mean, running_var);
    return launch_status();
}

// the same from plain fp32 [sum, sum of squares] (the fp32 validation executor's statistics: lp_f32_bn_stats_ordered)
extern "C" int lp_bn_finalize_f32(const float* sums, float count, int C, float eps, float momentum, float* mean, float* invstd,
                                  float* running_mean, float* running_var, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(sums && mean && invstd && C > 0 && count > 0.f);
    hipLaunchKernelGGL(bn_finalize_kernel<float>, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, count, count, 1, C, eps,
                       momentum, mean, invstd, running_mean, running_var);
    return launch_status();
}


static int bn_apply_impl(const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta, const void* residual,
                         int relu, int M, int C, int seg_rows, void* y, void* relu_bits, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(x && mean && invstd && gamma && beta && y && M > 0 && C > 0 && seg_rows >= 0 && seg_rows < M);
    if (C % 8 != 0) return LP_ERR_UNSUPPORTED;
    const size_t n_chunks = (size_t)M * (C / 8);
    hipLaunchKernelGGL(bn_apply_kernel<false>, dim3(bn_grid(n_chunks, C / 8)), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, mean, invstd,
                       gamma, beta, (const unsigned short*)residual, relu, n_chunks, C, (unsigned short*)y, (unsigned char*)relu_bits,
                       (size_t)seg_rows * (C / 8), BnResidualBn{});
    return launch_status();
}

// y = relu?( BatchNorm(x) + bf16( BatchNorm_d(zd) ) ): the block output of a layer's first block with its projection shortcut normalised in the
// same pass (mean_d / invstd_d: (segments, C) like mean / invstd)
extern "C" int lp_bn_apply_seg_rbn(const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta, const void* zd,
                                   const float* mean_d, const float* invstd_d, const float* gamma_d, const float* beta_d, int relu, int M, int C,
                                   int seg_rows, void* y, void* relu_bits, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(x && mean && invstd && gamma && beta && zd && mean_d && invstd_d && gamma_d && beta_d && y && M > 0 && C > 0 && seg_rows >= 0 &&
               seg_rows < M);
    if (C % 8 != 0) return LP_ERR_UNSUPPORTED;
    const size_t n_chunks = (size_t)M * (C / 8);
    hipLaunchKernelGGL(bn_apply_kernel<true>, dim3(bn_grid(n_chunks, C / 8)), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, mean, invstd,
                       gamma, beta, (const unsigned short*)zd, relu, n_chunks, C, (unsigned short*)y, (unsigned char*)relu_bits,
                       (size_t)seg_rows * (C / 8), BnResidualBn{mean_d, invstd_d, gamma_d, beta_d});
    return launch_status();
}

// The block-output pass of the "fp32 residual stream" policy (bn_apply_kernel<.., LO>): residual = hi [+ residual_lo], or - zd given - the
// projection shortcut zd normalised here with (mean_d, invstd_d, gamma_d, beta_d) and added UNROUNDED; y = bf16 of the result, y_lo (optional)
// = bf16(result - y).  Everything else as lp_bn_apply_seg / lp_bn_apply_seg_rbn.
extern "C" int lp_bn_apply_seg_lo(const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta, const void* residual,
                                  const void* residual_lo, const void* zd, const float* mean_d, const float* invstd_d, const float* gamma_d,
                                  const float* beta_d, int relu, int M, int C, int seg_rows, void* y, void* y_lo, void* relu_bits,
                                  lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(x && mean && invstd && gamma && beta && y && M > 0 && C > 0 && seg_rows >= 0 && seg_rows < M);
    LP_REQUIRE((zd == nullptr) || (mean_d && invstd_d && gamma_d && beta_d && residual == nullptr && residual_lo == nullptr));
    LP_REQUIRE(residual_lo == nullptr || residual != nullptr);
    if (C % 8 != 0) return LP_ERR_UNSUPPORTED;
    const size_t n_chunks = (size_t)M * (C / 8);
    const BnResidualLo rl{(const unsigned short*)residual_lo, (unsigned short*)y_lo};
    if (zd != nullptr)
        hipLaunchKernelGGL((bn_apply_kernel<true, true>), dim3(bn_grid(n_chunks, C / 8)), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, mean,
                           invstd, gamma, beta, (const unsigned short*)zd, relu, n_chunks, C, (unsigned short*)y, (unsigned char*)relu_bits,
                           (size_t)seg_rows * (C / 8), BnResidualBn{mean_d, invstd_d, gamma_d, beta_d}, rl);
    else
        hipLaunchKernelGGL((bn_apply_kernel<false, true>), dim3(bn_grid(n_chunks, C / 8)), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, mean,
                           invstd, gamma, beta, (const unsigned short*)residual, relu, n_chunks, C, (unsigned short*)y, (unsigned char*)relu_bits,
                           (size_t)seg_rows * (C / 8), BnResidualBn{}, rl);
    return launch_status();
}

extern "C" int lp_bn_apply(const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                           const void* residual, int relu, int M, int C, void* y, void* relu_bits, lp_stream_t stream) {
    return bn_apply_impl(x, mean, invstd, gamma, beta, residual, relu, M, C, 0, y, relu_bits, stream);
}

// two BatchNorm segments in one launch: rows [0, seg_rows) are normalised with mean / invstd row 0, the rest with row 1 ((2, C) each)
extern "C" int lp_bn_apply_seg(const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                               const void* residual, int relu, int M, int C, int seg_rows, void* y, void* relu_bits, lp_stream_t stream) {
    return bn_apply_impl(x, mean, invstd, gamma, beta, residual, relu, M, C, seg_rows, y, relu_bits, stream);
}

extern "C" int lp_bn_bwd_reduce(const void* dy, const void* y_out, const void* x, const float* mean, const float* invstd, int M, int C,
                                lp_fxsum* sums, void* workspace, size_t workspace_bytes, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(dy && x && mean && invstd && sums && workspace && M > 0 && C > 0);
    if (C % 8 != 0) return LP_ERR_UNSUPPORTED;
    dim3 grid(colreduce_blocks(M, C), 1);
    LP_REQUIRE(workspace_bytes >= (size_t)grid.x * 2 * C * sizeof(float));
    hipLaunchKernelGGL((colreduce_kernel<1>), grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short*)dy,
                       (const unsigned short*)y_out, (const unsigned short*)x, mean, invstd, M, C, (float*)workspace);
    hipLaunchKernelGGL(rows_reduce_kernel, dim3((C + 15) / 16), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, (int)grid.x, C, sums);
    return launch_status();
}

static int bn_bwd_apply_impl(const void* dy, const void* y_out, const void* x, const float* mean, const float* invstd, const float* gamma,
                             const lp_fxsum* sums, float count0, float count1, int M, int C, int seg_rows, void* dx, void* dres,
                             const lp_fxsum* sums_local, float* dbeta_acc, float* dgamma_acc, float* terms_ws, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(dy && x && mean && invstd && gamma && dx && M > 0 && C > 0 && count0 > 0.f && count1 > 0.f && seg_rows >= 0 && seg_rows < M);
    LP_REQUIRE(sums_local != nullptr || (dbeta_acc == nullptr && dgamma_acc == nullptr));
    if (C % 8 != 0 || (terms_ws == nullptr && C > kBnBwdMaxC)) return LP_ERR_UNSUPPORTED;
    const size_t n_chunks = (size_t)M * (C / 8);
    if (terms_ws != nullptr) {   // two launches: the conversion + parameter gradients (one thread per value), then the streaming kernel
        const int nseg = seg_rows > 0 ? 2 : 1;
        hipLaunchKernelGGL(bn_bwd_terms_kernel, dim3((nseg * 2 * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, 1.f / count0, 1.f / count1, C,
                           nseg, terms_ws, sums_local, dbeta_acc, dgamma_acc);
        hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3(bn_grid(n_chunks, C / 8, lp_switches().bn_bwd_wgs_per_cu)), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)dy,
                           (const unsigned short*)y_out, (const unsigned short*)x, mean, invstd, gamma, sums, 1.f / count0, n_chunks, C,
                           (unsigned short*)dx, (unsigned short*)dres, (size_t)seg_rows * (C / 8), 1.f / count1, sums_local, dbeta_acc, dgamma_acc,
                           (const float*)terms_ws);
        return launch_status();
    }
    hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(bn_grid(n_chunks, C / 8)), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)dy,
                       (const unsigned short*)y_out, (const unsigned short*)x, mean, invstd, gamma, sums, 1.f / count0, n_chunks, C,
                       (unsigned short*)dx, (unsigned short*)dres, (size_t)seg_rows * (C / 8), 1.f / count1, sums_local, dbeta_acc, dgamma_acc,
                       (const float*)nullptr);
    return launch_status();
}

extern "C" int lp_bn_bwd_apply(const void* dy, const void* y_out, const void* x, const float* mean, const float* invstd,
                               const float* gamma, const lp_fxsum* sums, float count, int M, int C, void* dx, void* dres,
                               const lp_fxsum* sums_local, float* dbeta_acc, float* dgamma_acc, float* terms_ws, lp_stream_t stream) {
    return bn_bwd_apply_impl(dy, y_out, x, mean, invstd, gamma, sums, count, count, M, C, 0, dx, dres, sums_local, dbeta_acc, dgamma_acc, terms_ws,
                             stream);
}

// two segments: mean / invstd (2, C), sums (2, 2, C), one row count per segment (times the world size under SyncBatchNorm)
extern "C" int lp_bn_bwd_apply_seg(const void* dy, const void* y_out, const void* x, const float* mean, const float* invstd,
                                   const float* gamma, const lp_fxsum* sums, float count0, float count1, int M, int C, int seg_rows, void* dx,
                                   void* dres, const lp_fxsum* sums_local, float* dbeta_acc, float* dgamma_acc, float* terms_ws,
                                   lp_stream_t stream) {
    return bn_bwd_apply_impl(dy, y_out, x, mean, invstd, gamma, sums, count0, count1, M, C, seg_rows, dx, dres, sums_local, dbeta_acc,
                             dgamma_acc, terms_ws, stream);
}

// lp_bn_bwd_apply_seg that ALSO leaves the two backward reductions of a second BatchNorm fed by the same masked gradient (a block's projection
// shortcut: zd, mean_d / invstd_d [segments][C]) in sums_d ([segments][2][C], added into: zero them first) - what lp_bn_bwd_reduce(dres, NULL, zd,
// ...) per segment would compute from one more pass over the gradient.  Needs the terms workspace and C / 8 dividing 256.
extern "C" size_t lp_bn_bwd_ds_workspace_bytes(int M, int C) {
    if (M <= 0 || C <= 0 || C % 8 != 0) return 0;
    return (size_t)2 * lp::bn_grid((size_t)M * (C / 8), C / 8, lp::lp_switches().bn_bwd_wgs_per_cu) * 2 * C * sizeof(float);
}

extern "C" int lp_bn_bwd_apply_seg_ds(const void* dy, const void* y_out, const void* x, const float* mean, const float* invstd,
                                      const float* gamma, const lp_fxsum* sums, float count0, float count1, int M, int C, int seg_rows,
                                      void* dx, void* dres, const lp_fxsum* sums_local, float* dbeta_acc, float* dgamma_acc, float* terms_ws,
                                      const void* zd, const float* mean_d, const float* invstd_d, lp_fxsum* sums_d, void* workspace,
                                      size_t workspace_bytes, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(dy && x && mean && invstd && gamma && dx && M > 0 && C > 0 && count0 > 0.f && count1 > 0.f && seg_rows >= 0 && seg_rows < M);
    LP_REQUIRE(sums_local != nullptr || (dbeta_acc == nullptr && dgamma_acc == nullptr));
    LP_REQUIRE(terms_ws && zd && mean_d && invstd_d && sums_d && workspace);
    if (C % 8 != 0 || 256 % (C / 8) != 0) return LP_ERR_UNSUPPORTED;
    const size_t n_chunks = (size_t)M * (C / 8);
    const int nseg = seg_rows > 0 ? 2 : 1;
    const int grid = bn_grid(n_chunks, C / 8, lp_switches().bn_bwd_wgs_per_cu);
    LP_REQUIRE(workspace_bytes >= (size_t)nseg * grid * 2 * C * sizeof(float));
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_bwd_terms_kernel, dim3((nseg * 2 * C + 255) / 256), dim3(256), 0, st, sums, 1.f / count0, 1.f / count1, C, nseg, terms_ws,
                       sums_local, dbeta_acc, dgamma_acc);
    hipLaunchKernelGGL((bn_bwd_apply_kernel<true, true>), dim3(grid), dim3(256), 0, st, (const unsigned short*)dy, (const unsigned short*)y_out,
                       (const unsigned short*)x, mean, invstd, gamma, sums, 1.f / count0, n_chunks, C, (unsigned short*)dx, (unsigned short*)dres,
                       (size_t)seg_rows * (C / 8), 1.f / count1, sums_local, dbeta_acc, dgamma_acc, (const float*)terms_ws,
                       BnBwdDs{(const unsigned short*)zd, mean_d, invstd_d, (float*)workspace});
    for (int sg = 0; sg < nseg; ++sg)
        hipLaunchKernelGGL(rows_reduce_kernel, dim3((C + 15) / 16), dim3(256), 0, st, (const float*)workspace + (size_t)sg * grid * 2 * C, grid, C,
                           sums_d + (size_t)sg * 2 * C);
    return launch_status();
}

extern "C" int lp_maxpool_fwd(const void* x, int B, int Hi, int Wi, int C, void* y, void* argmax_u8, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(x && y && argmax_u8 && B > 0 && Hi > 0 && Wi > 0 && C > 0);
    if (C % 8 != 0) return LP_ERR_UNSUPPORTED;
    const int Ho = (Hi - 1) / 2 + 1, Wo = (Wi - 1) / 2 + 1;
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for((size_t)B * Ho * Wo * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)x, B, Hi, Wi, C, Ho, Wo, (unsigned short*)y, (unsigned char*)argmax_u8);
    return launch_status();
}

extern "C" int lp_maxpool_bwd(const void* argmax_u8, const void* dy, int B, int Hi, int Wi, int C, void* dx, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(argmax_u8 && dy && dx && B > 0 && Hi > 0 && Wi > 0 && C > 0);
    if (C % 8 != 0) return LP_ERR_UNSUPPORTED;
    const int Ho = (Hi - 1) / 2 + 1, Wo = (Wi - 1) / 2 + 1;
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for((size_t)B * Hi * Wi * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned char*)argmax_u8, (const unsigned short*)dy, B, Hi, Wi, C, Ho, Wo, (unsigned short*)dx);
    return launch_status();
}

extern "C" int lp_bn_relu_maxpool_fwd(const void* z, const float* mean, const float* invstd, const float* gamma, const float* beta, int B,
                                      int Hi, int Wi, int C, void* y, void* argmax_u8, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(z && mean && invstd && gamma && beta && y && argmax_u8 && B > 0 && Hi > 0 && Wi > 0 && C > 0);
    if (C % 8 != 0 || 256 % (C / 8) != 0 || (long long)B * Hi >= (1LL << 31)) return LP_ERR_UNSUPPORTED;
    const int Ho = (Hi - 1) / 2 + 1, Wo = (Wi - 1) / 2 + 1;
    hipLaunchKernelGGL(bn_relu_maxpool_fwd_kernel, dim3(pool_row_blocks(B * Ho)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)z, mean, invstd, gamma, beta, B, Hi, Wi, C, Ho, Wo, (unsigned short*)y,
                       (unsigned char*)argmax_u8);
    return launch_status();
}

extern "C" int lp_bn_pool_bwd_reduce(const void* argmax_u8, const void* dy, const void* z, const float* mean, const float* invstd,
                                     const float* gamma, const float* beta, int B, int Hi, int Wi, int C, lp_fxsum* sums, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(argmax_u8 && dy && z && mean && invstd && gamma && beta && sums && B > 0 && Hi > 0 && Wi > 0 && C > 0);
    if (C % 8 != 0 || 256 % (C / 8) != 0) return LP_ERR_UNSUPPORTED;
    const int Ho = (Hi - 1) / 2 + 1, Wo = (Wi - 1) / 2 + 1;
    const long long pixels = (long long)B * Hi * Wi;
    if (pixels >= (1LL << 31)) return LP_ERR_UNSUPPORTED;
    if (pool_v2_ok(C, Hi, Wi, Ho, Wo)) {
        const int band = pool_v2_band(), wgs = B * ((Ho + band - 1) / band);   // one band per workgroup (4 resident per CU)
        hipLaunchKernelGGL(bn_pool_bwd_v2_kernel<false>, dim3(wgs), dim3(256), 0, (hipStream_t)stream,
                           (const unsigned char*)argmax_u8, (const unsigned short*)dy, (const unsigned short*)z, mean, invstd, gamma, beta, nullptr,
                           0.f, B, Hi, Wi, Ho, Wo, band, sums, nullptr, nullptr, nullptr, nullptr);
        return launch_status();
    }
    hipLaunchKernelGGL(bn_pool_bwd_reduce_kernel, dim3(pool_row_blocks(B * Hi, 4)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned char*)argmax_u8, (const unsigned short*)dy, (const unsigned short*)z, mean, invstd, gamma, beta, B, Hi,
                       Wi, C, Ho, Wo, sums);
    return launch_status();
}

// `sums`: the totals the correction terms use (all-reduced under SyncBatchNorm), NULL = none (eval-mode BatchNorm); `sums_local` +
// dbeta_acc / dgamma_acc: this rank's sums are added into the parameter gradients (NULL: not)
extern "C" int lp_bn_pool_bwd_apply(const void* argmax_u8, const void* dy, const void* z, const float* mean, const float* invstd,
                                    const float* gamma, const float* beta, const lp_fxsum* sums, float count, int B, int Hi, int Wi, int C,
                                    void* dx, const lp_fxsum* sums_local, float* dbeta_acc, float* dgamma_acc, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(argmax_u8 && dy && z && mean && invstd && gamma && beta && dx && B > 0 && Hi > 0 && Wi > 0 && C > 0 && count > 0.f);
    LP_REQUIRE(sums_local != nullptr || (dbeta_acc == nullptr && dgamma_acc == nullptr));
    if (C % 8 != 0 || 256 % (C / 8) != 0 || (long long)B * Hi >= (1LL << 31)) return LP_ERR_UNSUPPORTED;
    const int Ho = (Hi - 1) / 2 + 1, Wo = (Wi - 1) / 2 + 1;
    if (pool_v2_ok(C, Hi, Wi, Ho, Wo)) {
        const int band = pool_v2_band(), wgs = B * ((Ho + band - 1) / band);
        hipLaunchKernelGGL(bn_pool_bwd_v2_kernel<true>, dim3(wgs), dim3(256), 0, (hipStream_t)stream,
                           (const unsigned char*)argmax_u8, (const unsigned short*)dy, (const unsigned short*)z, mean, invstd, gamma, beta, sums,
                           1.f / count, B, Hi, Wi, Ho, Wo, band, nullptr, (unsigned short*)dx, sums_local, dbeta_acc, dgamma_acc);
        return launch_status();
    }
    hipLaunchKernelGGL(bn_pool_bwd_apply_kernel, dim3(pool_row_blocks(B * Hi)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned char*)argmax_u8, (const unsigned short*)dy, (const unsigned short*)z, mean, invstd, gamma, beta, sums,
                       1.f / count, B, Hi, Wi, C, Ho, Wo, (unsigned short*)dx, sums_local, dbeta_acc, dgamma_acc);
    return launch_status();
}

extern "C" int lp_images_to_nhwc4(const float* images, int B, int H, int W, void* out, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(images && out && B > 0 && H > 0 && W > 0);
    hipLaunchKernelGGL(nchw3_to_nhwc4_kernel, dim3(grid_for((size_t)B * H * W)), dim3(256), 0, (hipStream_t)stream, images, B, H * W,
                       (unsigned short*)out);
    return launch_status();
}

extern "C" int lp_pixel_shuffle(const void* in, int B, int h, int w, int c_out, int ld, int inverse, void* out, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(in && out && B > 0 && h > 0 && w > 0 && c_out > 0 && ld >= c_out);
    if ((c_out * 4) % 8 != 0) return LP_ERR_UNSUPPORTED;
    const size_t work = (size_t)B * h * w * (c_out * 4 / 8);
    if (inverse)
        hipLaunchKernelGGL((pixel_shuffle_kernel<true>), dim3(grid_for(work)), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)in, B,
                           h, w, c_out, ld, (unsigned short*)out);
    else
        hipLaunchKernelGGL((pixel_shuffle_kernel<false>), dim3(grid_for(work)), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)in, B,
                           h, w, c_out, ld, (unsigned short*)out);
    return launch_status();
}
