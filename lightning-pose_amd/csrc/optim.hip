// Fused Adam / AdamW over a flat fp32 parameter range + the bf16 operand copies the MFMA kernels read.  gfx950.
//
// Replaces torch.optim.Adam / AdamW (foreach) built in lightning_pose/models/base.py:458-479 with the two parameter
// groups of models/heatmap_tracker.py:193-205.  The backbone group runs with lr = 0 until UnfreezeBackbone raises it
// (callbacks.py:79-196): moments are still updated, exactly as the reference does.  All parameters,
// gradients and moments live in ONE flat buffer each, so a group is one launch and gradients all-reduce as a few
// large RCCL buckets.
#include "lp_common.h"

namespace lp {

// torch single-tensor Adam: m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
//   p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps) ;  AdamW: p *= 1 - lr*wd first ; Adam: g += wd*p
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, size_t n, float lr, float beta1, float beta2, float eps,
                                                   float weight_decay, int decoupled, float bc1, float bc2_sqrt, float grad_scale,
                                                   unsigned short* __restrict__ p_bf16) {
    const float step_size = lr / bc1;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float pv = p[i];
        float gv = g[i] * grad_scale;
        if (weight_decay != 0.f) {
            if (decoupled) pv *= 1.f - lr * weight_decay;
            else gv = fmaf(weight_decay, pv, gv);
        }
        const float mv = beta1 * m[i] + (1.f - beta1) * gv;
        const float vv = beta2 * v[i] + (1.f - beta2) * gv * gv;
        m[i] = mv;
        v[i] = vv;
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        pv -= step_size * (mv / denom);
        p[i] = pv;
        if (p_bf16 != nullptr) p_bf16[i] = f32_to_bf16(pv);
    }
}


__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, size_t n, unsigned short* __restrict__ dst) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = f32_to_bf16(src[i]);
}

// dst[c][b][a] = src[a][b][c]   (conv weights [Co][R*S][Ci] -> data-gradient copy [Ci][R*S][Co])
__global__ __launch_bounds__(256) void permute_cba_kernel(const unsigned short* __restrict__ src, int A, int Bm, int Cn,
                                                          unsigned short* __restrict__ dst) {
    const size_t n = (size_t)A * Bm * Cn;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int a = (int)(i % A);
        const size_t t = i / A;
        const int b = (int)(t % Bm), c = (int)(t / Bm);
        dst[i] = src[((size_t)a * Bm + b) * Cn + c];
    }
}

static int grid_for_n(size_t n) {
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    return blocks < 1 ? 1 : (int)blocks;
}

}  // namespace lp

extern "C" int lp_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1,
                            float beta2, float eps, float weight_decay, int decoupled, int step, float grad_scale, void* params_bf16,
                            lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(params && grads && exp_avg && exp_avg_sq && step >= 1);
    if (n == 0) return LP_OK;
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for_n(n)), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, n, lr, beta1,
                       beta2, eps, weight_decay, decoupled, bc1, bc2_sqrt, grad_scale, (unsigned short*)params_bf16);
    return launch_status();
}

extern "C" int lp_cast_bf16(const float* src, size_t n, void* dst, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(src && dst);
    if (n == 0) return LP_OK;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid_for_n(n)), dim3(256), 0, (hipStream_t)stream, src, n, (unsigned short*)dst);
    return launch_status();
}

extern "C" int lp_permute_cba(const void* src, int A, int B, int C, void* dst, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(src && dst && A > 0 && B > 0 && C > 0);
    hipLaunchKernelGGL(permute_cba_kernel, dim3(grid_for_n((size_t)A * B * C)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)src, A, B, C, (unsigned short*)dst);
    return launch_status();
}
