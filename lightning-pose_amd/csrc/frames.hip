// Batch producers on the device: decoded video frames -> the model's input batch, and labeled
// keypoints -> model-space keypoints + visibility (the heat-map targets themselves come from lp_heatmap_gen).  gfx950.
//
// Reference arithmetic (paths relative to the reference tree; the image operators live in NVIDIA DALI / imgaug, which are
// not vendored - their published definitions are restated here and in oracle/restated.py, parity UNPINNED for the images):
//   data/video/dali.py:135-192      video_pipe: fn.resize -> [fn.transforms.rotation/scale -> fn.warp_affine(fill_value=0,
//                                   inverse_map=False) -> fn.brightness_contrast -> fn.noise.shot] -> /255 ->
//                                   fn.crop_mirror_normalize(mean, std, output_layout="FCHW")
//   data/datasets.py:262-376        BaseTrackingDataset.__getitem__: imgaug Resize of image + keypoints, optional hflip with
//                                   the left/right keypoint swap
//   data/datasets.py:465-472        visibility synthesised from NaN labels (uniform_heatmaps)
//   data/datasets.py:496-508        HeatmapDataset.compute_heatmap: keypoints pushed out of the frame become NaN
//
// All kernels are HBM-bound byte / float streams: one lane per output pixel with x fastest (coalesced plane writes; the
// source taps of neighbouring lanes overlap and come from L2), no LDS, no host synchronisation.
#include "lp_common.h"

namespace lp {

struct NormSpec {
    float mean[3], inv_std[3];
};

// ---- antialiased linear resize (triangle filter whose radius grows with the down-scale factor) -------------------------
struct ResizeSpec {
    int Hs, Ws, H, W;
    long long frame_stride;  // bytes between source frames
    int row_stride;          // bytes between source rows
    float scale_y, scale_x;  // source px per output px
    float ry, rx;            // filter radius in source px: max(1, scale)
    int border;              // LP_BORDER_RENORM: window cut at the edge and renormalised;  LP_BORDER_CLAMP: edge pixels replicated
};

// window [lo, hi) of source pixels j with |j + 0.5 - c| < r, cut to [0, n) in renormalising mode
__device__ __forceinline__ void tri_window(float c, float r, int n, int border, int& lo, int& hi) {
    lo = (int)floorf(c - r + 0.5f);
    hi = (int)floorf(c + r + 0.5f);
    if (border == LP_BORDER_RENORM) {
        lo = lo < 0 ? 0 : lo;
        hi = hi > n ? n : hi;
    }
}

__device__ __forceinline__ float tri_weight(int j, float c, float inv_r) { return fmaxf(0.f, 1.f - fabsf(((float)j + 0.5f - c) * inv_r)); }

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

template <bool FINISH>
__global__ __launch_bounds__(256) void frames_resize_kernel(const unsigned char* __restrict__ src, ResizeSpec p, NormSpec nrm,
                                                            float* __restrict__ dst) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), s = blockIdx.z;
    if (x >= p.W || y >= p.H) return;
    const float cx = ((float)x + 0.5f) * p.scale_x, cy = ((float)y + 0.5f) * p.scale_y;
    const float irx = 1.f / p.rx, iry = 1.f / p.ry;
    int xlo, xhi, ylo, yhi;
    tri_window(cx, p.rx, p.Ws, p.border, xlo, xhi);
    tri_window(cy, p.ry, p.Hs, p.border, ylo, yhi);
    const unsigned char* frame = src + (size_t)s * p.frame_stride;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, wsum_y = 0.f, wsum_x = 0.f;
    for (int i = xlo; i < xhi; ++i) wsum_x += tri_weight(i, cx, irx);
    for (int j = ylo; j < yhi; ++j) {
        const float wy = tri_weight(j, cy, iry);
        wsum_y += wy;
        const unsigned char* row = frame + (size_t)clampi(j, 0, p.Hs - 1) * p.row_stride;
        float r0 = 0.f, r1 = 0.f, r2 = 0.f;
        for (int i = xlo; i < xhi; ++i) {
            const float wx = tri_weight(i, cx, irx);
            const unsigned char* px = row + clampi(i, 0, p.Ws - 1) * 3;
            r0 = fmaf(wx, (float)px[0], r0);
            r1 = fmaf(wx, (float)px[1], r1);
            r2 = fmaf(wx, (float)px[2], r2);
        }
        acc0 = fmaf(wy, r0, acc0);
        acc1 = fmaf(wy, r1, acc1);
        acc2 = fmaf(wy, r2, acc2);
    }
    const float inv = 1.f / (wsum_x * wsum_y);
    acc0 *= inv;
    acc1 *= inv;
    acc2 *= inv;
    if (FINISH) {  // /255, normalise, planes (the reference's "FCHW")
        const size_t plane = (size_t)p.H * p.W;
        float* o = dst + (size_t)s * 3 * plane + (size_t)y * p.W + x;
        o[0] = (acc0 * (1.f / 255.f) - nrm.mean[0]) * nrm.inv_std[0];
        o[plane] = (acc1 * (1.f / 255.f) - nrm.mean[1]) * nrm.inv_std[1];
        o[2 * plane] = (acc2 * (1.f / 255.f) - nrm.mean[2]) * nrm.inv_std[2];
    } else {  // interleaved fp32 in [0, 255]: what the augmentation kernel samples
        float* o = dst + (((size_t)s * p.H + y) * p.W + x) * 3;
        o[0] = acc0;
        o[1] = acc1;
        o[2] = acc2;
    }
}

// ---- bicubic resize without antialiasing (imgaug `iaa.Resize`'s default interpolation = OpenCV INTER_CUBIC: Keys kernel with
// A = -0.75, half-pixel centres, the 4 x 4 taps clamped to the image; reference data/datasets.py:137-143 builds
// `iaa.Resize({"height": ..., "width": ...})` as the last imgaug step of every labeled image).  imgaug hands back a uint8 image, so
// the interpolated value is rounded and saturated to [0, 255] before the /255 + normalise of the dataset's ToTensor / Normalize.
__device__ __forceinline__ void cubic_taps(float c, int& i0, float (&w)[4]) {
    const float A = -0.75f;
    const float f = c - 0.5f;
    const float fl = floorf(f);
    const float t = f - fl;
    i0 = (int)fl - 1;
    auto k1 = [&](float u) { return ((A + 2.f) * u - (A + 3.f)) * u * u + 1.f; };          // |u| <= 1
    auto k2 = [&](float u) { return ((A * u - 5.f * A) * u + 8.f * A) * u - 4.f * A; };    // 1 < |u| < 2
    w[0] = k2(t + 1.f);
    w[1] = k1(t);
    w[2] = k1(1.f - t);
    w[3] = k2(2.f - t);
}

template <bool FINISH>
__global__ __launch_bounds__(256) void frames_resize_cubic_kernel(const unsigned char* __restrict__ src, ResizeSpec p, NormSpec nrm,
                                                                  int round_u8, float* __restrict__ dst) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), s = blockIdx.z;
    if (x >= p.W || y >= p.H) return;
    int x0, y0;
    float wx[4], wy[4];
    cubic_taps(((float)x + 0.5f) * p.scale_x, x0, wx);
    cubic_taps(((float)y + 0.5f) * p.scale_y, y0, wy);
    const unsigned char* frame = src + (size_t)s * p.frame_stride;
    float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned char* row = frame + (size_t)clampi(y0 + j, 0, p.Hs - 1) * p.row_stride;
        float r[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned char* px = row + clampi(x0 + i, 0, p.Ws - 1) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) r[c] = fmaf(wx[i], (float)px[c], r[c]);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] = fmaf(wy[j], r[c], acc[c]);
    }
    if (round_u8) {
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] = fminf(fmaxf(floorf(acc[c] + 0.5f), 0.f), 255.f);
    }
    if (FINISH) {
        const size_t plane = (size_t)p.H * p.W;
        float* o = dst + (size_t)s * 3 * plane + (size_t)y * p.W + x;
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c * plane] = (acc[c] * (1.f / 255.f) - nrm.mean[c]) * nrm.inv_std[c];
    } else {
        float* o = dst + (((size_t)s * p.H + y) * p.W + x) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = acc[c];
    }
}

// ---- counter-based random numbers (Philox4x32-10): a pixel's stream depends only on (seed, frame, pixel) ---------------
struct Philox {  // scalar members only: nothing here is indexed dynamically, so the state stays in registers
    unsigned c0, c1, c2, k0, k1, o0, o1, o2, o3;
    int have;
    __device__ __forceinline__ void init(unsigned long long seed, unsigned ctr0, unsigned ctr1) {
        k0 = (unsigned)seed;
        k1 = (unsigned)(seed >> 32);
        c0 = ctr0;
        c1 = ctr1;
        c2 = 0;
        have = 0;
    }
    __device__ __forceinline__ void round4() {
        unsigned ka = k0, kb = k1, x0 = c0, x1 = c1, x2 = c2, x3 = 0;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const unsigned long long p0 = (unsigned long long)0xD2511F53u * x0, p1 = (unsigned long long)0xCD9E8D57u * x2;
            const unsigned y0 = (unsigned)(p1 >> 32) ^ x1 ^ ka, y1 = (unsigned)p1, y2 = (unsigned)(p0 >> 32) ^ x3 ^ kb, y3 = (unsigned)p0;
            x0 = y0;
            x1 = y1;
            x2 = y2;
            x3 = y3;
            ka += 0x9E3779B9u;
            kb += 0xBB67AE85u;
        }
        o0 = x0;
        o1 = x1;
        o2 = x2;
        o3 = x3;
        c2 += 1;  // next block of four
        have = 4;
    }
    __device__ __forceinline__ float uniform() {  // (0, 1)
        if (have == 0) round4();
        const unsigned v = have == 4 ? o0 : (have == 3 ? o1 : (have == 2 ? o2 : o3));
        --have;
        return ((float)(v >> 8) + 0.5f) * (1.f / 16777216.f);
    }
};

// Poisson(lam): product-of-uniforms for small means, rounded normal approximation above (|skew| < 0.3)
__device__ __forceinline__ float poisson(float lam, Philox& rng) {
    if (!(lam > 0.f)) return 0.f;
    if (lam < 12.f) {
        const float limit = __expf(-lam);
        float prod = rng.uniform();
        int k = 0;
        while (prod > limit && k < 96) {
            prod *= rng.uniform();
            ++k;
        }
        return (float)k;
    }
    const float u1 = rng.uniform(), u2 = rng.uniform();
    const float z = sqrtf(-2.f * logf(u1)) * cosf(6.283185307179586f * u2);
    return fmaxf(0.f, floorf(lam + sqrtf(lam) * z + 0.5f));
}

// ---- warp_affine (bilinear, fill 0) -> brightness / contrast -> shot noise -> /255 -> normalise -> planes --------------
struct AugmentSpec {
    int H, W;
    int warp;                // 0: identity sampling
    float m[6];              // destination -> source map on pixel-centre coordinates ((x + 0.5, y + 0.5) -> source + 0.5)
    float brightness, contrast, contrast_center;
    float shot_factor;       // 0: no noise
    unsigned long long seed;
};

__device__ __forceinline__ void fetch3(const float* __restrict__ frame, int W, int H, int xi, int yi, float w, float& a0, float& a1,
                                       float& a2) {
    if (xi < 0 || yi < 0 || xi >= W || yi >= H) return;  // fill_value = 0
    const float* px = frame + ((size_t)yi * W + xi) * 3;
    a0 = fmaf(w, px[0], a0);
    a1 = fmaf(w, px[1], a1);
    a2 = fmaf(w, px[2], a2);
}

__global__ __launch_bounds__(256) void frames_augment_kernel(const float* __restrict__ src, AugmentSpec p, NormSpec nrm,
                                                             float* __restrict__ dst) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), s = blockIdx.z;
    if (x >= p.W || y >= p.H) return;
    const float* frame = src + (size_t)s * p.H * p.W * 3;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
    if (p.warp) {
        const float dx = (float)x + 0.5f, dy = (float)y + 0.5f;
        const float sx = fmaf(p.m[0], dx, fmaf(p.m[1], dy, p.m[2])) - 0.5f, sy = fmaf(p.m[3], dx, fmaf(p.m[4], dy, p.m[5])) - 0.5f;
        const float fx0 = floorf(sx), fy0 = floorf(sy);
        const float ax = sx - fx0, ay = sy - fy0;
        // far outside (or non-finite): every tap is fill
        if (sx > -2.f && sy > -2.f && sx < (float)p.W + 1.f && sy < (float)p.H + 1.f) {
            const int x0 = (int)fx0, y0 = (int)fy0;
            fetch3(frame, p.W, p.H, x0, y0, (1.f - ax) * (1.f - ay), v0, v1, v2);
            fetch3(frame, p.W, p.H, x0 + 1, y0, ax * (1.f - ay), v0, v1, v2);
            fetch3(frame, p.W, p.H, x0, y0 + 1, (1.f - ax) * ay, v0, v1, v2);
            fetch3(frame, p.W, p.H, x0 + 1, y0 + 1, ax * ay, v0, v1, v2);
        }
    } else {
        const float* px = frame + ((size_t)y * p.W + x) * 3;
        v0 = px[0];
        v1 = px[1];
        v2 = px[2];
    }
    // out = brightness * (centre + contrast * (in - centre))
    const float cc = p.contrast_center, cb = p.brightness * p.contrast, c0 = p.brightness * (cc - p.contrast * cc);
    v0 = fmaf(cb, v0, c0);
    v1 = fmaf(cb, v1, c0);
    v2 = fmaf(cb, v2, c0);
    if (p.shot_factor > 0.f) {  // out = Poisson(max(in, 0) / factor) * factor
        Philox rng;
        rng.init(p.seed, (unsigned)(y * p.W + x), (unsigned)s);
        const float inv_f = 1.f / p.shot_factor;
        v0 = poisson(fmaxf(v0, 0.f) * inv_f, rng) * p.shot_factor;
        v1 = poisson(fmaxf(v1, 0.f) * inv_f, rng) * p.shot_factor;
        v2 = poisson(fmaxf(v2, 0.f) * inv_f, rng) * p.shot_factor;
    }
    const size_t plane = (size_t)p.H * p.W;
    float* o = dst + (size_t)s * 3 * plane + (size_t)y * p.W + x;
    o[0] = (v0 * (1.f / 255.f) - nrm.mean[0]) * nrm.inv_std[0];
    o[plane] = (v1 * (1.f / 255.f) - nrm.mean[1]) * nrm.inv_std[1];
    o[2 * plane] = (v2 * (1.f / 255.f) - nrm.mean[2]) * nrm.inv_std[2];
}

// ---- labeled keypoints: source px -> [affine] -> resize scale -> [hflip + left/right swap] -> out-of-frame = NaN; visibility ----
struct LabeledSpec {
    int B, K, H, W;
    int uniform_heatmaps;
};

__global__ __launch_bounds__(256) void labeled_keypoints_kernel(const float* __restrict__ kp, const float* __restrict__ src_hw,
                                                                const float* __restrict__ affine, const int* __restrict__ hflip,
                                                                const int* __restrict__ swap, const int* __restrict__ vis_in, LabeledSpec p,
                                                                float* __restrict__ kp_out, int* __restrict__ vis_out) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.B * p.K) return;
    const int b = idx / p.K, k = idx - b * p.K;
    const bool flip = hflip != nullptr && hflip[b] != 0;
    const int ks = (flip && swap != nullptr) ? swap[k] : k;  // the flipped sample's keypoint k is the source's keypoint swap[k]
    const float x0 = kp[((size_t)b * p.K + ks) * 2], y0 = kp[((size_t)b * p.K + ks) * 2 + 1];
    float x = x0, y = y0;
    if (affine != nullptr) {
        const float* a = affine + (size_t)b * 6;
        x = fmaf(a[0], x0, fmaf(a[1], y0, a[2]));
        y = fmaf(a[3], x0, fmaf(a[4], y0, a[5]));
    }
    x = x / src_hw[b * 2 + 1] * (float)p.W;  // imgaug keypoint projection: (x / from_width) * to_width
    y = y / src_hw[b * 2] * (float)p.H;
    if (flip) x = (float)p.W - x;
    const bool out = (x < 0.f) || (y < 0.f) || (x >= (float)p.W) || (y >= (float)p.H);
    if (out) x = y = __int_as_float(0x7fc00000);
    kp_out[(size_t)idx * 2] = x;
    kp_out[(size_t)idx * 2 + 1] = y;
    if (vis_out != nullptr) {
        int v;
        if (vis_in != nullptr) v = vis_in[(size_t)b * p.K + ks];
        else v = (x0 != x0) ? (p.uniform_heatmaps ? 1 : 0) : 2;  // from the label as stored, before any augmentation
        vis_out[idx] = v;
    }
}

}  // namespace lp

// ------------------------------------------------------------------------------------------------------- C ABI
static bool norm_spec(const lp_frame_norm* n, lp::NormSpec& out) {
    for (int c = 0; c < 3; ++c) {
        if (!(n->std[c] > 0.f)) return false;
        out.mean[c] = n->mean[c];
        out.inv_std[c] = 1.f / n->std[c];
    }
    return true;
}

extern "C" int lp_frames_resize(const void* src_u8, int S, int Hs, int Ws, long long frame_stride, int row_stride, int H, int W, int border,
                                const lp_frame_norm* finish_norm, float* dst, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(src_u8 && dst && S > 0 && Hs > 0 && Ws > 0 && H > 0 && W > 0);
    LP_REQUIRE(row_stride >= Ws * 3 && frame_stride >= (long long)row_stride * Hs);
    LP_REQUIRE(border == LP_BORDER_RENORM || border == LP_BORDER_CLAMP);
    if (S > 65535 || (H + 3) / 4 > 65535) return LP_ERR_UNSUPPORTED;
    ResizeSpec p{};
    p.Hs = Hs, p.Ws = Ws, p.H = H, p.W = W, p.frame_stride = frame_stride, p.row_stride = row_stride, p.border = border;
    p.scale_y = (float)((double)Hs / H), p.scale_x = (float)((double)Ws / W);
    p.ry = p.scale_y > 1.f ? p.scale_y : 1.f, p.rx = p.scale_x > 1.f ? p.scale_x : 1.f;
    NormSpec nrm{};
    const dim3 grid((W + 63) / 64, (H + 3) / 4, S);
    if (finish_norm) {
        LP_REQUIRE(norm_spec(finish_norm, nrm));
        hipLaunchKernelGGL(frames_resize_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned char*)src_u8, p, nrm, dst);
    } else {
        hipLaunchKernelGGL(frames_resize_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned char*)src_u8, p, nrm, dst);
    }
    return launch_status();
}

extern "C" int lp_frames_resize_cubic(const void* src_u8, int S, int Hs, int Ws, long long frame_stride, int row_stride, int H, int W,
                                      int round_u8, const lp_frame_norm* finish_norm, float* dst, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(src_u8 && dst && S > 0 && Hs > 0 && Ws > 0 && H > 0 && W > 0);
    LP_REQUIRE(row_stride >= Ws * 3 && frame_stride >= (long long)row_stride * Hs);
    if (S > 65535 || (H + 3) / 4 > 65535) return LP_ERR_UNSUPPORTED;
    ResizeSpec p{};
    p.Hs = Hs, p.Ws = Ws, p.H = H, p.W = W, p.frame_stride = frame_stride, p.row_stride = row_stride, p.border = LP_BORDER_CLAMP;
    p.scale_y = (float)((double)Hs / H), p.scale_x = (float)((double)Ws / W);
    NormSpec nrm{};
    const dim3 grid((W + 63) / 64, (H + 3) / 4, S);
    if (finish_norm) {
        LP_REQUIRE(norm_spec(finish_norm, nrm));
        hipLaunchKernelGGL(frames_resize_cubic_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned char*)src_u8, p, nrm,
                           round_u8, dst);
    } else {
        hipLaunchKernelGGL(frames_resize_cubic_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned char*)src_u8, p, nrm,
                           round_u8, dst);
    }
    return launch_status();
}

extern "C" int lp_frames_augment(const float* src_hwc, int S, int H, int W, const lp_frame_augment* aug, const lp_frame_norm* norm,
                                 float* dst_nchw, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(src_hwc && dst_nchw && aug && norm && S > 0 && H > 0 && W > 0);
    if (S > 65535 || (H + 3) / 4 > 65535) return LP_ERR_UNSUPPORTED;
    AugmentSpec p{};
    p.H = H, p.W = W, p.warp = aug->has_matrix != 0;
    if (p.warp) {  // `matrix` maps source -> destination (DALI inverse_map=False): sample through its inverse, formed in fp64
        const float* a = aug->matrix;
        const double det = (double)a[0] * a[4] - (double)a[1] * a[3];
        LP_REQUIRE(det != 0.0 && det == det);
        const double i00 = a[4] / det, i01 = -a[1] / det, i10 = -a[3] / det, i11 = a[0] / det;
        p.m[0] = (float)i00, p.m[1] = (float)i01, p.m[2] = (float)(-(i00 * a[2] + i01 * a[5]));
        p.m[3] = (float)i10, p.m[4] = (float)i11, p.m[5] = (float)(-(i10 * a[2] + i11 * a[5]));
    }
    p.brightness = aug->brightness, p.contrast = aug->contrast, p.contrast_center = aug->contrast_center;
    LP_REQUIRE(aug->shot_factor >= 0.f);
    p.shot_factor = aug->shot_factor, p.seed = aug->seed;
    NormSpec nrm{};
    LP_REQUIRE(norm_spec(norm, nrm));
    hipLaunchKernelGGL(frames_augment_kernel, dim3((W + 63) / 64, (H + 3) / 4, S), dim3(256), 0, (hipStream_t)stream, src_hwc, p, nrm,
                       dst_nchw);
    return launch_status();
}

extern "C" int lp_labeled_keypoints(const float* kp_src, const float* src_hw, const float* affine, const int* hflip, const int* swap,
                                    const int* vis_in, int uniform_heatmaps, int B, int K, int H, int W, float* kp_out, int* vis_out,
                                    lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(kp_src && src_hw && kp_out && B > 0 && K > 0 && H > 0 && W > 0);
    LabeledSpec p{B, K, H, W, uniform_heatmaps != 0};
    hipLaunchKernelGGL(labeled_keypoints_kernel, dim3((B * K + 255) / 256), dim3(256), 0, (hipStream_t)stream, kp_src, src_hw, affine, hflip,
                       swap, vis_in, p, kp_out, vis_out);
    return launch_status();
}
