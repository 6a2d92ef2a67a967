// 3x3 / stride 1 / pad 1 convolution with 64 input and 64 output channels (conv2 of layer1's three blocks), forward and data gradient, with
// the WHOLE filter resident in LDS and square 16 x 16 pixel tiles (round 3).
//
// Why a third kernel: on this shape conv_pipe_kernel's K step is 8 MFMAs per wave, so its per-step cost (counted wait, barrier, address
// arithmetic, the first fragments' LDS latency) and its per-tile store pass - a tile lasts 9 steps - are the time, not the operand stream
// (profiles/archive/r03q_halo_loop_experiments.txt: 253 us with the HALO form, 198 us with no loads at all, against a 52 us MFMA floor).  Here
//   * the 9 x 64 x 64 filter (73.7 KB as 576 rows of 128 B) is loaded ONCE per workgroup and stays in LDS for the whole persistent walk;
//   * a tile is a 16 x 16 block of one image, so its input neighbourhood is 18 x 18 pixels = 324 rows of 128 B (41 KB; a 256-pixel raster
//     run of a 96-wide image needs 460) - two of them fit beside the filter (157.7 KB of the CU's 160), the next tile's neighbourhood
//     loads (direct-to-LDS) while this one is consumed;
//   * with both operands resident the K loop - 9 taps x 4 k-slices = 72 MFMAs per wave - runs without a single barrier or counted wait:
//     ONE barrier per tile (the neighbourhood hand-over), fragment addresses are compile-time offsets from two per-lane bases.
// MFMA operand roles are swapped (weights as A, pixels as B) and the store passes are conv_pipe_kernel's (per wave through a private LDS
// corner, full-line 16-B global stores, per-thread BatchNorm sums; the data gradient reads back z, recomputes the ReLU mask and takes the
// BatchNorm-backward sums - the only store-pass form a 3x3 layer of the trunk uses).  The K order (tap-major, 64 channels per tap) and every
// rounding point equal conv_igemm_kernel's: outputs are bit-identical to it (tests/test_emu_conv_pipe.py, tests/test_gpu_fullsize.py).
#pragma once

namespace lp {

constexpr int kR2Halo = 18 * 18;                 // rows of a neighbourhood image
constexpr int kR2HaloRows = 328;                 // ... as staged (41 passes of 8 rows)
constexpr int kR2HaloB = kR2HaloRows * kPRowB;   // 41 984 B
constexpr int kR2WB = 9 * 64 * kPRowB;           // 73 728 B

// INFER (forward only; lp_conv_fwd_act, round 4): out = [relu](acc + bias) - the inference launch of these layers with the BatchNorm folded
// into weights and bias (conv2 of a bottleneck has no residual branch); its own instantiation, the training kernel's code is untouched.
template <int MODE, bool INFER = false>
__global__ __launch_bounds__(512) void conv_res2d_kernel(const unsigned short* __restrict__ X, const unsigned short* __restrict__ Wt,
                                                         unsigned x_bytes, unsigned w_bytes, int B, int H, int W, int ntiles, ConvEpilogue ep) {
    static_assert(MODE == kModeFwd || MODE == kModeDgrad, "forward or data gradient");
    static_assert(!INFER || MODE == kModeFwd, "the inference store pass belongs to the forward kernel");
    constexpr bool kFwd = MODE == kModeFwd;
    __shared__ __attribute__((aligned(16))) unsigned char smem[kR2WB + 2 * kR2HaloB];
    unsigned char* const wlds = smem;
    unsigned char* const halo0 = smem + kR2WB;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const buf_rsrc rsrc_x = make_buf_rsrc(X, x_bytes), rsrc_w = make_buf_rsrc(Wt, w_bytes);
    const int tiles_x = W >> 4, tiles_y = H >> 4, tiles_img = tiles_x * tiles_y;
    const int M = B * H * W;

    // ---- loader geometry: a wave instruction fills 8 rows x 128 B; lane -> (row in the piece, 16-B slot); the chunk a lane fetches is
    // slot ^ ((row >> 1) & 7) - the swizzle lives on the SOURCE side, the LDS side of a direct load is lane-linear
    const int rloc = lane >> 3, slot = lane & 7;
    // neighbourhood rows of this thread: j = i * 64 + wave * 8 + rloc for the passes i = 0 .. 4 (all waves) and, wave 0 only, rows 320 .. 327
    int hoff[6];       // byte offset of the row's pixel relative to the tile's first pixel, or INT_MIN for rows past the 324 real ones
    int hyx[6];        // (hy << 8) | hx of the row inside the 18 x 18 neighbourhood
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int j = (i < 5 ? i * 64 + wave * 8 : 320) + rloc;
        const int hy = j / 18, hx = j - hy * 18;
        const int chunk = slot ^ ((j >> 1) & 7);
        hyx[i] = (hy << 8) | hx;
        hoff[i] = j < kR2Halo ? ((hy - 1) * W + (hx - 1)) * 128 + chunk * 16 : (int)0x80000000;
    }
    auto halo_load = [&](int vt, int buf) {
        if (vt >= ntiles) return;   // (nothing to stage; no counted waits in this kernel)
        const int tile = xcd_remap(vt, ntiles);
        const int b = tile / tiles_img, t2 = tile - b * tiles_img;
        const int ty = t2 / tiles_x, tx = t2 - ty * tiles_x;
        const int y0 = ty * 16, x0 = tx * 16;
        const int base = ((b * H + y0) * W + x0) * 128;
        unsigned char* dst = halo0 + buf * kR2HaloB;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (i == 5 && wave != 0) break;
            const int hy = hyx[i] >> 8, hx = hyx[i] & 255;
            const int y = y0 + hy - 1, x = x0 + hx - 1;
            const bool ok = hoff[i] != (int)0x80000000 && y >= 0 && y < H && x >= 0 && x < W;
            const unsigned voff = ok ? (unsigned)(base + hoff[i]) : ~0u;   // border: the range check returns zeros
            buf_load16_lds(rsrc_x, dst + (i < 5 ? i * (64 * kPRowB) + wave * (8 * kPRowB) : 320 * kPRowB), voff, 0u);
        }
    };

    // ---- the filter, once: LDS row (tap * 64 + n) = the 64 k-values of output channel n under tap (r, s): global row n, byte offset tap * 128
    {
#pragma unroll
        for (int i = 0; i < 9; ++i) {   // pass i = tap i: rows i * 64 + wave * 8 + rloc
            const int n = wave * 8 + rloc;
            const int row = i * 64 + n;
            const int chunk = slot ^ ((row >> 1) & 7);
            buf_load16_lds(rsrc_w, wlds + i * (64 * kPRowB) + wave * (8 * kPRowB), (unsigned)(n * (9 * 128) + chunk * 16), (unsigned)(i * 128));
        }
    }
    halo_load(blockIdx.x, 0);

    // ---- MFMA side
    const int fr = lane & 31, fg = lane >> 5;
    // this lane's two pixels (mt = 0, 1): p = wm * 64 + mt * 32 + fr -> (py, px) = (p >> 4, p & 15); neighbourhood row of tap (0, 0)
    int hrow[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int p = wm * 64 + mt * 32 + fr;
        hrow[mt] = (p >> 4) * 18 + (p & 15);
    }
    const int wrow0 = wn * 32 + fr;   // weight row of tap 0

    // ---- store-pass geometry (conv_pipe_kernel's, NT = 1: a wave owns 64 pixels x 32 channels)
    constexpr int CP = 4, RP = 16;
    const int pc = lane % CP, prow = lane / CP;
    float s0[8], s1[8], mu[8], sc[8], be[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) s0[q] = s1[q] = 0.f;
    int st_seg_off = -1;
    const bool want_stats = ep.stats_sums != nullptr;
    const int N = 64;
    auto stats_flush = [&](float* scratch) {
        if (st_seg_off >= 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
#pragma unroll
                for (int msk = CP; msk < 64; msk <<= 1) {
                    s0[q] += __shfl_xor(s0[q], msk, 64);
                    s1[q] += __shfl_xor(s1[q], msk, 64);
                }
            }
            __syncthreads();
            if (lane < CP) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    scratch[(wave * 2 + 0) * 32 + lane * 8 + q] = s0[q];
                    scratch[(wave * 2 + 1) * 32 + lane * 8 + q] = s1[q];
                }
            }
            __syncthreads();
            if (tid < 2 * N) {
                const int comp = tid / N, cl = tid % N;
                const int wn_ = cl / 32, c = cl % 32;
                float t = 0.f;
#pragma unroll
                for (int w4 = 0; w4 < 4; ++w4) t += scratch[((wn_ * 4 + w4) * 2 + comp) * 32 + c];
                if (!kFwd && comp == 1) t *= ep.bn_invstd[st_seg_off + cl];   // sum dy (z - mean) -> sum dy xhat
                stats_emit(ep, 2 * st_seg_off + comp * N + cl, t);
            }
            __syncthreads();
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) s0[q] = s1[q] = 0.f;
    };

    f32x16 acc[2];
    int buf = 0;
    for (int vt = blockIdx.x; vt < ntiles; vt += gridDim.x) {
        const int tile = xcd_remap(vt, ntiles);
        const int b = tile / tiles_img, t2 = tile - b * tiles_img;
        const int ty = t2 / tiles_x, tx = t2 - ty * tiles_x;
        const int m_tile = (b * H + ty * 16) * W + tx * 16;   // output row of the tile's first pixel
        const int seg_off = (ep.seg_images > 0 && b >= ep.seg_images) ? N : 0;
        // the store pass's scratch: the neighbourhood image consumed by the PREVIOUS tile (free until the barrier below lets its refill start)
        unsigned char* const spare = halo0 + (buf ^ 1) * kR2HaloB;
        if (seg_off != st_seg_off) {   // (workgroup-uniform) first tile / the other BatchNorm segment
            if (want_stats) stats_flush(reinterpret_cast<float*>(spare));
            st_seg_off = seg_off;
            if (!kFwd) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int n = wn * 32 + pc * 8 + q;
                    mu[q] = ep.bn_mean[seg_off + n];
                    sc[q] = ep.bn_invstd[seg_off + n] * ep.bn_gamma[n];
                    be[q] = ep.bn_beta[n];
                }
            }
        }
        // the data gradient's read-back of z for this tile (both 32-pixel chunks; they travel under the MFMAs)
        unsigned rboff[2][2];
        u16x8 rbz[2][2];
        if (!kFwd) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int p = wm * 64 + mt * 32 + h * 16 + prow;
                    rboff[mt][h] = (unsigned)(m_tile + (p >> 4) * W + (p & 15)) * 64u + (unsigned)(wn * 32 + pc * 8);
                    rbz[mt][h] = load8(ep.bn_z + rboff[mt][h]);
                }
        }
        // the filter and this tile's neighbourhood have landed (the data gradient's 4 read-backs, issued last, may still fly)
        if (kFwd) LP_WAIT_VM(0);
        else LP_WAIT_VM(4);
        LP_RAW_BARRIER();              // ... everyone's have, and everyone is done with the other neighbourhood image (MFMAs and store pass)
        halo_load(vt + gridDim.x, buf ^ 1);
        const unsigned char* hb = halo0 + buf * kR2HaloB;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mt][e] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int r = tap / 3, s = tap - r * 3;
            const int dh = kFwd ? r * 18 + s : (2 - r) * 18 + (2 - s);   // the data gradient gathers dy at (y + 1 - r, x + 1 - s)
            const int wrow = tap * 64 + wrow0;
            const unsigned wbase = (unsigned)(wrow * kPRowB), wsw = (unsigned)((wrow >> 1) & 7);
            unsigned abase[2], asw[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int row = hrow[mt] + dh;
                abase[mt] = (unsigned)(row * kPRowB);
                asw[mt] = (unsigned)((row >> 1) & 7);
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const unsigned c = (unsigned)(kk * 2 + fg);
                const bf16x8 wv = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u16x8*>(wlds + wbase + ((c ^ wsw) << 4)));
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const bf16x8 av = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u16x8*>(hb + abase[mt] + ((c ^ asw[mt]) << 4)));
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv, av, acc[mt], 0, 0, 0);   // roles swapped: D[channel][pixel]
                }
            }
        }
        LP_RAW_BARRIER();   // every wave is done reading this neighbourhood image: its bytes are the store pass's scratch from here on
        unsigned char* const stg_all = halo0 + buf * kR2HaloB;
        if (kFwd) {
            // lane (pixel fr, half fg) holds the channels 8 j + 4 fg + (0..3) of its wave's 32 in acc[mt][4 j .. 4 j + 3]
            constexpr int ROWB = 64 + 16;
            unsigned char* stg = stg_all + wave * (32 * ROWB);
            if (INFER && ep.bias != nullptr) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(ep.bias + wn * 32 + 8 * j + 4 * fg);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[mt][4 * j + e] += bv[e];
                }
            }
            if (INFER && ep.relu_fwd) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[mt][e] = fmaxf(acc[mt][e], 0.f);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
                    const u32x2_t pk = {pack_bf16x2(acc[mt][4 * j], acc[mt][4 * j + 1]), pack_bf16x2(acc[mt][4 * j + 2], acc[mt][4 * j + 3])};
                    *reinterpret_cast<u32x2_t*>(stg + fr * ROWB + (8 * j + 4 * fg) * 2) = pk;
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int ps = 0; ps < 32 / RP; ++ps) {
                    const int row = ps * RP + prow;
                    const u16x8 w = *reinterpret_cast<const u16x8*>(stg + row * ROWB + pc * 16);
                    const int p = wm * 64 + mt * 32 + row;
                    const unsigned off = (unsigned)(m_tile + (p >> 4) * W + (p & 15)) * 64u + (unsigned)(wn * 32 + pc * 8);
                    *reinterpret_cast<u16x8*>(ep.out_bf16 + off) = w;
                    if (want_stats) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const float vr = bf16_to_f32(w[q]);
                            s0[q] += vr;
                            s1[q] = fmaf(vr, vr, s1[q]);
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        } else {
            constexpr int ROWF = 128 + 16;   // fp32 row of the wave's 32 channels + pad
            unsigned char* stg = stg_all + wave * (16 * ROWF);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if ((fr >> 4) == h) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const f32x4 p4 = {acc[mt][4 * j], acc[mt][4 * j + 1], acc[mt][4 * j + 2], acc[mt][4 * j + 3]};
                            *reinterpret_cast<f32x4*>(stg + (fr & 15) * ROWF + (8 * j + 4 * fg) * 4) = p4;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(stg + prow * ROWF + pc * 32);
                    const f32x4 hi = *reinterpret_cast<const f32x4*>(stg + prow * ROWF + pc * 32 + 16);
                    __builtin_amdgcn_wave_barrier();
                    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    float zc[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        zc[q] = bf16_to_f32(rbz[mt][h][q]) - mu[q];
                        if (!(fmaf(zc[q], sc[q], be[q]) > 0x1p-134f)) v[q] = 0.f;
                    }
                    const u16x8 w = pack_bf16x8(v);
                    *reinterpret_cast<u16x8*>(ep.out_bf16 + rboff[mt][h]) = w;
                    if (want_stats) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const float vr = bf16_to_f32(w[q]);
                            s0[q] += vr;
                            s1[q] = fmaf(vr, zc[q], s1[q]);
                        }
                    }
                }
            }
        }
        buf ^= 1;
    }
    LP_WAIT_VM(0);
    LP_RAW_BARRIER();
    if (want_stats) stats_flush(reinterpret_cast<float*>(halo0 + (buf ^ 1) * kR2HaloB));
    (void)M;
}

// ------------------------------------------------------------------------------------------------------------
// The 7x7 / stride 2 / pad 3 stem (NHWC4 bf16 input, 64 output channels, K = 8 x 8 x 4 = 256 with the 8th row / column of taps zero) in the
// same style: 16 x 16 output tiles, the [64][256] filter resident in LDS, the 38 x 40-pixel input neighbourhood of a tile (12 KB) staged by
// direct-to-LDS loads while the previous tile is consumed, 32 MFMAs per wave and tile without a barrier or counted wait.  The im2col row of
// an output pixel under filter row r is 64 contiguous bytes of input (8 taps x 4 channels), so the pixel operand of k-slice (r, half) is
// 16 B at input pixel (2 y - 3 + r, 2 x - 3 + 4 half + 2 fg): read straight from the staged image as two 8-B pieces (the image starts at the
// EVEN column 2 x0 - 4 so that the global 16-B chunks are whole pixel pairs inside or outside the frame; the fragments then start on odd
// pixels).  K order and rounding points are conv_igemm_kernel<64, stem>'s: bit-identical outputs.  The weights sit in LDS as one
// [64 rows][32 B] image per k-slice (a [64][512 B] image would put a wave's 32 rows on the same banks).
constexpr int kS2RowPx = 40, kS2Rows = 38;
constexpr int kS2ImgB = 768 * 16;   // 38 x 40 x 8 B = 12 160 B, staged as 12 wave pieces of 1 KB
constexpr int kS2WB = 64 * 512;

__global__ __launch_bounds__(512, 2) void conv_stem2d_kernel(const unsigned short* __restrict__ X, const unsigned short* __restrict__ Wt,
                                                          unsigned x_bytes, int B, int Ho, int Wo, int ntiles, ConvEpilogue ep) {
    constexpr int kScratchB = 8 * 32 * (64 + 16);   // the 8 waves' store-pass corners (20 KB): 76 KB in all, two workgroups per CU
    __shared__ __attribute__((aligned(16))) unsigned char smem[kS2WB + 2 * kS2ImgB + kScratchB];
    unsigned char* const wlds = smem;
    unsigned char* const img0 = smem + kS2WB;
    unsigned char* const scratch = smem + kS2WB + 2 * kS2ImgB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const int Hi = 2 * Ho, Wi = 2 * Wo;
    const buf_rsrc rsrc_x = make_buf_rsrc(X, x_bytes), rsrc_w = make_buf_rsrc(Wt, (unsigned)kS2WB);
    const int tiles_x = Wo >> 4, tiles_img = tiles_x * (Ho >> 4);

    // chunk q (16 B = 2 pixels) of the staged image: row q / 20, pixel pair q % 20; this thread stages q = tid and (waves 0 - 3) 512 + tid
    int crow[2], ccol[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = i * 512 + tid;
        crow[i] = q / 20;
        ccol[i] = (q - crow[i] * 20) * 2;
    }
    auto img_load = [&](int vt, int buf) {
        if (vt >= ntiles) return;
        const int tile = xcd_remap(vt, ntiles);
        const int b = tile / tiles_img, t2 = tile - b * tiles_img;
        const int ty = t2 / tiles_x, tx = t2 - ty * tiles_x;
        const int iy0 = 32 * ty - 3, ix0 = 32 * tx - 4;
        unsigned char* dst = img0 + buf * kS2ImgB;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i == 1 && wave >= 4) break;
            const int iy = iy0 + crow[i], ix = ix0 + ccol[i];
            const bool ok = crow[i] < kS2Rows && iy >= 0 && iy < Hi && ix >= 0 && ix < Wi;
            const unsigned voff = ok ? (unsigned)(((b * Hi + iy) * Wi + ix) * 8) : ~0u;
            buf_load16_lds(rsrc_x, dst + i * (512 * 16) + wave * 1024, voff, 0u);
        }
    };
    // the filter, once: wave instruction (kk, rows n0 .. n0 + 31): lane -> (n = n0 + lane / 2, fg = lane % 2), source chunk 2 kk + fg of row n
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = wave * 4 + i, kk = piece >> 1, n = (piece & 1) * 32 + (lane >> 1), fgl = lane & 1;
        buf_load16_lds(rsrc_w, wlds + piece * 1024, (unsigned)(n * 512 + (kk * 2 + fgl) * 16), 0u);
    }
    img_load(blockIdx.x, 0);

    const int fr = lane & 31, fg = lane >> 5;
    unsigned abase[2];   // byte offset of input pixel (2 py, 2 px + 2 fg + 1) in the staged image, per mt
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int p = wm * 64 + mt * 32 + fr;
        abase[mt] = (unsigned)(((2 * (p >> 4)) * kS2RowPx + 2 * (p & 15) + 2 * fg + 1) * 8);
    }
    const unsigned wbase = (unsigned)((wn * 32 + fr) * 32 + fg * 16);

    constexpr int CP = 4, RP = 16, N = 64;
    const int pc = lane % CP, prow = lane / CP;
    float s0[8], s1[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) s0[q] = s1[q] = 0.f;
    int st_seg_off = -1;
    const bool want_stats = ep.stats_sums != nullptr;
    auto stats_flush = [&](float* red) {
        if (st_seg_off >= 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
#pragma unroll
                for (int msk = CP; msk < 64; msk <<= 1) {
                    s0[q] += __shfl_xor(s0[q], msk, 64);
                    s1[q] += __shfl_xor(s1[q], msk, 64);
                }
            }
            __syncthreads();
            if (lane < CP) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    red[(wave * 2 + 0) * 32 + lane * 8 + q] = s0[q];
                    red[(wave * 2 + 1) * 32 + lane * 8 + q] = s1[q];
                }
            }
            __syncthreads();
            if (tid < 2 * N) {
                const int comp = tid / N, cl = tid % N;
                const int wn_ = cl / 32, c = cl % 32;
                float t = 0.f;
#pragma unroll
                for (int w4 = 0; w4 < 4; ++w4) t += red[((wn_ * 4 + w4) * 2 + comp) * 32 + c];
                stats_emit(ep, 2 * st_seg_off + comp * N + cl, t);
            }
            __syncthreads();
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) s0[q] = s1[q] = 0.f;
    };

    f32x16 acc[2];
    int buf = 0;
    for (int vt = blockIdx.x; vt < ntiles; vt += gridDim.x) {
        const int tile = xcd_remap(vt, ntiles);
        const int b = tile / tiles_img, t2 = tile - b * tiles_img;
        const int ty = t2 / tiles_x, tx = t2 - ty * tiles_x;
        const int m_tile = (b * Ho + ty * 16) * Wo + tx * 16;
        const int seg_off = (ep.seg_images > 0 && b >= ep.seg_images) ? N : 0;
        if (seg_off != st_seg_off) {
            if (want_stats) stats_flush(reinterpret_cast<float*>(scratch));   // (the waves' store-pass corners are idle between tiles)
            st_seg_off = seg_off;
        }
        LP_WAIT_VM(0);        // the filter and this tile's image have landed (and the previous tile's stores have left)
        LP_RAW_BARRIER();     // ... everyone's have, and everyone is done with the other image
        img_load(vt + gridDim.x, buf ^ 1);
        const unsigned char* ib = img0 + buf * kS2ImgB;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mt][e] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {   // k-slice (r, half): taps 4 half .. 4 half + 3 of filter row r
            const int r = kk >> 1, half = kk & 1;
            const bf16x8 wv = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u16x8*>(wlds + kk * 2048 + wbase));
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const unsigned char* ap = ib + abase[mt] + (r * kS2RowPx + 4 * half) * 8;
                const u16x4 lo = *reinterpret_cast<const u16x4*>(ap), hi = *reinterpret_cast<const u16x4*>(ap + 8);
                const u16x8 av = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv, __builtin_bit_cast(bf16x8, av), acc[mt], 0, 0, 0);
            }
        }
        // store pass: per wave through its private corner (no barrier: the images are only refilled behind the NEXT tile's barrier)
        constexpr int ROWB = 64 + 16;
        unsigned char* stg = scratch + wave * (32 * ROWB);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
                const u32x2_t pk = {pack_bf16x2(acc[mt][4 * j], acc[mt][4 * j + 1]), pack_bf16x2(acc[mt][4 * j + 2], acc[mt][4 * j + 3])};
                *reinterpret_cast<u32x2_t*>(stg + fr * ROWB + (8 * j + 4 * fg) * 2) = pk;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ps = 0; ps < 32 / RP; ++ps) {
                const int row = ps * RP + prow;
                const u16x8 w = *reinterpret_cast<const u16x8*>(stg + row * ROWB + pc * 16);
                const int p = wm * 64 + mt * 32 + row;
                const unsigned off = (unsigned)(m_tile + (p >> 4) * Wo + (p & 15)) * 64u + (unsigned)(wn * 32 + pc * 8);
                *reinterpret_cast<u16x8*>(ep.out_bf16 + off) = w;
                if (want_stats) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const float vr = bf16_to_f32(w[q]);
                        s0[q] += vr;
                        s1[q] = fmaf(vr, vr, s1[q]);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        buf ^= 1;
    }
    LP_WAIT_VM(0);
    LP_RAW_BARRIER();
    if (want_stats) stats_flush(reinterpret_cast<float*>(scratch));
}

}  // namespace lp
