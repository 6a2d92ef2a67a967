// Implicit-GEMM convolution on the CDNA4 matrix cores: forward, data-gradient and weight-gradient for the
// ResNet-50 trunk and the transposed-convolution head of the heatmap tracker.  gfx950, bf16 in / fp32 accumulate.
//
// Replaces the cuDNN calls behind `self.backbone(images)` (lightning_pose/models/base.py:398, built at
// models/backbones/factory.py:322-325) and `nn.ConvTranspose2d` in the head (models/heads/heatmap.py:60-69);
// SURVEY.md section 2.1 K1/K2, Appendix B lists every GEMM shape.
//
// Data layout (chosen for the hardware, not inherited): activations NHWC bf16 so the GEMM K index (r, s, c) is
// contiguous in c; weights bf16 [Cout][R][S][Cin] (= torch channels_last) so BOTH operands of every GEMM are
// "K-contiguous rows" and share one staging path; a second copy [Cin][R][S][Cout] feeds the data-gradient.
//
// Tiling: 256 threads = 4 waves (2 x 2).  Workgroup tile 128 x BN (BN = 128 or 64), K step 64.  Each wave owns a
// 64 x BN/2 sub-tile as 2 x (BN/64) MFMA tiles of `v_mfma_f32_32x32x16_bf16`.  Operands are staged global -> VGPR
// -> LDS (16 B per lane; the im2col gather, zero padding and stride handling happen in the address computation,
// nothing is materialised), double-buffered in LDS with ONE barrier per K step; LDS rows are padded to 144 B so
// the 16-B fragment reads are bank-conflict free.  Tile ids are remapped so each XCD's L2 sees a contiguous range
// of tiles that share the activation panel.  The weight-gradient contracts over pixels, which are the ROW index of both
// operands in memory: its tiles keep their memory orientation in LDS and the MFMA fragments come out of ds_read_b64_tr_b16 (the
// gfx950 LDS transpose read); the pixel slices' partial tiles go to a workspace and a second kernel sums them in a fixed order.
#include <stdlib.h>

#include "lp_common.h"

namespace lp {

constexpr int kBM = 128;
constexpr int kBK = 64;
constexpr int kLD = kBK + 8;  // LDS row stride (bf16 elements): 144 B, conflict-free for ds_read_b128

struct ConvGeom {
    int B, Hi, Wi, Ci;  // input  tensor (NHWC)
    int Ho, Wo, Co;     // output tensor (NHWC)
    int R, S, stride, pad;
};

struct ConvEpilogue {
    unsigned short* out_bf16;      // [M][ldo] or nullptr
    float* out_f32;                // [M][ldo] or nullptr
    int ldo;
    int n_store;                   // columns >= n_store are not written
    const float* bias;             // [n] or nullptr
    const unsigned short* addend;  // [M][ldo] bf16 added before the store (gradient accumulation), or nullptr
    const unsigned short* relu_mask;  // [M][ldo] bf16 activation; outputs where it is <= 0 are zeroed (ReLU backward), or nullptr
    const unsigned char* relu_bits;   // same mask at 1 bit per element ([M][ldo/8] bytes, written by lp_bn_apply), or nullptr
    // BatchNorm reductions fused into the store pass (bf16 outputs only; `stats_sums` != nullptr switches them on): the column sums of the
    // values actually stored (after rounding) - sum v and sum v^2 (forward: the statistics of the next BatchNorm) or, with bn_z,
    // sum v * xhat (backward: the two reductions of BatchNorm's gradient).  Every persistent workgroup adds up its tiles' sums in a fixed
    // order and adds the result into `stats_sums` ([segment][2][N] lp_fxsum) in FIXED POINT with integer atomics (lp_common.h: fx_add):
    // the totals repeat bit for bit from run to run (rounds 2 - 3: fp32 atomics in arrival order).
    lp_fxsum* stats_sums;          // [segment][2][N] totals; non-null = take the sums
    const unsigned short* bn_z;    // [M][ldo] bf16 pre-normalisation tensor the gradient belongs to, or nullptr
    const float* bn_mean;          // [N]
    const float* bn_invstd;        // [N]
    const float* bn_gamma;         // [N]  } only for mask_from_z: the ReLU mask is recomputed as
    const float* bn_beta;          // [N]  } bf16(gamma * invstd * (z - mean) + beta) > 0 instead of reading the activation
    int mask_from_z;
    // kEkAZB only (lp_bn_fuse.addend_half): `addend` is a [B][ceil(H / 2)][ceil(W / 2)][ldo] tensor - the data gradient of the block's stride-2
    // projection shortcut on ITS grid - added at the pixels with even row and column only (a full-lattice launch: host-checked)
    int addend_half;
    // kModeAttn (lp_attn_dscores): the store pass turns the accumulated dP = dO V^T into the score gradient
    // dS = attn_scale * P * (dP - D[row]) with P read at the output's own offsets and D = rowsum(dO * O) (GemmExt::d_*)
    const unsigned short* attn_p;
    const float* attn_d;
    float attn_scale;
    // kModeInfer (lp_conv_fwd_act): out = [relu](acc + bias + addend) - a BatchNorm folded into the weights and the bias, the
    // residual read as `addend`, the ReLU applied to the value itself
    int relu_fwd;
    // two BatchNorm segments in one launch (the labeled and the unlabeled frames of a semi-supervised step keep their own batch
    // statistics, as the reference's two forward calls do): images [0, seg_images) are segment 0, the rest segment 1; a tile never
    // straddles the boundary (host-checked: seg_images * rows per image is a multiple of the 128-row tile).  Segment s uses
    // bn_mean / bn_invstd + s * N and adds into stats_sums + s * 2 * N; 0 = one segment
    int seg_images;
    // kEkGeluFwd (lp_gemm_nt_gelu_fwd): GELU of the stored output, same offsets
    unsigned short* out2_bf16;
};

// one workgroup's contribution to a fused BatchNorm sum (`idx` = 2 * segment offset + component * N + column)
__device__ __forceinline__ void stats_emit(const ConvEpilogue& ep, int idx, float t) { fx_add(&ep.stats_sums[idx], t); }

// Row pitches of the two operands and an optional batch of independent GEMMs sharing one launch (lp_gemm_nt: attention's
// per-(image, head) products).  A convolution is the special case ldx = channels, ldw = filter length, one batch.
struct GemmExt {
    int ldx, ldw;              // elements between consecutive rows of the gathered tensor / of the weight matrix
    int nh, tiles_per_z;       // batch index z = tile / tiles_per_z = (zb, zh), zh < nh
    unsigned x_zb, x_zh;       // byte strides of the gathered tensor per batch index
    unsigned w_zb, w_zh;       // byte strides of the weight matrix
    unsigned o_zb, o_zh;       // element strides of the output
    unsigned d_zb, d_zh, d_row;  // kModeAttn: element strides of the per-row vector D (batch indices, output row)
};

// kModeAttn = kModeFwd with the soft-max backward in the store pass; kModeInfer = kModeFwd whose store pass adds a residual and
// applies the ReLU (inference with folded BatchNorm: its own instantiation, so the training kernels' code is untouched)
enum { kModeFwd = 0, kModeDgrad = 1, kModeStem = 2, kModeAttn = 3, kModeInfer = 4 };

// One launch covers a sub-lattice of output pixels and of filter taps.  Ordinary launches use the full lattices; the data
// gradient of a stride-2 convolution is split into its 4 output-parity classes, each of which only sees the taps of matching
// parity (4 + 2 + 2 + 1 of the 9 taps of a 3x3, 1 + 0 + 0 + 0 for a 1x1): 4x fewer MFMAs than masking all taps.
struct Lattice {
    int h0, hstep, nh, w0, wstep, nw;  // output rows of this launch: y = h0 + hstep * iy, iy < nh (same for x)
    int r0, rstep, nr, s0, sstep, ns;  // filter taps of this launch:  r = r0 + rstep * ir, ir < nr (same for s)
};

// n / d for 0 <= n < 2^31 without the ~40-instruction integer division: q = (n * ceil(2^sh / d)) >> sh
struct FastDiv {
    unsigned mul;
    int sh;
    int d;
};

static FastDiv make_fastdiv(int d) {
    FastDiv f;
    f.d = d;
    int s = 0;
    while ((1LL << s) < d) ++s;
    f.sh = 31 + s;
    f.mul = (unsigned)((((unsigned long long)1 << f.sh) + (unsigned)d - 1) / (unsigned)d);
    return f;
}

__device__ __forceinline__ int fdiv(int n, const FastDiv& f) { return (int)(((unsigned long long)(unsigned)n * f.mul) >> f.sh); }

// XCD-aware bijective tile remap (cdna_hip_programming.md T1): block b runs on XCD b % 8; give each XCD a
// contiguous chunk of the tile space.
__device__ __forceinline__ int xcd_remap(int bid, int ntiles) {
    const int q = ntiles >> 3, r = ntiles & 7;
    const int xcd = bid & 7, local = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

// row index in the output tensor of row m of this launch's pixel sub-lattice (identity for full-lattice launches)
__device__ __forceinline__ int out_row(int m, const Lattice& lat, const FastDiv& div_img, const FastDiv& div_row, int full_h, int full_w) {
    if (lat.hstep == 1 && lat.wstep == 1) return m;
    const int b = fdiv(m, div_img), rem = m - b * div_img.d;
    const int iy = fdiv(rem, div_row), ix = rem - iy * div_row.d;
    return (b * full_h + lat.h0 + lat.hstep * iy) * full_w + lat.w0 + lat.wstep * ix;
}

// one K step (64 deep) of a wave's 64 x (NT*32) tile: the fragments of k-slice kk+1 are read from LDS while the MFMAs of
// slice kk issue, so the LDS latency is exposed once per K step instead of once per slice
template <int NT>
__device__ __forceinline__ void mma_kstep(const unsigned short* sA, const unsigned short* sB, int wm, int wn, int lane,
                                          f32x16 (&acc)[2][NT]) {
    const int r = lane & 31, g = lane >> 5;
    const unsigned short* pa = sA + (wm * 64 + r) * kLD + g * 8;
    const unsigned short* pb = sB + (wn * (NT * 32) + r) * kLD + g * 8;
    bf16x8 a[2][2], b[2][NT];
    auto fetch = [&](int kk, int set) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) a[set][mt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u16x8*>(pa + mt * 32 * kLD + kk * 16));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[set][nt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u16x8*>(pb + nt * 32 * kLD + kk * 16));
    };
    fetch(0, 0);
#pragma unroll
    for (int kk = 0; kk < kBK / 16; ++kk) {
        if (kk + 1 < kBK / 16) fetch(kk + 1, (kk + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);  // keep the next slice's reads ahead of this slice's MFMAs
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kk & 1][mt], b[kk & 1][nt], acc[mt][nt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

__device__ __forceinline__ u16x8 zero8() {
    u16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    return z;
}

__device__ __forceinline__ u16x8 load8(const unsigned short* p) { return *reinterpret_cast<const u16x8*>(p); }

// 16-B load of data that is dead after this read (the gradient addend, ReLU masks): non-temporal, so it does not displace the
// tensors the next kernels re-read from L2 / MALL
__device__ __forceinline__ u16x8 load8_stream(const unsigned short* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(p));
#else
    return *reinterpret_cast<const u16x8*>(p);
#endif
}

// two 8-byte halves (stem: NHWC4 pixels are only 8-byte aligned)
__device__ __forceinline__ u16x8 load4x2(const unsigned short* p0, bool ok0, const unsigned short* p1, bool ok1) {
    u16x4 lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0};
    if (ok0) lo = *reinterpret_cast<const u16x4*>(p0);
    if (ok1) hi = *reinterpret_cast<const u16x4*>(p1);
    u16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return v;
}

// ------------------------------------------------------------------------------------------------------------
// forward / data-gradient:  out[m][n] = sum_k A[m][k] * Wt[n][k]
//   kModeFwd   m = (b, ho, wo)  k = (r, s, ci)   A = x[b][ho*st - pad + r][wo*st - pad + s][ci]
//   kModeDgrad m = (b, hi, wi)  k = (r, s, co)   A = dy[b][(hi + pad - r)/st][(wi + pad - s)/st][co] where divisible
//   kModeStem  forward of the 7x7/2 stem on NHWC4 input, k = (r, s8, c4) padded to 256
// ------------------------------------------------------------------------------------------------------------
template <int BN, int MODE>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const unsigned short* __restrict__ X, const unsigned short* __restrict__ Wt,
                                                         unsigned x_bytes, unsigned w_bytes, ConvGeom g, Lattice lat, GemmExt gx,
                                                         FastDiv div_img, FastDiv div_row, int M, int N, int K, int tiles_n,
                                                         int ntiles, ConvEpilogue ep) {
    constexpr int NT = BN / 64;
    // one LDS block: [2][128][72] A + [2][BN][72] B operand tiles, reused by the epilogue as a [128][BN+4] fp32 tile
    __shared__ __attribute__((aligned(16))) unsigned short smem[2 * (kBM + BN) * kLD];
    unsigned short(*sA)[kBM * kLD] = reinterpret_cast<unsigned short(*)[kBM * kLD]>(smem);
    unsigned short(*sB)[BN * kLD] = reinterpret_cast<unsigned short(*)[BN * kLD]>(smem + 2 * kBM * kLD);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int kchunk = tid & 7, rbase = tid >> 3;  // 8 x 16-B chunks per 64-wide K row; 32 rows per pass

    // Workgroups are persistent: each walks the tile list with stride gridDim.x, and the operands of the NEXT tile's first K
    // step are already in flight (in registers) while the current tile's epilogue runs, so the store pass of one tile overlaps
    // the load latency of the next.
    //
    // Per-thread description of the 4 A rows this thread stages.  Everything that depends only on the row is hoisted
    // out of the K loop: `rowoff` is the BYTE offset of the row's origin in the gathered tensor and `vmask` has one
    // bit per filter tap saying whether that tap reads a real pixel (zero padding / stride-2 parity / row >= M
    // otherwise).  Operands are fetched with raw buffer loads: a tap costs one wave-uniform offset add and one OR per row
    // (an invalid tap gets the offset 0xffffffff, which the hardware range check turns into zeros), the weight rows use a
    // fixed per-lane offset plus a scalar K offset - no selects, no 64-bit address arithmetic, no divergent branches.
    const buf_rsrc rsrc_x = make_buf_rsrc(X, x_bytes), rsrc_w = make_buf_rsrc(Wt, w_bytes);
    int pb[4], py[4], px[4];
    bool pv[4];
    unsigned rowoff[4];
    unsigned vmask[4];
    const int rows_y = lat.nh, rows_x = lat.nw;
    const int full_h = (MODE == kModeDgrad) ? g.Hi : g.Ho;  // the output tensor's spatial size
    const int full_w = (MODE == kModeDgrad) ? g.Wi : g.Wo;
    const int ck = (MODE == kModeDgrad) ? g.Co : g.Ci;  // channels of the gathered tensor
    const int src_h = (MODE == kModeDgrad) ? g.Ho : g.Hi;
    const int src_w = (MODE == kModeDgrad) ? g.Wo : g.Wi;
    const bool halved = (MODE == kModeDgrad) && g.stride == 2;  // stride-2 dgrad: tap (r,s) -> pixel ((py-r)/2, (px-s)/2)
    const int KT = K / kBK;  // 0 for a parity class without taps: the epilogue then just writes addend / zeros
    const bool stream_a = tiles_n == 1 && lat.nr * lat.ns == 1;

    u16x8 ra[4], rb[BN / 32];
    int tir = 0, tis = 0, tc = 0;  // filter-tap lattice index and channel offset of the NEXT K step to load
    unsigned voff[4];              // byte offsets of the 4 A rows for the current tap (~0 where the tap is padding)
    unsigned wtap = 0;             // byte offset of the current tap inside a weight row
    unsigned wrow[BN / 32];        // byte offset of this lane's chunk in each weight row it stages
    int m0n = 0, n0n = 0;        // origin of the tile being set up / loaded
    unsigned zon = 0;            // ... and its batch's offset into the output (elements)
    unsigned zdn = 0;            // ... and into the row vector D (kModeAttn)

    auto setup = [&](int vt) {
        int tile = xcd_remap(vt, ntiles);
        const int z = tile / gx.tiles_per_z;
        tile -= z * gx.tiles_per_z;
        const int zb = z / gx.nh, zh = z - zb * gx.nh;
        const unsigned zx = zb * gx.x_zb + zh * gx.x_zh, zw = zb * gx.w_zb + zh * gx.w_zh;
        zon = zb * gx.o_zb + zh * gx.o_zh;
        if (MODE == kModeAttn) zdn = zb * gx.d_zb + zh * gx.d_zh;
        const int tm_ = tile / tiles_n;
        m0n = tm_ * kBM;
        n0n = (tile - tm_ * tiles_n) * BN;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0n + rbase + 32 * i;
            pv[i] = m < M;
            const int mm = pv[i] ? m : 0;
            const int b = fdiv(mm, div_img);
            const int rem = mm - b * rows_y * rows_x;
            const int iy = fdiv(rem, div_row);
            const int y = lat.h0 + lat.hstep * iy, xq = lat.w0 + lat.wstep * (rem - iy * rows_x);
            pb[i] = b;
            if (MODE == kModeDgrad) {
                py[i] = y + g.pad;
                px[i] = xq + g.pad;
            } else {
                py[i] = y * g.stride - g.pad;
                px[i] = xq * g.stride - g.pad;
            }
            const int oy = halved ? (py[i] >> 1) : py[i], ox = halved ? (px[i] >> 1) : px[i];
            rowoff[i] = (unsigned)(((b * src_h + oy) * src_w + ox) * gx.ldx + kchunk * 8) * 2u + zx;
            unsigned mask = 0;
            if (MODE != kModeStem && pv[i]) {
                for (int ir = 0; ir < lat.nr; ++ir)
                    for (int it = 0; it < lat.ns; ++it) {
                        const int r = lat.r0 + lat.rstep * ir, t = lat.s0 + lat.sstep * it;
                        bool ok;
                        if (MODE == kModeDgrad) {
                            const int th = py[i] - r, tw = px[i] - t;
                            if (halved) ok = th >= 0 && tw >= 0 && !((th | tw) & 1) && (th >> 1) < g.Ho && (tw >> 1) < g.Wo;
                            else ok = th >= 0 && th < g.Ho && tw >= 0 && tw < g.Wo;
                        } else {
                            const int sy = py[i] + r, sx = px[i] + t;
                            ok = sy >= 0 && sy < g.Hi && sx >= 0 && sx < g.Wi;
                        }
                        mask |= (ok ? 1u : 0u) << (ir * lat.ns + it);
                    }
            }
            vmask[i] = mask;
        }
#pragma unroll
        for (int i = 0; i < BN / 32; ++i) {
            int n = n0n + rbase + 32 * i;
            if (n >= N) n = N - 1;  // rows past N are loaded (never stored): keeps the loop free of predicates
            wrow[i] = (unsigned)(n * gx.ldw + kchunk * 8) * 2u + zw;  // row stride = the FULL filter
        }
        tir = tis = tc = 0;
    };

    // `with_b = false` (the data-gradient's prefetch across its register-hungry store pass) leaves the weight rows for
    // load_b_deferred(): they are L2-resident, so fetching them after the store pass costs little and frees 16 VGPRs there
    unsigned b_deferred = 0;
    auto load_step = [&](int kt, bool with_b = true) {
        if (MODE == kModeStem) {  // A: two 8-byte pixel halves per chunk (NHWC4); B: the padded [64][256] weight rows
            const int r = kt * 2 + (kchunk >> 2), s0 = (kchunk & 3) * 2;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int hi = py[i] + r, wi = px[i] + s0;
                const bool okr = pv[i] && r < g.R && hi >= 0 && hi < g.Hi;
                const size_t base = ((size_t)(pb[i] * g.Hi + hi) * g.Wi + wi) * 4;
                ra[i] = load4x2(X + base, okr && wi >= 0 && wi < g.Wi, X + base + 4, okr && s0 + 1 < g.S && wi + 1 >= 0 && wi + 1 < g.Wi);
            }
#pragma unroll
            for (int i = 0; i < BN / 32; ++i) rb[i] = buf_load16(rsrc_w, wrow[i], (unsigned)(kt * kBK) * 2u);
        } else {
            // A tap is visited for ck / 64 consecutive K steps.  Its per-row offsets are computed once, when it is entered
            // (wave-uniform offset: fwd walks +(r, s); dgrad walks -(r, s), halved for stride 2, valid taps only); the channel
            // offset inside the tap rides in the loads' scalar offset, which the range check ignores.
            if (tc == 0) {
                const int tr = lat.r0 + lat.rstep * tir, ts = lat.s0 + lat.sstep * tis;
                const int qr = halved ? (tr >> 1) : tr, qs = halved ? (ts >> 1) : ts;
                const unsigned tapoff_b = (unsigned)(((MODE == kModeDgrad) ? -(qr * src_w + qs) : (qr * src_w + qs)) * gx.ldx) * 2u;
                const int tap = tir * lat.ns + tis;
                wtap = (unsigned)((tr * g.S + ts) * ck) * 2u;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned ok = (vmask[i] >> tap) & 1u;
                    voff[i] = (rowoff[i] + tapoff_b) | (ok - 1u);  // ok - 1 = 0 (valid) or ~0 (-> zeros)
                }
            }
            const unsigned tcb = (unsigned)tc * 2u;
            if (stream_a) {  // single column of tiles and a single tap: every activation byte is fetched exactly once
#pragma unroll
                for (int i = 0; i < 4; ++i) ra[i] = buf_load16_nt(rsrc_x, voff[i], tcb);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) ra[i] = buf_load16(rsrc_x, voff[i], tcb);
            }
            if (with_b) {
#pragma unroll
                for (int i = 0; i < BN / 32; ++i) rb[i] = buf_load16(rsrc_w, wrow[i], wtap + tcb);  // weights [N][R*S*C], K-contiguous
            } else {
                b_deferred = wtap + tcb;
            }
            tc += kBK;
            if (tc >= ck) {
                tc = 0;
                if (++tis == lat.ns) {
                    tis = 0;
                    ++tir;
                }
            }
        }
    };
    auto load_b_deferred = [&]() {
#pragma unroll
        for (int i = 0; i < BN / 32; ++i) rb[i] = buf_load16(rsrc_w, wrow[i], b_deferred);
    };
    auto store_step = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u16x8*>(&sA[buf][(rbase + 32 * i) * kLD + kchunk * 8]) = ra[i];
#pragma unroll
        for (int i = 0; i < BN / 32; ++i) *reinterpret_cast<u16x8*>(&sB[buf][(rbase + 32 * i) * kLD + kchunk * 8]) = rb[i];
    };

    f32x16 acc[2][NT];

    // Fused BatchNorm sums: thread t < 2 BN owns (component t / BN, column t % BN) and ADDS UP its column's tile sums over the tiles this
    // persistent workgroup walks (a fixed order); they leave - stats_emit: fixed point, integer atomics - when the column block or the
    // BatchNorm segment changes and at the end: at most 512 workgroups x 2 BN values per launch however many tiles there are
    float st_acc = 0.f;
    int st_n0 = -1, st_seg_off = 0;
    auto stats_flush = [&]() {
        if (st_n0 >= 0 && tid < 2 * BN) {
            const int comp = tid / BN, cl = tid % BN;
            if (st_n0 + cl < N) stats_emit(ep, 2 * st_seg_off + comp * N + st_n0 + cl, st_acc);
        }
        st_acc = 0.f;
    };

    // ---- epilogue of the tile at (m0, n0): D reg e of lane l is row (e&3) + 8*(e>>2) + 4*(l>>5), column l&31 of its 32x32 tile
    auto epilogue = [&](const int m0, const int n0, const unsigned zo, const unsigned zd = 0u) {
        const int col = lane & 31, rg = lane >> 5;
        if (ep.out_f32 == nullptr && (ep.n_store & 7) == 0 && (ep.ldo & 7) == 0) {
            // bf16 output: stage the fp32 tile in LDS (the operand buffers are free after the last barrier), then every lane
            // moves 16 B (8 channels) per store so a 128-column row leaves as two full 128-B lines; the addend (gradient
            // accumulation) is read the same way and added in fp32 before the single rounding to bf16
            constexpr int LDO = BN + 4;
            float* so = reinterpret_cast<float*>(smem);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int nl = wn * (NT * 32) + nt * 32 + col;
                    const float bias = (ep.bias && n0 + nl < N) ? ep.bias[n0 + nl] : 0.f;
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        so[(wm * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * rg) * LDO + nl] = acc[mt][nt][e] + bias;
                }
            __syncthreads();
            constexpr int CPR = BN / 8;          // 16-B chunks per tile row
            constexpr int RPP = 256 / CPR;       // rows per pass
            const int cc = tid % CPR, r0 = tid / CPR;
            const int n = n0 + cc * 8;
            const bool want_stats = ep.stats_sums != nullptr;
            // only the data-gradient has tensors to read back in its store pass (addend, pre-normalisation tensor, activation)
            constexpr bool kReads = (MODE == kModeDgrad || MODE == kModeInfer);  // the addend (gradient accumulation / residual)
            constexpr bool kBwd = (MODE == kModeDgrad);  // pre-normalisation tensor and ReLU masks: data gradient only
            constexpr bool kAttn = (MODE == kModeAttn);
            float s0[8], s1[8], mu[8], sc[8], be[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) s0[q] = s1[q] = mu[q] = sc[q] = be[q] = 0.f;
            // BatchNorm segment of this tile (workgroup-uniform)
            const int seg_off = (ep.seg_images > 0 && m0 >= ep.seg_images * rows_y * rows_x) ? N : 0;
            if (n < ep.n_store) {
                if (kBwd && ep.bn_z) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        mu[q] = ep.bn_mean[seg_off + n + q];
                        if (ep.mask_from_z) {
                            sc[q] = ep.bn_invstd[seg_off + n + q] * ep.bn_gamma[n + q];
                            be[q] = ep.bn_beta[n + q];
                        }
                    }
                }
                // The global reads of the store pass are issued in batches of 4 row passes (16 B per lane each), so a tile keeps
                // up to 48 KB of epilogue traffic in flight; one pass at a time left the gradient kernels of the 1x1 layers
                // latency-bound at ~3 TB/s.
                constexpr int NP = kBM / RPP, HB = NP < 4 ? NP : 4;
#pragma unroll
                for (int i0 = 0; i0 < NP; i0 += HB) {
                    unsigned off[HB];
                    bool rv[HB];
                    u16x8 la[HB], lz[HB], lm[HB];
                    unsigned lb[HB];
                    float ld_[HB];
#pragma unroll
                    for (int i = 0; i < HB; ++i) {
                        const int m = m0 + r0 + (i0 + i) * RPP;
                        rv[i] = m < M;
                        off[i] = (unsigned)out_row(rv[i] ? m : 0, lat, div_img, div_row, full_h, full_w) * (unsigned)ep.ldo + (unsigned)n + zo;
                    }
                    if (kAttn) {  // the probabilities (re-read by the dV product: ordinary loads) and this row's D
#pragma unroll
                        for (int i = 0; i < HB; ++i) {
                            la[i] = load8(ep.attn_p + off[i]);
                            ld_[i] = ep.attn_d[zd + (unsigned)(rv[i] ? m0 + r0 + (i0 + i) * RPP : 0) * gx.d_row];
                        }
                    }
                    if (kReads && ep.addend && ep.addend_half) {   // the addend on the half-resolution grid, at the even pixels only (full lattice: host-checked)
#pragma unroll
                        for (int i = 0; i < HB; ++i) {
                            const int mm = rv[i] ? m0 + r0 + (i0 + i) * RPP : 0;
                            const int bi = fdiv(mm, div_img), rem = mm - bi * div_img.d;
                            const int yy = fdiv(rem, div_row), xx = rem - yy * div_row.d;
                            const bool on = rv[i] && !((yy | xx) & 1);
                            const unsigned oa = (unsigned)((bi * ((full_h + 1) >> 1) + (yy >> 1)) * ((full_w + 1) >> 1) + (xx >> 1)) * (unsigned)ep.ldo + (unsigned)n;
                            la[i] = on ? load8_stream(ep.addend + oa) : zero8();
                        }
                    } else if (kReads && ep.addend) {
#pragma unroll
                        for (int i = 0; i < HB; ++i) la[i] = load8_stream(ep.addend + off[i]);
                    }
                    if (kBwd && ep.bn_z) {
#pragma unroll
                        for (int i = 0; i < HB; ++i) lz[i] = load8(ep.bn_z + off[i]);
                    }
                    if (kBwd && ep.relu_mask) {
#pragma unroll
                        for (int i = 0; i < HB; ++i) lm[i] = load8_stream(ep.relu_mask + off[i]);
                    }
                    if (kBwd && ep.relu_bits) {
#pragma unroll
                        for (int i = 0; i < HB; ++i) lb[i] = ep.relu_bits[off[i] >> 3];
                    }
#pragma unroll
                    for (int i = 0; i < HB; ++i) {
                        const int rl = r0 + (i0 + i) * RPP;
                        if (rv[i]) {
                            const f32x4 lo = *reinterpret_cast<const f32x4*>(so + rl * LDO + cc * 8);
                            const f32x4 hi = *reinterpret_cast<const f32x4*>(so + rl * LDO + cc * 8 + 4);
                            float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                            if (kAttn) {
#pragma unroll
                                for (int q = 0; q < 8; ++q)
                                    v[q] = (n + q < N) ? ep.attn_scale * bf16_to_f32(la[i][q]) * (v[q] - ld_[i]) : 0.f;
                            }
                            if (kReads && ep.addend) {
#pragma unroll
                                for (int q = 0; q < 8; ++q) v[q] += bf16_to_f32(la[i][q]);
                            }
                            float zc[8];  // z - mean
                            if (kBwd && ep.bn_z) {
#pragma unroll
                                for (int q = 0; q < 8; ++q) zc[q] = bf16_to_f32(lz[i][q]) - mu[q];
                                if (ep.mask_from_z) {
                                    // lp_bn_apply stored bf16(max(o, 0)) with this same o; that is > 0 exactly when o exceeds
                                    // half the smallest bf16 subnormal (round-to-nearest-even)
#pragma unroll
                                    for (int q = 0; q < 8; ++q)
                                        if (!(fmaf(zc[q], sc[q], be[q]) > 0x1p-134f)) v[q] = 0.f;
                                }
                            }
                            if (kBwd && ep.relu_mask) {
#pragma unroll
                                for (int q = 0; q < 8; ++q)
                                    if (!bf16_positive(lm[i][q])) v[q] = 0.f;
                            }
                            if (kBwd && ep.relu_bits) {
#pragma unroll
                                for (int q = 0; q < 8; ++q)
                                    if (!((lb[i] >> q) & 1u)) v[q] = 0.f;
                            }
                            if (MODE == kModeInfer && ep.relu_fwd) {
#pragma unroll
                                for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], 0.f);
                            }
                            const u16x8 w = pack_bf16x8(v);
                            *reinterpret_cast<u16x8*>(ep.out_bf16 + off[i]) = w;
                            if (want_stats) {
#pragma unroll
                                for (int q = 0; q < 8; ++q) {
                                    const float vr = bf16_to_f32(w[q]);
                                    s0[q] += vr;
                                    s1[q] = fmaf(vr, (kBwd && ep.bn_z) ? zc[q] : vr, s1[q]);  // backward: x invstd below
                                }
                            }
                        }
                    }
                }
                if (kBwd && ep.bn_z && want_stats) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) s1[q] *= ep.bn_invstd[seg_off + n + q];
                }
            }
            if (want_stats) {
                // column sums of this tile: [2][RPP][BN] partials through LDS, then one thread per (component, column)
                __syncthreads();  // every lane is done reading the fp32 tile
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    so[(0 * RPP + r0) * BN + cc * 8 + q] = s0[q];
                    so[(1 * RPP + r0) * BN + cc * 8 + q] = s1[q];
                }
                __syncthreads();
                if (ep.stats_sums != nullptr && (n0 != st_n0 || seg_off != st_seg_off)) {  // (workgroup-uniform)
                    stats_flush();
                    st_n0 = n0;
                    st_seg_off = seg_off;
                }
                if (tid < 2 * BN) {
                    const int comp = tid / BN, cl = tid % BN;
                    float t = 0.f;
#pragma unroll
                    for (int r = 0; r < RPP; ++r) t += so[(comp * RPP + r) * BN + cl];
                    if (n0 + cl < N) st_acc += t;
                }
            }
            return;
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = n0 + wn * (NT * 32) + nt * 32 + col;
                if (n >= ep.n_store) continue;
                const float bias = ep.bias ? ep.bias[n] : 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = m0 + wm * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * rg;
                    if (m < M) {
                        float v = acc[mt][nt][e] + bias;
                        const size_t o = (size_t)out_row(m, lat, div_img, div_row, full_h, full_w) * ep.ldo + n + zo;
                        if (ep.addend) v += bf16_to_f32(ep.addend[o]);
                        if (ep.relu_mask) {
                            if (!bf16_positive(ep.relu_mask[o])) v = 0.f;
                        }
                        if (MODE == kModeInfer && ep.relu_fwd) v = fmaxf(v, 0.f);
                        if (ep.out_bf16) ep.out_bf16[o] = f32_to_bf16(v);
                        if (ep.out_f32) ep.out_f32[o] = v;
                    }
                }
            }
    };

    int vt = blockIdx.x;
    if (KT == 0) {
        // a parity class no filter tap reaches: the "gradient" is the addend (or zero), masked and reduced like any other tile.
        // Kept apart so that the main walk below issues and consumes its prefetch unconditionally.
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;
        for (; vt < ntiles; vt += gridDim.x) {
            const int tile = xcd_remap(vt, ntiles);  // (never batched: only the stride-2 data gradient has empty classes)
            const int tm_ = tile / tiles_n;
            epilogue(tm_ * kBM, (tile - tm_ * tiles_n) * BN, 0u);
            __syncthreads();
        }
        if (ep.stats_sums != nullptr) stats_flush();
        return;
    }
    setup(vt);
    load_step(0);
    for (;;) {
        const int m0 = m0n, n0 = n0n;
        const unsigned zo = zon;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;
        store_step(0);
        __syncthreads();
        for (int kt = 0; kt < KT; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < KT) load_step(kt + 1);  // global loads in flight under the MFMAs
            mma_kstep<NT>(sA[cur], sB[cur], wm, wn, lane, acc);
            if (kt + 1 < KT) store_step(cur ^ 1);
            __syncthreads();
        }
        vt += gridDim.x;
        const unsigned zd = zdn;
        if (vt >= ntiles) {
            epilogue(m0, n0, zo, zd);
            if (ep.stats_sums != nullptr) stats_flush();
            break;
        }
        // next tile's row descriptors and first operands: in flight while this tile is stored
        setup(vt);
        load_step(0, MODE != kModeDgrad);
        epilogue(m0, n0, zo, zd);
        if (MODE == kModeDgrad) load_b_deferred();
        __syncthreads();  // the epilogue's LDS tile is dead before the next tile's operands land in it
    }
}

}  // namespace lp

#include "conv_pipe.h"
#include "conv_res2d.h"

namespace lp {

// Direct, batched use of the weight-gradient kernel as a TN GEMM (lp_gemm_tn: attention's dV = P^T dO and dK = dS^T Q):
// out[z][j][n] = sum_m x[z][m][j] * y[z][m][n], one "slice" per batch z, result stored as bf16 instead of going to the split-K
// workspace.  A null `out` means the ordinary weight-gradient.
struct TnExt {
    unsigned short* out;        // [z][j][n] bf16, pitch ldo; nullptr = weight-gradient mode
    int ldx, ldy, ldo;          // row pitches (elements) of x, y, out
    int nh;                     // batch index z = (zb, zh), zh < nh
    unsigned x_zb, x_zh;        // byte strides of x / y per batch index
    unsigned y_zb, y_zh;
    unsigned o_zb, o_zh;        // element strides of out
    // weight-gradient mode only: += the column sums of dy (a Linear / ConvTranspose layer's bias gradient), taken by the tiles of
    // the first row block while they stage dy anyway (fp32 atomics: one per column, slice and launch); nullptr = off
    float* col_sums;
};

// ------------------------------------------------------------------------------------------------------------
// weight gradient:  dW[n][j] += sum_m xg[m][j] * dy[m][n],  j = (r, s, ci),  m = (b, ho, wo) split over blockIdx.y
// computed as D[j][n] (rows j from the gathered activations, columns n from dy), both transposed into LDS.
// Each pixel slice writes its fp32 partial tile to a workspace in accumulator order (fully coalesced 256-B stores);
// wgrad_reduce_kernel then sums the slices in a fixed order and adds the result into dW - deterministic, and ~20x
// cheaper than fp32 atomics (measured: 17 M atomics per launch cost 450 us, the same bytes as plain stores ~20 us).
// ------------------------------------------------------------------------------------------------------------
#ifndef LP_WGRAD_DEEP
#define LP_WGRAD_DEEP 1
#endif
constexpr bool kWgradDeep = LP_WGRAD_DEEP != 0;   // (build-time A/B switch)

template <int BN, bool STEM, bool CS = false>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const unsigned short* __restrict__ X, const unsigned short* __restrict__ DY,
                                                         unsigned x_bytes, unsigned dy_bytes, ConvGeom g, int M, int Kw, int tiles,
                                                         int tiles_n, int m_per_split, FastDiv div_hw, FastDiv div_wo,
                                                         float* __restrict__ ws, TnExt tn) {
    constexpr int NT = BN / 64;
    constexpr int RB = BN / 32;  // dy rows (pixels) per thread per K step: 4 (BN=128) or 2 (BN=64)
    // Operand tiles stay in their memory orientation, [pixel][channel] (K = pixel is the ROW index): the 16-B chunks go to LDS
    // with plain ds_write_b128 and the MFMA fragments (8 consecutive pixels of one channel per lane) come out of
    // ds_read_b64_tr_b16.  Row stride = tile width + 64 B, so the 4 pixel rows of a transpose read sit in distinct bank quarters.
    constexpr int LDA = kBM + 32, LDB = BN + 32;
    __shared__ __attribute__((aligned(16))) unsigned short sA[2][kBK * LDA];
    __shared__ __attribute__((aligned(16))) unsigned short sB[2][kBK * LDB];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // 1-D grid over (pixel slice, tile) with the slice slow: the XCD remap hands each XCD a contiguous range of slices with ALL
    // their tiles, so the tiles that re-read one slice of x / dy share an L2 instead of fetching it once per XCD
    const int work = xcd_remap(blockIdx.x, gridDim.x);
    const int slice = work / tiles, tile = work - slice * tiles;
    const int j0 = (tile / tiles_n) * kBM, n0 = (tile % tiles_n) * BN;
    const bool direct = tn.out != nullptr;          // TN-GEMM mode: slice = batch index, rows [0, M) of that batch
    const int m_begin = direct ? 0 : slice * m_per_split;
    const int m_end = direct ? M : min(M, m_begin + m_per_split);
    unsigned zx = 0, zy = 0, zo = 0;
    if (direct) {
        const int zb = slice / tn.nh, zh = slice - zb * tn.nh;
        zx = zb * tn.x_zb + zh * tn.x_zh;
        zy = zb * tn.y_zb + zh * tn.y_zh;
        zo = zb * tn.o_zb + zh * tn.o_zh;
    }

    // A operand (gathered activations): thread -> 8 consecutive j (one 16-B chunk) of 4 consecutive pixels; 16 consecutive
    // lanes cover the 256-B tile row of one pixel
    const int jc = tid & 15, pgA = tid >> 4;
    const int j = j0 + jc * 8;
    const bool jv = j < Kw;
    int tr = 0, ts = 0, tcn = 0;
    if (STEM) {  // j = (r, s8, c4): 32 per r; chunk = pixels (s0, s0+1) x 4 channels
        tr = j >> 5;
        ts = (j & 31) >> 2;
    } else {
        const int tap = (jv ? j : 0) / g.Ci;
        tcn = (jv ? j : 0) - tap * g.Ci;
        tr = tap / g.S;
        ts = tap - tr * g.S;
    }
    // B operand (dy): thread -> 8 consecutive n of RB consecutive pixels
    constexpr int NCH = BN / 8;
    const int nc = tid % NCH, pgB = tid / NCH;
    const int nB = n0 + nc * 8;
    const bool nv = nB < g.Co;

    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;

    u16x8 ra[4], rb[RB];     // the K step in flight ...
    u16x8 ra2[4], rb2[RB];   // ... and, on the two-deep path below, the one after it
    const int hw = g.Ho * g.Wo;
    const bool want_cs = CS && j0 == 0;  // workgroup-uniform (CS: the instantiation that also takes dy's column sums)
    float cs[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) cs[q] = 0.f;

    // Fast addressing for stride-1 "same" convolutions (every 1x1 and 3x3 of the trunk except the three stride-2 blocks): the
    // input pixel of output pixel m under tap (r, s) is m + const, so a lane's byte offsets are fixed (raw buffer loads: the K
    // step rides in the scalar offset, padding taps get the offset ~0 and read zeros) and the only per-step arithmetic is the
    // (row, column) walk that decides padding.  Everything else (stride 2, the stem, a ragged last step) takes the generic path.
    const bool fast = direct || (!STEM && g.stride == 1 && g.Hi == g.Ho && g.Wi == g.Wo && g.Wo >= 8 && g.Ho >= 8);
    const bool nomask = fast && g.R == 1 && g.S == 1 && g.pad == 0;
    const buf_rsrc rsrc_x = make_buf_rsrc(X, x_bytes), rsrc_dy = make_buf_rsrc(DY, dy_bytes);
    const int pitch_x = direct ? tn.ldx : g.Ci, pitch_y = direct ? tn.ldy : g.Co;
    const unsigned ci2 = (unsigned)pitch_x * 2u;
    const int qw = 64 / g.Wo, rw = 64 - qw * g.Wo;
    int fwo = 0, fho = 0, lo_h = 0, span_h = 0, lo_w = 0, span_w = 0;
    unsigned voffA = ~0u, voffB[RB];
    if (fast) {
        const int p0 = m_begin + pgA * 4;
        const int pc = p0 < M ? p0 : 0;
        const int rem = pc - fdiv(pc, div_hw) * hw;
        fho = fdiv(rem, div_wo);
        fwo = rem - fho * g.Wo;
        if (jv) voffA = (unsigned)((p0 + (tr - g.pad) * g.Wi + (ts - g.pad)) * pitch_x + tcn) * 2u + zx;
        lo_h = max(0, g.pad - tr);
        span_h = min(g.Ho, g.Hi + g.pad - tr) - lo_h;
        lo_w = max(0, g.pad - ts);
        span_w = min(g.Wo, g.Wi + g.pad - ts) - lo_w;
#pragma unroll
        for (int i = 0; i < RB; ++i) voffB[i] = nv ? (unsigned)((m_begin + pgB * RB + i) * pitch_y + nB) * 2u + zy : ~0u;
    }
    const unsigned jinv = jv ? 0u : ~0u;

    // (NM = nomask as a compile-time constant: with both forms in one loop body the two register sets of the two-deep path below
    // meet in phi copies and every store waits for all loads)
    auto load_fast = [&](auto nm_c, int mk, u16x8 (&ra)[4], u16x8 (&rb)[RB]) {  // whole K step inside [m_begin, m_end)
        constexpr bool NM = decltype(nm_c)::value;
        const unsigned step = (unsigned)(mk - m_begin);
        const unsigned soffA = step * ci2, soffB = step * (unsigned)pitch_y * 2u;
        if (NM) {
            // (TN-GEMM mode takes its ragged last step here too: rows >= m_end get the offset ~0 and read zeros)
            const bool ragged = mk + kBK > m_end;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned rinv = (ragged && mk + pgA * 4 + i >= m_end) ? ~0u : 0u;
                ra[i] = buf_load16(rsrc_x, (voffA + i * ci2) | jinv | rinv, soffA);
            }
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const unsigned rinv = (ragged && mk + pgB * RB + i >= m_end) ? ~0u : 0u;
                rb[i] = buf_load16(rsrc_dy, voffB[i] | rinv, soffB);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int wo = fwo + i, ho = fho;
                if (wo >= g.Wo) {
                    wo -= g.Wo;
                    ho = (ho + 1 == g.Ho) ? 0 : ho + 1;
                }
                const bool ok = (unsigned)(ho - lo_h) < (unsigned)span_h && (unsigned)(wo - lo_w) < (unsigned)span_w;
                // the range check sees only the vector offset, and voffA alone may be "negative" for the taps above / left of
                // the first pixel: the step offset has to be added into it, not ride in the scalar offset
                ra[i] = buf_load16(rsrc_x, (voffA + soffA + i * ci2) | (ok ? 0u : ~0u) | jinv, 0u);
            }
            fwo += rw;
            fho += qw;
            if (fwo >= g.Wo) {
                fwo -= g.Wo;
                ++fho;
            }
            if (fho >= g.Ho) fho -= g.Ho;
            if (fho >= g.Ho) fho -= g.Ho;
#pragma unroll
            for (int i = 0; i < RB; ++i) rb[i] = buf_load16(rsrc_dy, voffB[i], soffB);
        }
    };

    auto load_generic = [&](int mk, u16x8 (&ra)[4], u16x8 (&rb)[RB]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = mk + pgA * 4 + i;
            bool ok = jv && m < m_end;
            const int mm = ok ? m : 0;
            const int b = fdiv(mm, div_hw), rem = mm - b * hw;
            const int ho = fdiv(rem, div_wo), wo = rem - ho * g.Wo;
            const int hi = ho * g.stride - g.pad + tr, wi = wo * g.stride - g.pad + ts;
            if (STEM) {
                const bool okr = ok && tr < g.R && hi >= 0 && hi < g.Hi;
                const size_t base = ((size_t)(b * g.Hi + hi) * g.Wi + wi) * 4;
                ra[i] = load4x2(X + base, okr && wi >= 0 && wi < g.Wi, X + base + 4, okr && ts + 1 < g.S && wi + 1 >= 0 && wi + 1 < g.Wi);
            } else {
                ok = ok && hi >= 0 && hi < g.Hi && wi >= 0 && wi < g.Wi;
                ra[i] = ok ? load8(X + ((size_t)(b * g.Hi + hi) * g.Wi + wi) * g.Ci + tcn) : zero8();
            }
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int m = mk + pgB * RB + i;
            rb[i] = (nv && m < m_end) ? load8(DY + (size_t)m * g.Co + nB) : zero8();
        }
    };
    auto load_step = [&](int mk) {
        if (fast && (direct || mk + kBK <= m_end)) {
            if (nomask) load_fast(std::true_type{}, mk, ra, rb);
            else load_fast(std::false_type{}, mk, ra, rb);
        } else load_generic(mk, ra, rb);
    };
    auto store_step = [&](int buf, const u16x8 (&ra)[4], const u16x8 (&rb)[RB]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u16x8*>(&sA[buf][(pgA * 4 + i) * LDA + jc * 8]) = ra[i];
#pragma unroll
        for (int i = 0; i < RB; ++i) *reinterpret_cast<u16x8*>(&sB[buf][(pgB * RB + i) * LDB + nc * 8]) = rb[i];
        if (want_cs) {  // rows past the slice and columns past Co were loaded as zeros
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int q = 0; q < 8; ++q) cs[q] += bf16_to_f32(rb[i][q]);
        }
    };

    // fragment of 32 channels x 16 pixels (k-slice kk) out of a [pixel][channel] tile: lane l of 16-lane group q = l / 16
    // supplies pixel row 8 * (q / 2) + (l % 16) / 4 (+4 for the second read), channels 16 * (q % 2) + 4 * (l % 4) .. +3, and
    // receives channel 16 * (q % 2) + l % 16 = l % 32, pixels 8 * (l / 32) .. +7: exactly the MFMA operand layout
    const int fq = lane >> 4, fi = lane & 15;
    const int frow = (fq >> 1) * 8 + (fi >> 2), fcol = (fq & 1) * 16 + (fi & 3) * 4;
    auto frag = [&](const unsigned short* tile_, int ld, int ch0, int kk) -> bf16x8 {
        const unsigned short* p = tile_ + (kk * 16 + frow) * ld + ch0 + fcol;
        const s16x4_t lo = lds_read_tr16(p), hi = lds_read_tr16(p + 4 * ld);
        typedef __attribute__((ext_vector_type(8))) short s16x8_t;
        const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };
    auto mma_step = [&](int buf) {
        bf16x8 a[2][2], b[2][NT];
        auto fetch = [&](int kk, int set) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) a[set][mt] = frag(sA[buf], LDA, wm * 64 + mt * 32, kk);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b[set][nt] = frag(sB[buf], LDB, wn * (NT * 32) + nt * 32, kk);
        };
        fetch(0, 0);
#pragma unroll
        for (int kk = 0; kk < kBK / 16; ++kk) {
            if (kk + 1 < kBK / 16) fetch(kk + 1, (kk + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);  // keep the next slice's reads ahead of this slice's MFMAs
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kk & 1][mt], b[kk & 1][nt], acc[mt][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    const int KT = (m_end - m_begin + kBK - 1) / kBK;
    // Two K steps in flight (register sets ra / ra2) where every step takes the fast loads: the HBM-bound shapes (the 1x1 layers of
    // layer1 / layer2: 25 KB per step against 512 MFMA cycles) are bound by the bytes a CU keeps in flight, and with one step per
    // workgroup that is 2 x 25 KB.  The loop is unrolled by two so that each set is a fixed array (a store of one set waits with a counted
    // vmcnt while the other set's loads stay in flight).
    const bool deep = kWgradDeep && fast && KT >= 2 && (direct || (m_end - m_begin) % kBK == 0);
    auto deep_loop = [&](auto nm) {
        load_fast(nm, m_begin, ra, rb);
        load_fast(nm, m_begin + kBK, ra2, rb2);
        store_step(0, ra, rb);
        __syncthreads();
        int kt = 0;
        for (; kt + 3 < KT; kt += 2) {   // buffer 0 holds step kt, set 2 step kt + 1
            load_fast(nm, m_begin + (kt + 2) * kBK, ra, rb);
            mma_step(0);
            store_step(1, ra2, rb2);
            __syncthreads();
            load_fast(nm, m_begin + (kt + 3) * kBK, ra2, rb2);
            mma_step(1);
            store_step(0, ra, rb);
            __syncthreads();
        }
        const bool three = kt + 2 < KT;   // two or three steps left
        if (three) load_fast(nm, m_begin + (kt + 2) * kBK, ra, rb);
        mma_step(0);
        store_step(1, ra2, rb2);
        __syncthreads();
        mma_step(1);
        if (three) {
            store_step(0, ra, rb);
            __syncthreads();
            mma_step(0);
        }
        __syncthreads();
    };
    if (deep) {
        if (nomask) deep_loop(std::true_type{});
        else deep_loop(std::false_type{});
    } else if (KT > 0) {
        load_step(m_begin);
        store_step(0, ra, rb);
        __syncthreads();
        for (int kt = 0; kt < KT; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < KT) load_step(m_begin + (kt + 1) * kBK);
            mma_step(cur);
            if (kt + 1 < KT) store_step(cur ^ 1, ra, rb);
            __syncthreads();
        }
    }

    if (want_cs) {  // fold the 256 / NCH row groups, then one atomic per column (the operand tiles are dead after the last barrier)
        float* red = reinterpret_cast<float*>(&sA[0][0]);
#pragma unroll
        for (int q = 0; q < 8; ++q) red[pgB * BN + nc * 8 + q] = cs[q];
        __syncthreads();
        if (tid < BN && n0 + tid < g.Co) {
            float t = 0.f;
            for (int r = 0; r < 256 / NCH; ++r) t += red[r * BN + tid];
            atomicAdd(&tn.col_sums[n0 + tid], t);
        }
    }
    if (direct) {  // out[z][j][n] = bf16(acc): reg e of lane l is row j = (e&3) + 8*(e>>2) + 4*(l>>5), column n = l & 31 of its block
        const int col = lane & 31, rg = lane >> 5;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = n0 + wn * (NT * 32) + nt * 32 + col;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int jr = j0 + wm * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * rg;
                    if (jr < Kw && n < g.Co) tn.out[zo + (size_t)jr * tn.ldo + n] = f32_to_bf16(acc[mt][nt][e]);
                }
            }
        return;
    }
    // partial tile -> workspace[slice][tile][wave][mt][nt][e][lane]
    float* dst = ws + ((((size_t)slice * tiles + tile) * 4 + wave) * (2 * NT * 16)) * 64 + lane;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) dst[((mt * NT + nt) * 16 + e) * 64] = acc[mt][nt][e];
}

// dW[n][j] += sum over slices of the partial tiles (one thread per accumulator element; slices summed in order)
template <int BN>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, int slices, int tiles, int tiles_n, int Kw,
                                                           int Co, float* __restrict__ dW) {
    constexpr int NT = BN / 64;
    constexpr int PER_TILE = 4 * 2 * NT * 16 * 64;
    const size_t total = (size_t)tiles * PER_TILE;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        float sum = 0.f;
        for (int sl = 0; sl < slices; ++sl) sum += ws[(size_t)sl * total + i];
        const int lane = (int)(i & 63);
        size_t t = i >> 6;
        const int e = (int)(t & 15);
        t >>= 4;
        const int nt = (int)(t % NT);
        t /= NT;
        const int mt = (int)(t & 1);
        t >>= 1;
        const int wave = (int)(t & 3);
        const int tile = (int)(t >> 2);
        const int wm = wave >> 1, wn = wave & 1;
        const int j = (tile / tiles_n) * kBM + wm * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        const int n = (tile % tiles_n) * BN + wn * (NT * 32) + nt * 32 + (lane & 31);
        if (j < Kw && n < Co) dW[(size_t)n * Kw + j] += sum;
    }
}

// same reduction for layers with MANY slices and few tiles (layer1: 768 slices of one tile): one workgroup per 64
// consecutive accumulator elements, its 4 waves stride over the slices, then combine through LDS (fixed order)
template <int BN>
__global__ __launch_bounds__(256) void wgrad_reduce_sliced_kernel(const float* __restrict__ ws, int slices, int tiles, int tiles_n,
                                                                  int Kw, int Co, float* __restrict__ dW) {
    constexpr int NT = BN / 64;
    constexpr int PER_TILE = 4 * 2 * NT * 16 * 64;
    __shared__ float part[4][64];
    const size_t total = (size_t)tiles * PER_TILE;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t i = (size_t)blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int sl = w;
    for (; sl + 12 < slices; sl += 16) {  // 4 independent loads in flight per lane
        s0 += ws[(size_t)sl * total + i];
        s1 += ws[(size_t)(sl + 4) * total + i];
        s2 += ws[(size_t)(sl + 8) * total + i];
        s3 += ws[(size_t)(sl + 12) * total + i];
    }
    for (; sl < slices; sl += 4) s0 += ws[(size_t)sl * total + i];
    part[w][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (w == 0) {
        const float sum = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
        size_t t = i >> 6;
        const int e = (int)(t & 15);
        t >>= 4;
        const int nt = (int)(t % NT);
        t /= NT;
        const int mt = (int)(t & 1);
        t >>= 1;
        const int wave = (int)(t & 3);
        const int tile = (int)(t >> 2);
        const int wm = wave >> 1, wn = wave & 1;
        const int j = (tile / tiles_n) * kBM + wm * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        const int n = (tile % tiles_n) * BN + wn * (NT * 32) + nt * 32 + (lane & 31);
        if (j < Kw && n < Co) dW[(size_t)n * Kw + j] += sum;
    }
}

template <int BN>
static void launch_wgrad_reduce(const float* ws, int slices, int tiles, int tiles_n, int Kw, int Co, float* dw, hipStream_t st) {
    constexpr int PER_TILE = 4 * 2 * (BN / 64) * 16 * 64;
    if (slices >= 16) {
        hipLaunchKernelGGL((wgrad_reduce_sliced_kernel<BN>), dim3(tiles * (PER_TILE / 64)), dim3(256), 0, st, ws, slices, tiles, tiles_n, Kw,
                           Co, dw);
    } else {
        const int blocks = tiles * (PER_TILE / 256);
        hipLaunchKernelGGL((wgrad_reduce_kernel<BN>), dim3(blocks < 4096 ? blocks : 4096), dim3(256), 0, st, ws, slices, tiles, tiles_n, Kw,
                           Co, dw);
    }
}

// rows of BatchNorm segment 0 in a launch whose images have `rows_per_image` output rows each; -1 if a 128-row tile would straddle
// the boundary (unsupported), `M` itself for a single segment
static long long seg_split_rows(int seg_images, int B, long long rows_per_image, long long M) {
    if (seg_images <= 0) return M;
    if (seg_images >= B) return -1;
    const long long r = (long long)seg_images * rows_per_image;
    return (r % kBM == 0) ? r : -1;
}

// what every fused entry point does with its lp_bn_fuse: the epilogue's targets
static void bn_fuse_begin(ConvEpilogue& ep, const lp_bn_fuse* bn) {
    ep.stats_sums = bn->sums;
    ep.seg_images = bn->seg_images;
}

// which kernel family the most recent convolution entry point of this thread launched (lp_conv_last_kernel: the bench labels its
// per-launch timings with the kernel that actually ran; tests assert the path they mean to exercise)
static thread_local int g_last_conv_kernel = LP_CONV_KERNEL_IGEMM;

// Persistent launch of conv_igemm_kernel: at most 2 workgroups per CU (the LDS limit) x 256 CUs, a multiple of 8 so the
// stride walk keeps every workgroup on its XCD's tile range.  LP_CONV_MAX_WGS overrides the cap (tests use it to force several
// tiles per workgroup on small problems).
static int igemm_max_wgs() {
    if (lp_switches().conv_max_wgs > 0) return lp_switches().conv_max_wgs;
    static int v = [] {
        int dev = 0, cus = 0;  // 2 workgroups per CU are resident (LDS-bound): 512 on a full MI355X, fewer on a partition
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        return 2 * cus;
    }();
    return v;
}

// `gemm` (lp_gemm_nt only): operand pitches, batch strides and explicit operand sizes; nz = number of batched GEMMs
template <int BN, int MODE>
static int launch_igemm(const void* x, const void* w, const ConvGeom& g, const Lattice& lat, int M, int N, int K, const ConvEpilogue& ep,
                         hipStream_t st, const GemmExt* gemm = nullptr, int nz = 1, unsigned gemm_x_bytes = 0,
                         unsigned gemm_w_bytes = 0) {
    const int tm = (M + kBM - 1) / kBM, tn = (N + BN - 1) / BN, per_z = tm * tn, ntiles = per_z * nz;
    const int grid = ntiles < igemm_max_wgs() ? ntiles : igemm_max_wgs();
    const int ck = MODE == kModeDgrad ? g.Co : g.Ci;
    // byte sizes of the gathered tensor and of the weight matrix (the entry points keep both below 4 GiB)
    unsigned x_bytes = (unsigned)(2ull * (MODE == kModeDgrad ? (size_t)g.B * g.Ho * g.Wo * g.Co : (size_t)g.B * g.Hi * g.Wi * g.Ci));
    unsigned w_bytes = (unsigned)(2ull * (size_t)N * (MODE == kModeStem ? (size_t)K : (size_t)g.R * g.S * ck));
    GemmExt gx{ck, MODE == kModeStem ? K : g.R * g.S * ck, 1, per_z, 0u, 0u, 0u, 0u, 0u, 0u};
    if (gemm) {
        gx = *gemm;
        gx.tiles_per_z = per_z;
        x_bytes = gemm_x_bytes;
        w_bytes = gemm_w_bytes;
    }
    g_last_conv_kernel = LP_CONV_KERNEL_IGEMM;
    hipLaunchKernelGGL((conv_igemm_kernel<BN, MODE>), dim3(grid), dim3(256), 0, st, (const unsigned short*)x, (const unsigned short*)w,
                       x_bytes, w_bytes, g, lat, gx, make_fastdiv(lat.nh * lat.nw), make_fastdiv(lat.nw), M, N, K, tn, ntiles, ep);
    return grid;
}

// Pipelined kernel (conv_pipe.h): one 512-thread workgroup per CU walks 256 x BN tiles.  LP_CONV_PIPE=0 sends everything to
// conv_igemm_kernel instead (A/B runs, and the tests that compare the two kernels bit for bit).
static bool conv_pipe_enabled() { return lp_switches().conv_pipe != 0; }

static int pipe_max_wgs() {
    if (lp_switches().conv_max_wgs > 0) return lp_switches().conv_max_wgs;   // (tests: several tiles per workgroup on small problems)
    static int cus = [] {
        int dev = 0, c = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
        return c;
    }();
    return cus;
}

// what the pipelined kernel covers: dense bf16 output of a trunk convolution (no fp32 copy; a bias only in the forward store pass, which
// the Linear layers of lp_gemm_nt use), K a multiple of 64 and > 0, N a
// multiple of its column block, fused BatchNorm sums on the atomic path only, and a BatchNorm segment boundary that falls on a 256-row tile
static bool pipe_eligible(const ConvEpilogue& ep, int M, int N, int K, int ck, long long seg_rows, bool bias_ok = false) {
    if (!conv_pipe_enabled()) return false;
    if (ep.out_bf16 == nullptr || ep.out_f32 != nullptr || (ep.bias != nullptr && !bias_ok) || ep.ldo != N || ep.n_store != N) return false;
    if (K <= 0 || ck % kBK != 0 || N % (N > 64 ? 128 : 64) != 0 || M <= 0) return false;
    if (ep.seg_images > 0 && seg_rows % kPM != 0) return false;
    return true;
}

// the store-pass form of a data gradient (conv_pipe.h: kEk*); -1 = a combination only conv_igemm_kernel implements
static int pipe_dgrad_kind(const ConvEpilogue& ep) {
    if (ep.bn_z != nullptr) {
        if (ep.stats_sums == nullptr || ep.relu_mask != nullptr) return -1;
        if (ep.mask_from_z && !ep.addend && !ep.relu_bits) return kEkZ;
        if (!ep.mask_from_z && ep.addend && ep.relu_bits) return kEkAZB;
        return -1;
    }
    if (ep.stats_sums != nullptr) return -1;
    if (ep.relu_bits != nullptr) return ep.relu_mask == nullptr ? kEkPB : -1;   // (one mask source per launch)
    return kEkPlain;
}

// HALO form (conv_pipe.h): 3x3 / stride 1 / pad 1 with the tile's input neighbourhood staged once per 64-channel slice.  Eligible when the
// neighbourhood of every 256-pixel tile (in padded raster coordinates) fits the kernel's halo image: `cap_rows` = 512 (BN = 64) or 384.
// LP_CONV_HALO=0 keeps those layers on the per-tap ring (A/B runs, bit-identity tests).
static bool pipe_halo_ok(const ConvGeom& g, int M, int ck, int cap_rows) {
    if (lp_switches().conv_halo == 0) return false;
    if (g.R != 3 || g.S != 3 || g.stride != 1 || g.pad != 1 || g.Hi != g.Ho || g.Wi != g.Wo || ck % kBK != 0) return false;
    struct Memo { int B, H, W, cap; bool ok; };
    static thread_local Memo memo[8];
    static thread_local int memo_n = 0;
    for (int i = 0; i < memo_n; ++i)
        if (memo[i].B == g.B && memo[i].H == g.Hi && memo[i].W == g.Wi && memo[i].cap == cap_rows) return memo[i].ok;
    const long long W = g.Wi, H = g.Hi, Wp = W + 2;
    auto padded = [&](long long m) {
        const long long r = m / W, b = r / H;
        return (r + 1 + 2 * b) * Wp + (m - r * W) + 1;
    };
    bool ok = true;
    for (long long m0 = 0; m0 < M && ok; m0 += kPM) {
        const long long last = m0 + kPM - 1 < M ? m0 + kPM - 1 : M - 1;
        ok = padded(last) - padded(m0) + 2 * (W + 3) + 1 <= cap_rows;
    }
    if (memo_n < 8) memo[memo_n++] = Memo{g.B, g.Hi, g.Wi, cap_rows, ok};
    return ok;
}

// conv_res2d_kernel (conv_res2d.h): 3x3 / stride 1 / pad 1 with 64 channels in and out on 16 x 16 pixel tiles, the whole filter resident
// in LDS.  LP_CONV_RES2D=0 leaves those layers to conv_pipe_kernel's HALO form (A/B runs, bit-identity tests).
static bool res2d_ok(const ConvGeom& g, int ck, int N, const ConvEpilogue& ep) {
    if (lp_switches().conv_res2d == 0) return false;
    return g.R == 3 && g.S == 3 && g.stride == 1 && g.pad == 1 && g.Hi == g.Ho && g.Wi == g.Wo && ck == 64 && N == 64 && g.Hi % 16 == 0 &&
           g.Wi % 16 == 0;
}

template <int MODE, bool INFER = false>
static int launch_res2d(const void* x, const void* w, const ConvGeom& g, const ConvEpilogue& ep, hipStream_t st) {
    const int ntiles = g.B * (g.Hi / 16) * (g.Wi / 16);
    const int grid = ntiles < pipe_max_wgs() ? ntiles : pipe_max_wgs();
    const unsigned x_bytes = (unsigned)(2ull * g.B * g.Hi * g.Wi * 64), w_bytes = (unsigned)(2ull * 64 * 9 * 64);
    g_last_conv_kernel = LP_CONV_KERNEL_RES2D;
    hipLaunchKernelGGL((conv_res2d_kernel<MODE, INFER>), dim3(grid), dim3(512), 0, st, (const unsigned short*)x, (const unsigned short*)w, x_bytes, w_bytes,
                       g.B, g.Hi, g.Wi, ntiles, ep);
    return grid;
}

template <int BN, int MODE, int EK, bool HALO = false>
static int launch_pipe(const void* x, const void* w, const ConvGeom& g, const Lattice& lat, int M, int N, int K, const ConvEpilogue& ep,
                        hipStream_t st) {
    const int tm = (M + kPM - 1) / kPM, tn = N / BN, ntiles = tm * tn;
    const int grid = ntiles < pipe_max_wgs() ? ntiles : pipe_max_wgs();
    const int ck = MODE == kModeDgrad ? g.Co : g.Ci;
    const unsigned x_bytes = (unsigned)(2ull * (MODE == kModeDgrad ? (size_t)g.B * g.Ho * g.Wo * g.Co : (size_t)g.B * g.Hi * g.Wi * g.Ci));
    const unsigned w_bytes = (unsigned)(2ull * (size_t)N * g.R * g.S * ck);
    g_last_conv_kernel = HALO ? LP_CONV_KERNEL_PIPE_HALO : LP_CONV_KERNEL_PIPE;
    const HaloDivs hd{make_fastdiv(g.Hi), make_fastdiv(g.Wi + 2), make_fastdiv(g.Hi + 2)};
    hipLaunchKernelGGL((conv_pipe_kernel<BN, MODE, EK, HALO>), dim3(grid), dim3(512), 0, st, (const unsigned short*)x, (const unsigned short*)w,
                       x_bytes, w_bytes, g, lat, make_fastdiv(lat.nh * lat.nw), make_fastdiv(lat.nw), M, N, K, tn, ntiles, ep, hd);
    return grid;
}

template <int BN>
static int launch_pipe_dgrad(int kind, const void* x, const void* w, const ConvGeom& g, const Lattice& lat, int M, int N, int K,
                             const ConvEpilogue& ep, hipStream_t st) {
    if (kind == kEkZ && BN == 64 && ep.bias == nullptr && res2d_ok(g, g.Co, N, ep)) return launch_res2d<kModeDgrad>(x, w, g, ep, st);
    if (kind == kEkZ && pipe_halo_ok(g, M, g.Co, BN == 64 ? 512 : 384)) return launch_pipe<BN, kModeDgrad, kEkZ, true>(x, w, g, lat, M, N, K, ep, st);
    if (kind == kEkZ) return launch_pipe<BN, kModeDgrad, kEkZ>(x, w, g, lat, M, N, K, ep, st);
    if (kind == kEkAZB) return launch_pipe<BN, kModeDgrad, kEkAZB>(x, w, g, lat, M, N, K, ep, st);
    if (kind == kEkPB) return launch_pipe<BN, kModeDgrad, kEkPB>(x, w, g, lat, M, N, K, ep, st);
    return launch_pipe<BN, kModeDgrad, kEkPlain>(x, w, g, lat, M, N, K, ep, st);
}

}  // namespace lp
#include "conv_stem_wgrad.h"
namespace lp {

constexpr int kWgradWgs = 512;  // workgroups per weight-gradient launch (tiles x pixel slices)

struct WgradPlan {
    int tj, tn, split, per;
    bool wide;
    size_t ws_floats;
};

static WgradPlan plan_wgrad(int M, int Kw, int Co, int split_hint, int target_wgs) {
    WgradPlan p;
    p.tj = (Kw + kBM - 1) / kBM;
    p.wide = Co > 64;
    p.tn = p.wide ? (Co + 127) / 128 : 1;
    // the chip holds 512 of these workgroups at once (2 per CU, LDS-bound): aim just below a whole number of rounds
    int split = split_hint > 0 ? split_hint : target_wgs / (p.tj * p.tn);
    const int ksteps = (M + kBK - 1) / kBK;
    if (split > ksteps) split = ksteps;
    if (split < 1) split = 1;
    if (split > 65535) split = 65535;
    p.per = ((ksteps + split - 1) / split) * kBK;
    p.split = (M + p.per - 1) / p.per;
    p.ws_floats = (size_t)p.split * p.tj * p.tn * 4 * 2 * (p.wide ? 2 : 1) * 16 * 64;
    return p;
}

// ---- pipelined weight gradient (conv_pipe.h): 256 x BN tiles, one (tile, pixel slice) per workgroup, ~one workgroup per CU
struct WgradPipePlan {
    bool ok, swap;
    int Ka, Cb, tiles_a, tiles_b, bn, split, per;
    size_t ws_floats;
};

// `force`: ignore the two performance rules (tile waste, HBM-bound shapes) - LP_WGRAD_PIPE=2, the tests' switch
static WgradPipePlan plan_wgrad_pipe(const ConvGeom& g, int split_hint, bool force = false) {
    WgradPipePlan p{};
    const int M = g.B * g.Ho * g.Wo, Kw = g.R * g.S * g.Ci;
    const bool one = g.R == 1 && g.S == 1 && g.stride == 1 && g.pad == 0 && g.Hi == g.Ho && g.Wi == g.Wo;
    if (g.Ci % 8 != 0 || g.Co % 64 != 0 || M < 4 * kBK) return p;
    // a = the 256-wide side.  Ordinary: a = (r, s, ci), b = co.  A 1x1 layer with few input channels is transposed (a = co, b = ci)
    p.swap = one && Kw % 256 != 0 && g.Co % 256 == 0 && Kw % 64 == 0;
    p.Ka = p.swap ? g.Co : Kw;
    p.Cb = p.swap ? Kw : g.Co;
    p.bn = p.Cb % 128 == 0 ? 128 : 64;
    if (p.Cb % p.bn != 0) return p;
    p.tiles_a = (p.Ka + 255) / 256;
    if (!force && (long long)p.tiles_a * 256 * 8 > (long long)p.Ka * 9) return p;   // a ragged last a-tile may waste an eighth at most (3x3 of 64 channels: 576 -> 768 lost, measured)
    // HBM-bound shapes (few FLOPs per operand byte: the 1x1 layers of layer1 / layer2) gain nothing from the bigger tile and lose a few
    // per cent to the 64-B request granularity its LDS swizzle forces on the loads (measured per layer, profiles/archive/r03g_layer_table.txt)
    if (!force && (long long)p.Ka * p.Cb < 120LL * (p.Ka + p.Cb)) return p;
    p.tiles_b = p.Cb / p.bn;
    const int tiles = p.tiles_a * p.tiles_b;
    const int cus = pipe_max_wgs();
    const int ksteps = (M + kBK - 1) / kBK;
    int split = split_hint;
    if (split <= 0) {
        // one workgroup per CU at a time: rounds x K steps per slice, plus the partial tiles' round trip through the workspace
        // (256 x bn fp32 written and read per (tile, slice); a K step moves about as many bytes as 6 % of one partial tile at full rate)
        double best = 1e30;
        for (int s = 1; s <= 2 * cus && s <= ksteps / 2; ++s) {
            if (s != 1 && (tiles * s) % cus > 0 && (tiles * (s + 1) + cus - 1) / cus == (tiles * s + cus - 1) / cus) continue;  // not the fullest split of its round count
            const double rounds = (double)((tiles * s + cus - 1) / cus), per = (double)((ksteps + s - 1) / s);
            const double cost = rounds * per + 0.04 * (p.bn / 128.0) * s * tiles + 2.0 * rounds;   // (in K steps; 2 per round: ring fill)
            if (cost < best) best = cost, split = s;
        }
        if (split <= 0) split = 1;
    }
    if (split > ksteps / 2) split = ksteps / 2;
    if (split < 1) split = 1;
    p.per = ((ksteps + split - 1) / split) * kBK;
    p.split = (M + p.per - 1) / p.per;
    p.ws_floats = (size_t)p.split * tiles * 256 * p.bn;
    p.ok = true;
    return p;
}

static void launch_wgrad_pipe(const WgradPipePlan& p, const void* x, const void* dy, const ConvGeom& g, float* dw, float* ws, hipStream_t st) {
    const int M = g.B * g.Ho * g.Wo, Kw = g.R * g.S * g.Ci;
    const unsigned x_bytes = (unsigned)(2ull * g.B * g.Hi * g.Wi * g.Ci), dy_bytes = (unsigned)(2ull * M * g.Co);
    WgradPipeGeom wg{};
    const void *pa = x, *qb = dy;
    unsigned pa_bytes = x_bytes, qb_bytes = dy_bytes;
    if (p.swap) {   // a = co (rows of dy), b = ci (rows of x): both plain
        wg = WgradPipeGeom{g.B, g.Ho, g.Wo, g.Ho, g.Wo, g.Co, 1, 1, 1, 0, p.Ka, p.Cb, 1};
        pa = dy, qb = x, pa_bytes = dy_bytes, qb_bytes = x_bytes;
    } else {
        const int plain = g.R == 1 && g.S == 1 && g.stride == 1 && g.pad == 0 && g.Hi == g.Ho && g.Wi == g.Wo;
        wg = WgradPipeGeom{g.B, g.Hi, g.Wi, g.Ho, g.Wo, g.Ci, g.R, g.S, g.stride, g.pad, p.Ka, p.Cb, plain};
    }
    const int tiles = p.tiles_a * p.tiles_b;
    const FastDiv dhw = make_fastdiv(g.Ho * g.Wo), dwo = make_fastdiv(g.Wo);
    g_last_conv_kernel = LP_CONV_KERNEL_WGRAD_PIPE;
    const int sa = p.swap ? Kw : 1, sb = p.swap ? 1 : Kw;   // dW[co][kw]: ordinary a = kw index, b = co; swapped a = co, b = kw index
    if (p.bn == 128) {
        hipLaunchKernelGGL((conv_wgrad_pipe_kernel<128>), dim3(tiles * p.split), dim3(512), 0, st, (const unsigned short*)pa,
                           (const unsigned short*)qb, pa_bytes, qb_bytes, wg, M, tiles, p.tiles_b, p.per, dhw, dwo, ws);
        hipLaunchKernelGGL((wgrad_pipe_reduce_kernel<128>), dim3(tiles * 256 * 128 / 64), dim3(256), 0, st, ws, p.split, tiles, p.tiles_b, p.Ka,
                           p.Cb, sa, sb, dw);
    } else {
        hipLaunchKernelGGL((conv_wgrad_pipe_kernel<64>), dim3(tiles * p.split), dim3(512), 0, st, (const unsigned short*)pa,
                           (const unsigned short*)qb, pa_bytes, qb_bytes, wg, M, tiles, p.tiles_b, p.per, dhw, dwo, ws);
        hipLaunchKernelGGL((wgrad_pipe_reduce_kernel<64>), dim3(tiles * 256 * 64 / 64), dim3(256), 0, st, ws, p.split, tiles, p.tiles_b, p.Ka,
                           p.Cb, sa, sb, dw);
    }
}

static bool geom_ok(const lp_conv_geom* c) {
    return c && c->B > 0 && c->Hi > 0 && c->Wi > 0 && c->Ci > 0 && c->Ho > 0 && c->Wo > 0 && c->Co > 0 && c->R > 0 && c->S > 0 &&
           c->stride > 0 && c->pad >= 0;
}

static ConvGeom to_geom(const lp_conv_geom* c) {
    ConvGeom g{c->B, c->Hi, c->Wi, c->Ci, c->Ho, c->Wo, c->Co, c->R, c->S, c->stride, c->pad};
    return g;
}

}  // namespace lp

// out[b][ho][wo][co] = sum x[b][ho*st-pad+r][wo*st-pad+s][ci] * w[co][r][s][ci]  (+bias) ; x, w bf16
static int conv_fwd_impl(const void* x, const void* w, const lp_conv_geom* geom, const float* bias, void* out_bf16, float* out_f32, int ldo,
                         int n_store, const lp_bn_fuse* bn, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(x && w && geom_ok(geom) && (out_bf16 || out_f32) && ldo > 0);
    ConvGeom g = to_geom(geom);
    if (g.Ci % kBK != 0 || g.R * g.S > 32 || (long long)g.B * g.Hi * g.Wi * g.Ci >= (1LL << 31) ||
        (long long)g.B * g.Ho * g.Wo * ldo >= (1LL << 32))
        return LP_ERR_UNSUPPORTED;
    const int M = g.B * g.Ho * g.Wo, N = g.Co, K = g.R * g.S * g.Ci;
    ConvEpilogue ep{(unsigned short*)out_bf16, out_f32, ldo, n_store > 0 ? n_store : N, bias};
    long long split = M;
    if (bn) {
        LP_REQUIRE(bn->sums && bn->seg_images >= 0);
        if (N % 8 != 0) return LP_ERR_UNSUPPORTED;
        split = seg_split_rows(bn->seg_images, g.B, (long long)g.Ho * g.Wo, M);
        if (split < 0) return LP_ERR_UNSUPPORTED;
        bn_fuse_begin(ep, bn);
    }
    hipStream_t st = (hipStream_t)stream;
    const Lattice lat{0, 1, g.Ho, 0, 1, g.Wo, 0, 1, g.R, 0, 1, g.S};
    if (pipe_eligible(ep, M, N, K, g.Ci, split, true)) {
        if (N > 64) {
            const bool halo = pipe_halo_ok(g, M, g.Ci, 384);
            if (halo) launch_pipe<128, kModeFwd, kEkNone, true>(x, w, g, lat, M, N, K, ep, st);
            else launch_pipe<128, kModeFwd, kEkNone>(x, w, g, lat, M, N, K, ep, st);
        } else if (ep.bias == nullptr && res2d_ok(g, g.Ci, N, ep)) {
            launch_res2d<kModeFwd>(x, w, g, ep, st);
        } else if (pipe_halo_ok(g, M, g.Ci, 512)) {
            launch_pipe<64, kModeFwd, kEkNone, true>(x, w, g, lat, M, N, K, ep, st);
        } else {
            launch_pipe<64, kModeFwd, kEkNone>(x, w, g, lat, M, N, K, ep, st);
        }
    } else if (N > 64) launch_igemm<128, kModeFwd>(x, w, g, lat, M, N, K, ep, st);
    else launch_igemm<64, kModeFwd>(x, w, g, lat, M, N, K, ep, st);
    return launch_status();
}

extern "C" int lp_conv_last_kernel(void) { return lp::g_last_conv_kernel; }

extern "C" int lp_conv_fwd(const void* x, const void* w, const lp_conv_geom* geom, const float* bias, void* out_bf16, float* out_f32,
                           int ldo, int n_store, lp_stream_t stream) {
    return conv_fwd_impl(x, w, geom, bias, out_bf16, out_f32, ldo, n_store, nullptr, stream);
}

// out = [relu](conv(x, w) + bias + residual): the inference form of conv -> BatchNorm [-> + identity] [-> ReLU] once the BatchNorm
// (running statistics) is folded into w and bias (lp_bn_fold)
extern "C" int lp_conv_fwd_act(const void* x, const void* w, const lp_conv_geom* geom, const float* bias, const void* residual_bf16, int relu,
                               void* out_bf16, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(x && w && geom_ok(geom) && out_bf16);
    ConvGeom g = to_geom(geom);
    if (g.Ci % kBK != 0 || g.Co % 8 != 0 || g.R * g.S > 32 || (long long)g.B * g.Hi * g.Wi * g.Ci >= (1LL << 31) ||
        (long long)g.B * g.Ho * g.Wo * g.Co >= (1LL << 32))
        return LP_ERR_UNSUPPORTED;
    const int M = g.B * g.Ho * g.Wo, N = g.Co, K = g.R * g.S * g.Ci;
    ConvEpilogue ep{};
    ep.out_bf16 = (unsigned short*)out_bf16, ep.ldo = N, ep.n_store = N, ep.bias = bias;
    ep.addend = (const unsigned short*)residual_bf16, ep.relu_fwd = relu != 0;
    const Lattice lat{0, 1, g.Ho, 0, 1, g.Wo, 0, 1, g.R, 0, 1, g.S};
    hipStream_t st = (hipStream_t)stream;
    // the pipelined forward kernel with the residual and the ReLU in its store pass (LP_INFER_PIPE=0: A/B runs keep conv_igemm_kernel<infer>)
    const bool ip = lp_switches().infer_pipe != 0;
    if (ip && conv_pipe_enabled() && ep.addend == nullptr && res2d_ok(g, g.Ci, N, ep)) {   // (LP_CONV_PIPE=0 keeps these layers on conv_igemm_kernel too)
        // layer1's 64 -> 64 3x3 layers: 16 x 16 tiles, the filter resident in LDS (round 4: the inference store pass of conv_res2d_kernel)
        launch_res2d<kModeFwd, true>(x, w, g, ep, st);
        return launch_status();
    }
    if (ip && pipe_eligible(ep, M, N, K, g.Ci, M, true)) {
        if (N > 64) {
            if (pipe_halo_ok(g, M, g.Ci, 384)) launch_pipe<128, kModeFwd, kEkInfer, true>(x, w, g, lat, M, N, K, ep, st);
            else launch_pipe<128, kModeFwd, kEkInfer>(x, w, g, lat, M, N, K, ep, st);
        } else if (pipe_halo_ok(g, M, g.Ci, 512)) {
            launch_pipe<64, kModeFwd, kEkInfer, true>(x, w, g, lat, M, N, K, ep, st);
        } else {
            launch_pipe<64, kModeFwd, kEkInfer>(x, w, g, lat, M, N, K, ep, st);
        }
        return launch_status();
    }
    g_last_conv_kernel = LP_CONV_KERNEL_IGEMM;
    if (N > 64) launch_igemm<128, kModeInfer>(x, w, g, lat, M, N, K, ep, st);
    else launch_igemm<64, kModeInfer>(x, w, g, lat, M, N, K, ep, st);
    return launch_status();
}

// conv + the [sum, sum of squares] of its (bf16-rounded) output per channel: the statistics pass of the BatchNorm that follows
extern "C" int lp_conv_fwd_bn(const void* x, const void* w, const lp_conv_geom* geom, void* out_bf16, const lp_bn_fuse* bn,
                              lp_stream_t stream) {
    LP_REQUIRE(bn && geom && out_bf16);
    return conv_fwd_impl(x, w, geom, nullptr, out_bf16, nullptr, geom->Co, 0, bn, stream);
}

// C[z][m][n] = sum_k A[z][m][k] * B[z][n][k] (+ bias[n]): the forward kernel as a plain (batched, strided) NT GEMM
extern "C" int lp_gemm_nt(const void* a, int lda, const void* b, int ldb, void* c_bf16, float* c_f32, int ldc, int M, int N, int K,
                          int n_store, const float* bias, const lp_gemm_batch* batch, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(a && b && (c_bf16 || c_f32) && M > 0 && N > 0 && K > 0 && lda >= K && ldb >= K && ldc > 0);
    if (K % kBK != 0 || lda % 8 != 0 || ldb % 8 != 0) return LP_ERR_UNSUPPORTED;
    const int nb = batch ? batch->nb : 1, nh = batch ? batch->nh : 1;
    LP_REQUIRE(nb > 0 && nh > 0);
    const long long a_b = batch ? batch->a_b : 0, a_h = batch ? batch->a_h : 0, b_b = batch ? batch->b_b : 0, b_h = batch ? batch->b_h : 0;
    const long long c_b = batch ? batch->c_b : 0, c_h = batch ? batch->c_h : 0;
    LP_REQUIRE(a_b >= 0 && a_h >= 0 && b_b >= 0 && b_h >= 0 && c_b >= 0 && c_h >= 0);
    const long long a_elems = (nb - 1) * a_b + (nh - 1) * a_h + (long long)(M - 1) * lda + K;
    const long long b_elems = (nb - 1) * b_b + (nh - 1) * b_h + (long long)(N - 1) * ldb + K;
    const long long c_elems = (nb - 1) * c_b + (nh - 1) * c_h + (long long)M * ldc;
    if (a_elems >= (1LL << 31) || b_elems >= (1LL << 31) || c_elems >= (1LL << 32)) return LP_ERR_UNSUPPORTED;
    if ((a_b | a_h | b_b | b_h) % 8 != 0) return LP_ERR_UNSUPPORTED;  // 16-B operand chunks
    ConvGeom g{1, 1, M, K, 1, M, N, 1, 1, 1, 0};
    ConvEpilogue ep{(unsigned short*)c_bf16, c_f32, ldc, n_store > 0 ? n_store : N, bias};
    GemmExt gx{lda, ldb, nh, 0, (unsigned)(2 * a_b), (unsigned)(2 * a_h), (unsigned)(2 * b_b), (unsigned)(2 * b_h), (unsigned)c_b,
               (unsigned)c_h};
    const Lattice lat{0, 1, 1, 0, 1, M, 0, 1, 1, 0, 1, 1};
    hipStream_t st = (hipStream_t)stream;
    const int nstore = ep.n_store;
    // a dense, unbatched product (every Linear layer of the ViT, forward and data gradient) is a 1x1 convolution: the pipelined kernel
    // (LP_GEMM_PIPE=0, A/B: keeps the Linear layers on conv_igemm_kernel)
    if (lp_switches().gemm_pipe != 0 && nb * nh == 1 && lda == K && ldb == K && ldc == N && pipe_eligible(ep, M, N, K, K, M, true)) {
        if (N > 64) launch_pipe<128, kModeFwd, kEkNone>(a, b, g, lat, M, N, K, ep, st);
        else launch_pipe<64, kModeFwd, kEkNone>(a, b, g, lat, M, N, K, ep, st);
        return launch_status();
    }
    if (nstore > 64) launch_igemm<128, kModeFwd>(a, b, g, lat, M, N, K, ep, st, &gx, nb * nh, (unsigned)(2 * a_elems), (unsigned)(2 * b_elems));
    else launch_igemm<64, kModeFwd>(a, b, g, lat, M, N, K, ep, st, &gx, nb * nh, (unsigned)(2 * a_elems), (unsigned)(2 * b_elems));
    return launch_status();
}

// lp_gemm_nt (+ bias) that also writes GELU of its output (conv_pipe.h: kEkGeluFwd)
extern "C" int lp_gemm_nt_gelu_fwd(const void* a, const void* b, const float* bias, void* c_bf16, void* act_bf16, int M, int N, int K,
                                   lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(a && b && c_bf16 && act_bf16 && M > 0 && N > 0 && K > 0);
    if ((long long)M * K >= (1LL << 31) || (long long)N * K >= (1LL << 31) || (long long)M * N >= (1LL << 32)) return LP_ERR_UNSUPPORTED;
    ConvGeom g{1, 1, M, K, 1, M, N, 1, 1, 1, 0};
    ConvEpilogue ep{(unsigned short*)c_bf16, nullptr, N, N, bias};
    ep.out2_bf16 = (unsigned short*)act_bf16;
    const Lattice lat{0, 1, 1, 0, 1, M, 0, 1, 1, 0, 1, 1};
    if (N % 128 != 0 || !pipe_eligible(ep, M, N, K, K, M, true)) return LP_ERR_UNSUPPORTED;
    launch_pipe<128, kModeFwd, kEkGeluFwd>(a, b, g, lat, M, N, K, ep, (hipStream_t)stream);
    return launch_status();
}

// lp_gemm_nt with GELU's backward in the store pass (conv_pipe.h: kEkGeluBwd)
extern "C" int lp_gemm_nt_gelu_bwd(const void* a, const void* b, const void* u_bf16, void* c_bf16, int M, int N, int K, lp_fxsum* colsum,
                                   lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(a && b && u_bf16 && c_bf16 && M > 0 && N > 0 && K > 0);
    if ((long long)M * K >= (1LL << 31) || (long long)N * K >= (1LL << 31) || (long long)M * N >= (1LL << 32)) return LP_ERR_UNSUPPORTED;
    ConvGeom g{1, 1, M, K, 1, M, N, 1, 1, 1, 0};
    ConvEpilogue ep{(unsigned short*)c_bf16, nullptr, N, N, nullptr};
    ep.addend = (const unsigned short*)u_bf16;
    ep.stats_sums = colsum;
    const Lattice lat{0, 1, 1, 0, 1, M, 0, 1, 1, 0, 1, 1};
    if (N % 128 != 0 || !pipe_eligible(ep, M, N, K, K, M)) return LP_ERR_UNSUPPORTED;
    launch_pipe<128, kModeFwd, kEkGeluBwd>(a, b, g, lat, M, N, K, ep, (hipStream_t)stream);
    return launch_status();
}

// Attention backward, score gradient: dS[z] = scale * P[z] o (dO[z] V[z]^T - D[z] 1^T), D = rowsum(dO o O) (lp_attn_rowdot).  The
// product is lp_gemm_nt's; the soft-max backward happens in its store pass, so dP is never written and P is read once here.
extern "C" int lp_attn_dscores(const void* d_out, int ld_do, const void* v, int ldv, const void* p_bf16, const float* d_rows, int d_row_stride,
                               long long d_b, long long d_h, float scale, void* ds_bf16, int ldc, int M, int N, int K,
                               const lp_gemm_batch* batch, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(d_out && v && p_bf16 && d_rows && ds_bf16 && batch && M > 0 && N > 0 && K > 0 && ld_do >= K && ldv >= K && ldc >= N &&
               d_row_stride > 0 && d_b >= 0 && d_h >= 0);
    if (K % kBK != 0 || ld_do % 8 != 0 || ldv % 8 != 0 || ldc % 8 != 0) return LP_ERR_UNSUPPORTED;
    const int nb = batch->nb, nh = batch->nh;
    LP_REQUIRE(nb > 0 && nh > 0 && batch->a_b >= 0 && batch->a_h >= 0 && batch->b_b >= 0 && batch->b_h >= 0 && batch->c_b >= 0 && batch->c_h >= 0);
    const long long a_elems = (nb - 1) * batch->a_b + (nh - 1) * batch->a_h + (long long)(M - 1) * ld_do + K;
    const long long b_elems = (nb - 1) * batch->b_b + (nh - 1) * batch->b_h + (long long)(N - 1) * ldv + K;
    const long long c_elems = (nb - 1) * batch->c_b + (nh - 1) * batch->c_h + (long long)M * ldc;
    const long long d_elems = (nb - 1) * d_b + (nh - 1) * d_h + (long long)(M - 1) * d_row_stride + 1;
    if (a_elems >= (1LL << 31) || b_elems >= (1LL << 31) || c_elems >= (1LL << 32) || d_elems >= (1LL << 32)) return LP_ERR_UNSUPPORTED;
    if ((batch->a_b | batch->a_h | batch->b_b | batch->b_h | batch->c_b | batch->c_h) % 8 != 0) return LP_ERR_UNSUPPORTED;
    ConvGeom g{1, 1, M, K, 1, M, N, 1, 1, 1, 0};
    ConvEpilogue ep{};
    ep.out_bf16 = (unsigned short*)ds_bf16;
    ep.ldo = ldc;
    ep.n_store = ldc;  // the pad columns [N, ldc) are written as zeros, as lp_softmax_rows_bwd left them
    ep.attn_p = (const unsigned short*)p_bf16;
    ep.attn_d = d_rows;
    ep.attn_scale = scale;
    GemmExt gx{ld_do, ldv, nh, 0, (unsigned)(2 * batch->a_b), (unsigned)(2 * batch->a_h), (unsigned)(2 * batch->b_b), (unsigned)(2 * batch->b_h),
               (unsigned)batch->c_b, (unsigned)batch->c_h, (unsigned)d_b, (unsigned)d_h, (unsigned)d_row_stride};
    const Lattice lat{0, 1, 1, 0, 1, M, 0, 1, 1, 0, 1, 1};
    hipStream_t st = (hipStream_t)stream;
    if (ldc > 64) launch_igemm<128, kModeAttn>(d_out, v, g, lat, M, N, K, ep, st, &gx, nb * nh, (unsigned)(2 * a_elems), (unsigned)(2 * b_elems));
    else launch_igemm<64, kModeAttn>(d_out, v, g, lat, M, N, K, ep, st, &gx, nb * nh, (unsigned)(2 * a_elems), (unsigned)(2 * b_elems));
    return launch_status();
}

// dx[b][hi][wi][ci] = sum dy[b][ho][wo][co] * wd[ci][r][s][co] over taps with ho*st - pad + r == hi  (+ addend)
static int conv_dgrad_impl(const void* dy, const void* wd, const lp_conv_geom* geom, const float* bias, const void* addend,
                           const void* relu_mask, void* dx_bf16, float* dx_f32, int ldo, int n_store, int skip_empty_classes,
                           const lp_bn_fuse* bn, lp_stream_t stream, const void* relu_bits = nullptr) {
    using namespace lp;
    LP_REQUIRE(dy && wd && geom_ok(geom) && (dx_bf16 || dx_f32) && ldo > 0);
    ConvGeom g = to_geom(geom);
    if (g.Co % kBK != 0 || (g.stride != 1 && g.stride != 2) || g.R * g.S > 32 || (long long)g.B * g.Ho * g.Wo * g.Co >= (1LL << 31) ||
        (long long)g.B * g.Hi * g.Wi * ldo >= (1LL << 32))
        return LP_ERR_UNSUPPORTED;
    const int N = g.Ci;
    ConvEpilogue ep{(unsigned short*)dx_bf16, dx_f32, ldo, n_store > 0 ? n_store : N, bias, (const unsigned short*)addend,
                    (const unsigned short*)relu_mask};
    if (relu_bits != nullptr) {   // lp_conv_dgrad_bits: the ReLU mask at 1 bit per element of the dense bf16 result ([rows][N / 8] bytes)
        LP_REQUIRE(bn == nullptr && relu_mask == nullptr && dx_bf16 && !dx_f32 && ldo == N && ep.n_store == N);
        if (N % 8 != 0) return LP_ERR_UNSUPPORTED;
        ep.relu_bits = (const unsigned char*)relu_bits;
    }
    if (bn) {
        // the reductions cover every pixel exactly once, so no class may be skipped and the output must be the dense bf16 tensor
        LP_REQUIRE(bn->z && bn->mean && bn->invstd && bn->sums && dx_bf16 && !dx_f32 && !skip_empty_classes &&
                   ldo == N && ep.n_store == N && (!bn->mask_from_z || (bn->gamma && bn->beta)));
        if (N % 8 != 0) return LP_ERR_UNSUPPORTED;
        LP_REQUIRE(!(bn->relu_bits && (relu_mask || bn->mask_from_z)) && bn->seg_images >= 0);
        bn_fuse_begin(ep, bn);
        ep.bn_z = (const unsigned short*)bn->z;
        ep.bn_mean = bn->mean;
        ep.bn_invstd = bn->invstd;
        ep.bn_gamma = bn->gamma;
        ep.bn_beta = bn->beta;
        ep.mask_from_z = bn->mask_from_z;
        ep.relu_bits = (const unsigned char*)bn->relu_bits;
        ep.addend_half = bn->addend_half;
        LP_REQUIRE(!ep.addend_half || (addend && bn->relu_bits && g.stride == 1));
    }
    hipStream_t st = (hipStream_t)stream;
    int n_launches = 0;
    bool seg_ok = true;
    auto launch = [&](const Lattice& lat) {
        const int M = g.B * lat.nh * lat.nw, K = lat.nr * lat.ns * g.Co;
        if (M <= 0) return;
        const long long seg_rows = bn ? seg_split_rows(bn->seg_images, g.B, (long long)lat.nh * lat.nw, M) : M;
        if (seg_rows < 0 || ++n_launches > 4) {
            seg_ok = false;
            return;
        }
        const int kind = pipe_dgrad_kind(ep);
        const bool full_lattice = lat.hstep == 1 && lat.wstep == 1;   // (kEkAZB recomputes its output offsets from the row index)
        if (ep.addend_half && !full_lattice) {   // (stride 1: required above)
            seg_ok = false;
            return;
        }
        if (kind >= 0 && (kind != kEkAZB || full_lattice) && pipe_eligible(ep, M, N, K, g.Co, seg_rows)) {
            if (N > 64) launch_pipe_dgrad<128>(kind, dy, wd, g, lat, M, N, K, ep, st);
            else launch_pipe_dgrad<64>(kind, dy, wd, g, lat, M, N, K, ep, st);
        } else if (N > 64) launch_igemm<128, kModeDgrad>(dy, wd, g, lat, M, N, K, ep, st);
        else launch_igemm<64, kModeDgrad>(dy, wd, g, lat, M, N, K, ep, st);
    };
    if (bn && bn->seg_images > 0) {  // check every launch's segment boundary BEFORE anything is enqueued
        if (bn->seg_images >= g.B) return LP_ERR_UNSUPPORTED;
        const int hh[2] = {g.stride == 1 ? g.Hi : (g.Hi + 1) / 2, g.stride == 1 ? g.Hi : g.Hi / 2};
        const int ww[2] = {g.stride == 1 ? g.Wi : (g.Wi + 1) / 2, g.stride == 1 ? g.Wi : g.Wi / 2};
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b)
                if (hh[a] * ww[b] > 0 && ((long long)bn->seg_images * hh[a] * ww[b]) % kBM != 0) return LP_ERR_UNSUPPORTED;
    }
    if (g.stride == 1) {
        launch(Lattice{0, 1, g.Hi, 0, 1, g.Wi, 0, 1, g.R, 0, 1, g.S});
    } else {
        // stride 2: one launch per parity class of (hi + pad, wi + pad); taps of the same parity only
        for (int ph = 0; ph < 2; ++ph)
            for (int pw = 0; pw < 2; ++pw) {
                const int h0 = ((ph - g.pad) % 2 + 2) % 2, w0 = ((pw - g.pad) % 2 + 2) % 2;
                const int nh = h0 < g.Hi ? (g.Hi - h0 + 1) / 2 : 0, nw = w0 < g.Wi ? (g.Wi - w0 + 1) / 2 : 0;
                const int nr = ph < g.R ? (g.R - ph + 1) / 2 : 0, ns = pw < g.S ? (g.S - pw + 1) / 2 : 0;
                // a class no tap reaches only copies addend (or zeros): the caller may declare dx already correct there
                if (skip_empty_classes && nr * ns == 0) continue;
                launch(Lattice{h0, 2, nh, w0, 2, nw, ph, 2, nr, pw, 2, ns});
            }
    }
    if (!seg_ok) return LP_ERR_UNSUPPORTED;
    return launch_status();
}

extern "C" int lp_conv_dgrad(const void* dy, const void* wd, const lp_conv_geom* geom, const float* bias, const void* addend,
                             const void* relu_mask, void* dx_bf16, float* dx_f32, int ldo, int n_store, int skip_empty_classes,
                             lp_stream_t stream) {
    return conv_dgrad_impl(dy, wd, geom, bias, addend, relu_mask, dx_bf16, dx_f32, ldo, n_store, skip_empty_classes, nullptr, stream);
}

// the same with the ReLU mask read at 1 bit per element (what lp_bn_apply writes beside the activation) instead of from the activation
extern "C" int lp_conv_dgrad_bits(const void* dy, const void* wd, const lp_conv_geom* geom, const void* addend, const void* relu_bits,
                                  void* dx_bf16, int skip_empty_classes, lp_stream_t stream) {
    LP_REQUIRE(relu_bits && geom);
    return conv_dgrad_impl(dy, wd, geom, nullptr, addend, nullptr, dx_bf16, nullptr, geom->Ci, 0, skip_empty_classes, nullptr, stream, relu_bits);
}

// data gradient + ReLU backward + the two reductions of the BatchNorm backward that consumes dx (sum dx, sum dx * xhat)
extern "C" int lp_conv_dgrad_bn(const void* dy, const void* wd, const lp_conv_geom* geom, const void* addend, const void* relu_mask,
                                void* dx_bf16, const lp_bn_fuse* bn, lp_stream_t stream) {
    LP_REQUIRE(bn && geom);
    return conv_dgrad_impl(dy, wd, geom, nullptr, addend, relu_mask, dx_bf16, nullptr, geom->Ci, 0, 0, bn, stream);
}

// stem_wgrad_nb_kernel (conv_stem_wgrad.h): a K step is 64 consecutive pixels of one output row, byte offsets are 32-bit
static bool stem_wgrad_nb_ok(const lp::ConvGeom& g) {
    return lp::lp_switches().stem_wgrad_nb != 0 && g.Wo % 64 == 0 && g.Hi == 2 * g.Ho && g.Wi == 2 * g.Wo &&
           128ull * g.B * g.Ho * g.Wo < (1ull << 31) && 8ull * g.B * g.Hi * g.Wi < (1ull << 31);   // (the kernel forms these byte offsets in signed int)
}
// one workgroup (all 256 gradient rows) per pixel slice, ~4 resident per CU (40 KB of LDS each); the workspace holds two 128-row j-tiles per slice
static lp::WgradPlan plan_stem_wgrad_nb(int M, int split_hint) {
    lp::WgradPlan p{};
    const int ksteps = M / lp::kBK;
    int split = split_hint > 0 ? split_hint : 768;   // (512 - 1024 slices measured equal within 4 %, more are slower: profiles/r05m_stem_wgrad_variants.txt)
    if (split > ksteps) split = ksteps;
    if (split < 1) split = 1;
    p.per = ((ksteps + split - 1) / split) * lp::kBK;
    p.split = (M + p.per - 1) / p.per;
    p.tj = 2, p.tn = 1, p.wide = false;
    p.ws_floats = (size_t)p.split * 2 * 4 * 2 * 16 * 64;
    return p;
}

extern "C" size_t lp_conv_wgrad_workspace_bytes(const lp_conv_geom* geom, int split_hint) {
    using namespace lp;
    if (!geom_ok(geom)) return 0;
    ConvGeom g = to_geom(geom);
    const bool stem = (g.Ci == 4 && g.R == 7);
    const int Kw = stem ? 256 : g.R * g.S * g.Ci;
    size_t fl = plan_wgrad(g.B * g.Ho * g.Wo, Kw, g.Co, split_hint, kWgradWgs).ws_floats;
    if (stem && stem_wgrad_nb_ok(g)) {
        const size_t nb = plan_stem_wgrad_nb(g.B * g.Ho * g.Wo, split_hint).ws_floats;
        if (nb > fl) fl = nb;
    }
    if (!stem) {   // either kernel may take the launch (LP_CONV_PIPE): room for both plans
        const WgradPipePlan pp = plan_wgrad_pipe(g, split_hint, true);   // (the forced plan is the larger one)
        if (pp.ok && pp.ws_floats > fl) fl = pp.ws_floats;
    }
    return fl * sizeof(float);
}

// dw[co][r][s][ci] (fp32, accumulated into) += sum_m x_gather[m][(r,s,ci)] * dy[m][co]
static int conv_wgrad_impl(const void* x, const void* dy, const lp_conv_geom* geom, float* dw, float* dbias, int split_hint, void* workspace,
                           size_t workspace_bytes, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(x && dy && geom_ok(geom) && dw && workspace);
    ConvGeom g = to_geom(geom);
    if (g.Ci % 8 != 0 || g.Co % 8 != 0 || (long long)g.B * g.Hi * g.Wi * g.Ci >= (1LL << 31) ||
        (long long)g.B * g.Ho * g.Wo * g.Co >= (1LL << 31))
        return LP_ERR_UNSUPPORTED;
    const int M = g.B * g.Ho * g.Wo, Kw = g.R * g.S * g.Ci;
    const unsigned x_bytes = (unsigned)(2ull * g.B * g.Hi * g.Wi * g.Ci), dy_bytes = (unsigned)(2ull * M * g.Co);
    const WgradPlan p = plan_wgrad(M, Kw, g.Co, split_hint, kWgradWgs);
    LP_REQUIRE(workspace_bytes >= p.ws_floats * sizeof(float));
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)workspace;
    const int wpe = lp_switches().wgrad_pipe;   // (LP_WGRAD_PIPE, A/B: 0 keeps the weight gradients on conv_wgrad_kernel; 2 = wherever it can run)
    if (!dbias && conv_pipe_enabled() && wpe != 0) {
        const WgradPipePlan pp = plan_wgrad_pipe(g, split_hint, wpe == 2);
        if (pp.ok && workspace_bytes >= pp.ws_floats * sizeof(float)) {
            launch_wgrad_pipe(pp, x, dy, g, dw, ws, st);
            return launch_status();
        }
    }
    g_last_conv_kernel = LP_CONV_KERNEL_WGRAD;
    const int tiles = p.tj * p.tn;
    TnExt ext{};
    ext.col_sums = dbias;
    const dim3 grid(tiles * p.split), block(256);
    const FastDiv dhw = make_fastdiv(g.Ho * g.Wo), dwo = make_fastdiv(g.Wo);
    const unsigned short *xs = (const unsigned short*)x, *dys = (const unsigned short*)dy;
#define LP_WGRAD_LAUNCH(BN_, CS_)                                                                                                \
    hipLaunchKernelGGL((conv_wgrad_kernel<BN_, false, CS_>), grid, block, 0, st, xs, dys, x_bytes, dy_bytes, g, M, Kw, tiles, p.tn, \
                       p.per, dhw, dwo, ws, ext)
    if (p.wide) {
        if (dbias) LP_WGRAD_LAUNCH(128, true);
        else LP_WGRAD_LAUNCH(128, false);
        launch_wgrad_reduce<128>(ws, p.split, tiles, p.tn, Kw, g.Co, dw, st);
    } else {
        if (dbias) LP_WGRAD_LAUNCH(64, true);
        else LP_WGRAD_LAUNCH(64, false);
        launch_wgrad_reduce<64>(ws, p.split, tiles, p.tn, Kw, g.Co, dw, st);
    }
#undef LP_WGRAD_LAUNCH
    return launch_status();
}

extern "C" int lp_conv_wgrad(const void* x, const void* dy, const lp_conv_geom* geom, float* dw, int split_hint, void* workspace,
                             size_t workspace_bytes, lp_stream_t stream) {
    return conv_wgrad_impl(x, dy, geom, dw, nullptr, split_hint, workspace, workspace_bytes, stream);
}

// same, and dbias[co] += sum_m dy[m][co] out of the same pass over dy (Linear / ConvTranspose2d layers)
extern "C" int lp_conv_wgrad_bias(const void* x, const void* dy, const lp_conv_geom* geom, float* dw, float* dbias, int split_hint,
                                  void* workspace, size_t workspace_bytes, lp_stream_t stream) {
    LP_REQUIRE(dbias);
    return conv_wgrad_impl(x, dy, geom, dw, dbias, split_hint, workspace, workspace_bytes, stream);
}

// out[z][j][n] = sum_m x[z][m][j] * y[z][m][n]  (bf16 in / out, fp32 accumulate): the weight-gradient kernel as a batched TN GEMM
extern "C" int lp_gemm_tn(const void* x, int ldx, const void* y, int ldy, void* out_bf16, int ldo, int M, int J, int N,
                          const lp_gemm_batch* batch, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(x && y && out_bf16 && M > 0 && J > 0 && N > 0 && ldx >= J && ldy >= N && ldo >= N);
    if (ldx % 8 != 0 || ldy % 8 != 0 || N % 8 != 0) return LP_ERR_UNSUPPORTED;
    const int nb = batch ? batch->nb : 1, nh = batch ? batch->nh : 1;
    LP_REQUIRE(nb > 0 && nh > 0);
    const long long a_b = batch ? batch->a_b : 0, a_h = batch ? batch->a_h : 0, b_b = batch ? batch->b_b : 0, b_h = batch ? batch->b_h : 0;
    const long long c_b = batch ? batch->c_b : 0, c_h = batch ? batch->c_h : 0;
    LP_REQUIRE(a_b >= 0 && a_h >= 0 && b_b >= 0 && b_h >= 0 && c_b >= 0 && c_h >= 0);
    if ((a_b | a_h | b_b | b_h) % 8 != 0) return LP_ERR_UNSUPPORTED;
    // J (the rows of the result) is read in 16-B chunks: the last chunk of a row of x must stay inside its pitch
    const int Jr = (J + 7) / 8 * 8;
    LP_REQUIRE(Jr <= ldx);
    const long long x_elems = (nb - 1) * a_b + (nh - 1) * a_h + (long long)(M - 1) * ldx + Jr;
    const long long y_elems = (nb - 1) * b_b + (nh - 1) * b_h + (long long)(M - 1) * ldy + N;
    const long long o_elems = (nb - 1) * c_b + (nh - 1) * c_h + (long long)J * ldo;
    if (x_elems >= (1LL << 31) || y_elems >= (1LL << 31) || o_elems >= (1LL << 32)) return LP_ERR_UNSUPPORTED;
    ConvGeom g{1, 1, M, Jr, 1, M, N, 1, 1, 1, 0};   // a 1x1 "convolution" over M pixels: Ci = J (chunk-rounded), Co = N
    const int tj = (Jr + kBM - 1) / kBM;
    TnExt tn{(unsigned short*)out_bf16, ldx, ldy, ldo, nh, (unsigned)(2 * a_b), (unsigned)(2 * a_h), (unsigned)(2 * b_b), (unsigned)(2 * b_h),
             (unsigned)c_b, (unsigned)c_h};
    hipStream_t st = (hipStream_t)stream;
    const int nz = nb * nh;
    if (N > 64) {
        const int tn_ = (N + 127) / 128;
        hipLaunchKernelGGL((conv_wgrad_kernel<128, false>), dim3(tj * tn_ * nz), dim3(256), 0, st, (const unsigned short*)x,
                           (const unsigned short*)y, (unsigned)(2 * x_elems), (unsigned)(2 * y_elems), g, M, J, tj * tn_, tn_, M,
                           make_fastdiv(M), make_fastdiv(M), (float*)nullptr, tn);
    } else {
        hipLaunchKernelGGL((conv_wgrad_kernel<64, false>), dim3(tj * nz), dim3(256), 0, st, (const unsigned short*)x,
                           (const unsigned short*)y, (unsigned)(2 * x_elems), (unsigned)(2 * y_elems), g, M, J, tj, 1, M, make_fastdiv(M),
                           make_fastdiv(M), (float*)nullptr, tn);
    }
    return launch_status();
}

// 7x7/2 stem on NHWC4 bf16 input (channel 3 = 0): weights [64][7+1][8][4] zero padded (K = 256)
static int stem_fwd_impl(const void* x4, const void* w, const lp_conv_geom* geom, void* out_bf16, const lp_bn_fuse* bn,
                         lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(x4 && w && geom_ok(geom) && out_bf16);
    ConvGeom g = to_geom(geom);
    if (g.R != 7 || g.S != 7 || g.stride != 2 || g.pad != 3 || g.Ci != 4 || g.Co != 64) return LP_ERR_UNSUPPORTED;
    const int M = g.B * g.Ho * g.Wo;
    ConvEpilogue ep{(unsigned short*)out_bf16, nullptr, 64, 64};
    if (bn) {
        LP_REQUIRE(bn->sums && bn->seg_images >= 0);
        if (seg_split_rows(bn->seg_images, g.B, (long long)g.Ho * g.Wo, M) < 0) return LP_ERR_UNSUPPORTED;
        bn_fuse_begin(ep, bn);
    }
    // conv_stem2d_kernel (conv_res2d.h): 16 x 16 output tiles, filter resident in LDS.  LP_STEM_2D=0 (A/B runs, bit-identity tests) keeps
    // conv_igemm_kernel<64, stem>
    if (conv_pipe_enabled() && lp_switches().stem_2d != 0 && g.Ho % 16 == 0 && g.Wo % 16 == 0 && g.Hi == 2 * g.Ho && g.Wi == 2 * g.Wo) {
        const int ntiles = g.B * (g.Ho / 16) * (g.Wo / 16);
        const int grid = ntiles < 2 * pipe_max_wgs() ? ntiles : 2 * pipe_max_wgs();
        g_last_conv_kernel = LP_CONV_KERNEL_RES2D;
        hipLaunchKernelGGL(conv_stem2d_kernel, dim3(grid), dim3(512), 0, (hipStream_t)stream, (const unsigned short*)x4, (const unsigned short*)w,
                           (unsigned)(2ull * g.B * g.Hi * g.Wi * 4), g.B, g.Ho, g.Wo, ntiles, ep);
    } else {
        const Lattice lat{0, 1, g.Ho, 0, 1, g.Wo, 0, 1, g.R, 0, 1, g.S};
        launch_igemm<64, kModeStem>(x4, w, g, lat, M, 64, 256, ep, (hipStream_t)stream);
    }
    return launch_status();
}

extern "C" int lp_stem_fwd(const void* x4, const void* w, const lp_conv_geom* geom, void* out_bf16, lp_stream_t stream) {
    return stem_fwd_impl(x4, w, geom, out_bf16, nullptr, stream);
}

extern "C" int lp_stem_fwd_bn(const void* x4, const void* w, const lp_conv_geom* geom, void* out_bf16, const lp_bn_fuse* bn,
                              lp_stream_t stream) {
    LP_REQUIRE(bn);
    return stem_fwd_impl(x4, w, geom, out_bf16, bn, stream);
}

// dw[64][8][8][4] fp32 (K = 256 layout of lp_stem_fwd) += ...
extern "C" int lp_stem_wgrad(const void* x4, const void* dy, const lp_conv_geom* geom, float* dw, int split_hint, void* workspace,
                             size_t workspace_bytes, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(x4 && dy && geom_ok(geom) && dw && workspace);
    ConvGeom g = to_geom(geom);
    if (g.R != 7 || g.S != 7 || g.stride != 2 || g.pad != 3 || g.Ci != 4 || g.Co != 64) return LP_ERR_UNSUPPORTED;
    const int M = g.B * g.Ho * g.Wo, Kw = 256;
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)workspace;
    if (stem_wgrad_nb_ok(g)) {   // one workgroup per pixel slice forms all 256 gradient rows from the staged input neighbourhood
        const WgradPlan p = plan_stem_wgrad_nb(M, split_hint);
        LP_REQUIRE(workspace_bytes >= p.ws_floats * sizeof(float));
        g_last_conv_kernel = LP_CONV_KERNEL_STEM_WGRAD_NB;
        hipLaunchKernelGGL(stem_wgrad_nb_kernel, dim3(p.split), dim3(256), 0, st, (const unsigned short*)x4, (const unsigned short*)dy,
                           (unsigned)(8ull * g.B * g.Hi * g.Wi), (unsigned)(128ull * M), g, M, p.per, make_fastdiv(g.Ho * g.Wo), make_fastdiv(g.Wo), ws);
        launch_wgrad_reduce<64>(ws, p.split, 2, 1, Kw, 64, dw, st);
        return launch_status();
    }
    const WgradPlan p = plan_wgrad(M, Kw, 64, split_hint, kWgradWgs);
    LP_REQUIRE(workspace_bytes >= p.ws_floats * sizeof(float));
    g_last_conv_kernel = LP_CONV_KERNEL_WGRAD;
    hipLaunchKernelGGL((conv_wgrad_kernel<64, true>), dim3(p.tj * p.split), dim3(256), 0, st, (const unsigned short*)x4,
                       (const unsigned short*)dy, 0u, 0u, g, M, Kw, p.tj, 1, p.per, make_fastdiv(g.Ho * g.Wo), make_fastdiv(g.Wo), ws, TnExt{});
    launch_wgrad_reduce<64>(ws, p.split, p.tj, 1, Kw, 64, dw, st);
    return launch_status();
}
