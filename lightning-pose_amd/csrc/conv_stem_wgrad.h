// Weight gradient of the ResNet stem (7x7, stride 2, pad 3; NHWC4 input, 64 output channels) with the input NEIGHBOURHOOD staged in LDS
// instead of an im2col tile (round 5).
//
// conv_wgrad_kernel<64, STEM> treats the stem like any other layer: per 64-pixel K step it gathers a [64 pixels][128 taps x channels] tile
// of x from global memory per j-tile (two j-tiles: 32 KB of 8-B gathers + the 8 KB dy tile TWICE per 64 pixels, every address through the
// generic per-pixel divisions) and ran at 0.67 ms against a 0.21 ms byte floor - the last kernel of the backward pass, fully exposed.
// Here a K step is 64 consecutive pixels of ONE output row (the host checks Wo % 64 == 0), whose 7x7/2 taps all lie in 7 input rows x
// 136 input pixels: that neighbourhood (7.4 KB, contiguous 16-B pieces of the NHWC4 image; rows / columns outside the image read zeros
// through the buffer range check) is staged once per step next to the dy tile (8 KB, loaded once), and ONE workgroup forms all
// 7 x 8 x 4 = 224 (+ 32 padding) rows of the gradient from it: the MFMA fragment "8 consecutive pixels of tap (r, s), channel c" is two
// ds_read_b64_tr_b16 whose lanes address pixel (2 p + s) of row r directly - an NHWC4 pixel is exactly the 8-B run the transpose read
// takes from a lane - so no im2col image exists anywhere.  Global -> LDS traffic per 64 pixels: 15.6 KB instead of 48 KB.
//
//   workgroup  256 threads = 4 waves; wave w owns filter rows r = 2 w, 2 w + 1 (32 gradient rows (s8, c4) each) x all 64 output channels
//   LDS        2 x (7 x 1088 B neighbourhood + 64 x 192 B dy tile) = 39.8 KB: the next step's loads fly during this step's MFMAs
//   output     fp32 partial tiles per pixel slice in conv_wgrad_kernel<64>'s workspace order (two 128-row j-tiles), so the same
//              fixed-order reduction (launch_wgrad_reduce<64>) adds them into dw[64][8][8][4]; the padding entries (r = 7, s = 7) are
//              written as zeros (s = 7 reads real pixels - its products are dropped here; filter row 7 is never computed)
// The accumulation order per element (pixels ascending inside a slice, slices ascending in the reduction) is conv_wgrad_kernel's; the
// slice boundaries differ (one tile per slice instead of two), so results agree with it to fp32 reassociation across slices only.
#pragma once

namespace lp {

constexpr int kSnRows = 7, kSnChunks = 68, kSnRowB = kSnChunks * 16;   // neighbourhood: input rows x 16-B pieces (2 NHWC4 pixels each)
constexpr int kSnBytes = kSnRows * kSnRowB, kSnPieces = kSnRows * kSnChunks;
constexpr int kSnLdb = 64 + 32;                                        // dy tile row pitch (elements), as conv_wgrad_kernel<64>

#ifndef LP_STEM_NB_WGS
#define LP_STEM_NB_WGS 2   // workgroups per CU the register budget allows (A/B builds: 3 caps the kernel at 168 registers)
#endif
__global__ __launch_bounds__(256, LP_STEM_NB_WGS) void stem_wgrad_nb_kernel(const unsigned short* __restrict__ X, const unsigned short* __restrict__ DY,
                                                            unsigned x_bytes, unsigned dy_bytes, ConvGeom g, int M, int m_per_split,
                                                            FastDiv div_hw, FastDiv div_wo, float* __restrict__ ws) {
    __shared__ __attribute__((aligned(16))) unsigned char sN[2][kSnBytes];
    __shared__ __attribute__((aligned(16))) unsigned short sB[2][kBK * kSnLdb];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slice = xcd_remap(blockIdx.x, gridDim.x);
    const int m_begin = slice * m_per_split, m_end = min(M, m_begin + m_per_split);
    const buf_rsrc rsrc_x = make_buf_rsrc(X, x_bytes), rsrc_dy = make_buf_rsrc(DY, dy_bytes);
    const int hw = g.Ho * g.Wo;

    // loader: pieces tid and tid + 256 of the neighbourhood (row rr, pieces cc: input pixels 2 wo0 - 4 + 2 cc, + 1), dy rows 2 pgB, + 1
    int rr[2], cc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = tid + 256 * i;
        rr[i] = q / kSnChunks;
        cc[i] = q - rr[i] * kSnChunks;
    }
    const bool second = tid + 256 < kSnPieces;
    const int ncB = tid & 7, pgB = tid >> 3;
    u16x8 ra[2], rb[2];
    auto load_step = [&](int mk) {   // mk = first pixel of the step: (b, ho, wo0), wo0 a multiple of 64
        const int b = fdiv(mk, div_hw), rem = mk - b * hw;
        const int ho = fdiv(rem, div_wo), wo0 = rem - ho * g.Wo;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int hi = 2 * ho - 3 + rr[i], wi = 2 * wo0 - 4 + 2 * cc[i];
            const bool ok = (i == 0 || second) && (unsigned)hi < (unsigned)g.Hi && (unsigned)wi < (unsigned)g.Wi;   // (Wi even: a piece is inside or outside as a whole)
            ra[i] = buf_load16(rsrc_x, ok ? (unsigned)(((b * g.Hi + hi) * g.Wi + wi) * 8) : ~0u, 0u);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) rb[i] = buf_load16(rsrc_dy, (unsigned)(((mk + pgB * 2 + i) * 64 + ncB * 8) * 2), 0u);
    };
    auto store_step = [&](int buf) {
        *reinterpret_cast<u16x8*>(&sN[buf][tid * 16]) = ra[0];
        if (second) *reinterpret_cast<u16x8*>(&sN[buf][(tid + 256) * 16]) = ra[1];
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<u16x8*>(&sB[buf][(pgB * 2 + i) * kSnLdb + ncB * 8]) = rb[i];
    };

    // fragments (lp_common.h: lds_read_tr16): lane l of 16-lane group q supplies pixel 8 (q / 2) + (l % 16) / 4 of the k-slice and the run
    // 4 (q % 2) + l % 4 of its 16-column block - for dy 4 consecutive channels, for x the 4 channels of tap s = that run, i.e. input pixel
    // 2 p + s - 3 of filter row r, which sits at piece-pixel 2 p + s + 1 of neighbourhood row r (the row starts at input pixel 2 wo0 - 4)
    const int fq = lane >> 4, fi = lane & 15;
    const int frow = (fq >> 1) * 8 + (fi >> 2), frun = (fq & 1) * 4 + (fi & 3);
    const unsigned a_off = (unsigned)((2 * frow + frun + 1) * 8);
    auto frag_x = [&](int buf, int r, int kk) -> bf16x8 {
        const unsigned short* p = reinterpret_cast<const unsigned short*>(&sN[buf][r * kSnRowB + a_off + kk * (16 * 2 * 8)]);
        const s16x4_t lo = lds_read_tr16(p), hi = lds_read_tr16(p + 4 * 2 * 4);   // pixels + 4: 4 x 2 input pixels x 4 channels further
        typedef __attribute__((ext_vector_type(8))) short s16x8_t;
        const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };
    auto frag_dy = [&](int buf, int nt, int kk) -> bf16x8 {
        const unsigned short* p = &sB[buf][(kk * 16 + frow) * kSnLdb + nt * 32 + frun * 4];
        const s16x4_t lo = lds_read_tr16(p), hi = lds_read_tr16(p + 4 * kSnLdb);
        typedef __attribute__((ext_vector_type(8))) short s16x8_t;
        const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };

    const int nmt = wave == 3 ? 1 : 2;   // filter row 7 does not exist
    f32x16 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;
    auto mma_step = [&](int buf) {
        bf16x8 a[2][2], b[2][2];
        auto fetch = [&](int kk, int set) {
            a[set][0] = frag_x(buf, 2 * wave, kk);
            if (nmt == 2) a[set][1] = frag_x(buf, 2 * wave + 1, kk);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) b[set][nt] = frag_dy(buf, nt, kk);
        };
        fetch(0, 0);
#pragma unroll
        for (int kk = 0; kk < kBK / 16; ++kk) {
            if (kk + 1 < kBK / 16) fetch(kk + 1, (kk + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);   // keep the next slice's reads ahead of this slice's MFMAs
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kk & 1][0], b[kk & 1][nt], acc[0][nt], 0, 0, 0);
            if (nmt == 2) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kk & 1][1], b[kk & 1][nt], acc[1][nt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    const int KT = (m_end - m_begin) / kBK;   // (M and the slice length are multiples of 64: host-checked)
    if (KT > 0) {
        load_step(m_begin);
        store_step(0);
        __syncthreads();
        for (int kt = 0; kt < KT; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < KT) load_step(m_begin + (kt + 1) * kBK);
            mma_step(cur);
            if (kt + 1 < KT) store_step(cur ^ 1);
            __syncthreads();
        }
    }

    // partial tile -> workspace[slice][j-tile][wave'][mt][e][lane] of conv_wgrad_kernel<64> (waves' = (row half, column half) of a 128-row
    // j-tile): this wave's rows 64 w .. + 63 are row half w & 1 of j-tile w >> 1, its column block nt is column half nt.
    // Register e of lane l is gradient row (e & 3) + 8 (e >> 2) + 4 (l >> 5) of its 32-block = (s, c) with s = row >> 2: s = 7 <=> e >= 12, l >= 32
    const bool s7 = lane >= 32;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            float* dst = ws + ((((size_t)slice * 2 + (wave >> 1)) * 4 + ((wave & 1) * 2 + nt)) * (2 * 16) + mt * 16) * 64 + lane;
#pragma unroll
            for (int e = 0; e < 16; ++e) dst[e * 64] = ((e >= 12 && s7) || mt >= nmt) ? 0.f : acc[mt][nt][e];
        }
}

}  // namespace lp
