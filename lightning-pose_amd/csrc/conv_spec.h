// Producer / consumer wave-specialised implicit-GEMM convolution (round 5): conv_pipe_kernel's operand ring (conv_pipe.h) with the two jobs
// of a wave - moving operands and multiplying them - given to DIFFERENT waves of the workgroup.
//
// Why.  Round 4's loop probes (profiles/r04z_specialised_waves.txt): in conv_pipe_kernel every one of the 8 waves issues its share of a K
// step's direct-to-LDS loads AND its MFMAs.  A `buffer_load ... lds` costs the wave that issues it ~60 - 110 cycles in which it issues no
// MFMA, the 8 symmetric waves are always in the same phase, and so the matrix pipe idles while everybody loads: the loop tops out at
// 1060 - 1160 TFLOP/s over the chip, against 2200 - 2500 for the MFMAs + barrier alone.  With 4 CONSUMER waves (one per SIMD: fragment reads
// and MFMAs only, a 128 x 64 wave tile = 128 accumulator registers) and 4 PRODUCER waves (one per SIMD: all loads of a K step) on the
// same 256 x 128 tile and 3-stage ring, the loop measured 1430 - 1580.  What the probe left open is the store pass, which in the open costs
// a K = 256 tile 45 % on top (conv_pipe.h) and would cost four consumer waves twice that.  Here it is HANDED OVER:
//
//   tile end    barrier E (every consumer is done reading the ring stage / halo image the tile consumed last)
//               consumers convert their accumulators to bf16 and write the whole 256 x 128 tile - 64 KB - into LDS that is dead at that
//               moment: the stage consumed last (48 KB) + a 16-KB spare (ring form), or the halo image of the tile's last channel slice
//               (48 KB) + the weight stage consumed last (16 KB) (HALO form).  ~16 ds_write_b64 x 2 per lane; then straight on to the next
//               tile's K loop - its first two K steps are already in the ring.
//   next tile   at the barrier of its K step 0 the PRODUCERS read the tile back (each producer exactly the 16 KB that ITS OWN next loads
//               will overwrite - stage pieces and halo pieces are dealt round-robin by piece index mod 4 - so no further synchronisation is
//               needed: a producer issues a load into a piece only after its own read of that piece has returned), and store it over the
//               tile's first four K steps, 4 x 1 KB per producer and step, 16-B row pieces = full 256-B row segments per 16 lanes; the fused
//               BatchNorm sums and - data gradient - the ReLU mask recomputed from the saved pre-normalisation tensor z travel with the
//               stores, on the producers, whose registers are free (no accumulators).
//
// A staged tile is 64 pieces of 1 KB: piece P = tile rows 4 P .. 4 P + 3, each 256 B = 16 chunks of 8 channels; chunk c of row r sits at
// position c ^ (r & 15) (consumers write 8-B halves: conflict-free ds_write_b64 for 16 consecutive rows; producers read lane-linear).
// Producer p owns the pieces P = p (mod 4): a lane then always holds the same 8 channels, and its BatchNorm sums are per-THREAD running
// sums over all tiles of the walk, as in conv_pipe_kernel.
//
// Scope: forward (kEkNone) and the data gradient whose store pass takes BatchNorm's backward sums and recomputes the ReLU mask from z
// (kEkZ: conv3 / conv2 of every block), BN = 128, at least 4 (forward) / 7 (data gradient) K steps per tile; per-tap ring form and HALO form.
// Everything else stays on conv_pipe_kernel.  Numerics: the K order and every rounding point equal conv_pipe_kernel's - outputs are
// bit-identical to it (tests/test_emu_conv_spec.py, on the device tests/test_gpu_fullsize.py); only the fp32 order of the fused sums differs.
//
// vmcnt discipline of a producer.  Loads retire in order among themselves, stores may overtake them, so `s_waitcnt vmcnt(n)` proves that a
// load has landed iff n <= the number of YOUNGER LOADS (stores in flight only make it wait longer).  Every counted wait below uses exactly
// the number of loads this wave issued after the ones it needs; the read-back of a staged tile uses inline-asm ds_read_b128 so that the
// compiler does not order it behind the pending direct-to-LDS loads with a vmcnt(0).
#pragma once

namespace lp {

#if defined(__HIP_DEVICE_COMPILE__)
#define LP_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define LP_SETPRIO(n) __builtin_amdgcn_s_setprio(n)
// 16 B from LDS without the compiler's knowledge of the access (see lds_read_tr16_async): the caller waits with LP_SPEC_TOUCH4
__device__ __forceinline__ u16x8 lds_read16_async(const unsigned char* p) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
    u32x4_t v;
    const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)p;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a) : "memory");
    return __builtin_bit_cast(u16x8, v);
}
#define LP_SPEC_TOUCH4(a, b, c, d) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory")
#else
#define LP_WAIT_LGKM0() ((void)0)
#define LP_SETPRIO(n) ((void)0)
__device__ __forceinline__ u16x8 lds_read16_async(const unsigned char* p) { return *reinterpret_cast<const u16x8*>(p); }
#define LP_SPEC_TOUCH4(a, b, c, d) ((void)0)
#endif

constexpr int kSpecPieceB = 1024;   // bytes of a staged piece = one wave instruction of 16 B per lane

template <int MODE, int EK, bool HALO>
__global__ __launch_bounds__(512) void conv_spec_kernel(const unsigned short* __restrict__ X, const unsigned short* __restrict__ Wt,
                                                        unsigned x_bytes, unsigned w_bytes, ConvGeom g, Lattice lat, FastDiv div_img,
                                                        FastDiv div_row, int M, int N, int K, int tiles_n, int ntiles, ConvEpilogue ep,
                                                        HaloDivs hd) {
    static_assert((MODE == kModeFwd && EK == kEkNone) || (MODE == kModeDgrad && EK == kEkZ), "forward, or the data gradient with the z read-back");
    constexpr int BN = 128;
    constexpr int kStageA = HALO ? 0 : kPM * kPRowB, kStageB = BN * kPRowB, kStage = kStageA + kStageB;
    constexpr int kHaloRows = 384, kHaloB = kHaloRows * kPRowB;   // (host-checked against the geometry: pipe_halo_ok)
    constexpr int NA = kHaloRows / 64;                            // K steps over which a halo image is issued (8 KB each)
    constexpr int kTail = HALO ? 2 * kHaloB : 16 * kSpecPieceB;   // behind the ring: the two halo images, or the 16-KB spare of a staged tile
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * kStage + kTail];
    unsigned char* const tail0 = smem + 3 * kStage;
    constexpr bool kFwd = MODE == kModeFwd;
    constexpr bool bwd = EK == kEkZ;
    constexpr int kRingLoads = HALO ? 4 : 12;    // ring loads per producer and K step (HALO: its 4 weight pieces)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= 4;
    const int rows_y = lat.nh, rows_x = lat.nw;
    const int full_h = (MODE == kModeDgrad) ? g.Hi : g.Ho;
    const int full_w = (MODE == kModeDgrad) ? g.Wi : g.Wo;
    const int ck = (MODE == kModeDgrad) ? g.Co : g.Ci;   // channels of the gathered tensor
    const int KT = K / kBK;
    const int hW = g.Wi, hH = g.Hi, hWp = g.Wi + 2, hHp = g.Hi + 2;   // (HALO: stride 1, "same")
    const bool want_stats = ep.stats_sums != nullptr;
    auto padded = [&](int m) __attribute__((always_inline)) {   // padded raster coordinate of pixel row m (conv_pipe.h)
        const int rm = fdiv(m, div_row), bm = fdiv(rm, hd.h);
        return (rm + 1 + 2 * bm) * hWp + (m - rm * hW) + 1;
    };
    auto padded_row = [&](int m) __attribute__((always_inline)) {
        const int rm = fdiv(m, div_row), bm = fdiv(rm, hd.h);
        return rm + 1 + 2 * bm;
    };
    auto tile_of = [&](int vt, int& m0, int& n0, int& seg_off) __attribute__((always_inline)) {
        const int tile = xcd_remap(vt, ntiles);
        const int tm_ = tile / tiles_n;
        m0 = tm_ * kPM;
        n0 = (tile - tm_ * tiles_n) * BN;
        seg_off = (ep.seg_images > 0 && m0 >= ep.seg_images * rows_y * rows_x) ? N : 0;
    };
    // where a staged tile lives once the tile that consumed ring stage `st` (HALO: halo image `hb`) last has ended
    auto region_a = [&](int st, int hb) __attribute__((always_inline)) -> unsigned char* { return HALO ? tail0 + hb * kHaloB : smem + st * kStage; };
    auto region_b = [&](int st) __attribute__((always_inline)) -> unsigned char* { return HALO ? smem + st * kStage : tail0; };

    if (producer) {
        // =====================================================================================================================
        // PRODUCER: the loader walk of conv_pipe_kernel for a quarter of every stage (pieces p, p + 4, ...), + the handed-over store pass
        // =====================================================================================================================
        const int pw = wave - 4;
        const buf_rsrc rsrc_x = make_buf_rsrc(X, x_bytes), rsrc_w = make_buf_rsrc(Wt, w_bytes);
        const int src_h = (MODE == kModeDgrad) ? g.Ho : g.Hi;
        const int src_w = (MODE == kModeDgrad) ? g.Wo : g.Wi;
        const bool halved = (MODE == kModeDgrad) && g.stride == 2;
        const int ldw = g.R * g.S * ck;
        const int rloc = lane >> 3, slot = lane & 7;
        constexpr int PA = HALO ? 1 : 8;   // pixel pieces (8 rows each) per producer and K step
        unsigned rowoff[PA], vmask[PA], voff[PA], wrow[4];
        int py[PA], px[PA];
        int tir = 0, tis = 0, tc = 0;
        unsigned wtap = 0;
        int ld_vt = blockIdx.x, ld_kt = 0;
        auto setup = [&](int vt) __attribute__((always_inline)) {
            if (vt >= ntiles) {   // past the end of the walk: the ring keeps turning on loads that fetch nothing, so the counted waits stay valid
#pragma unroll
                for (int i = 0; i < PA; ++i) vmask[i] = 0u, rowoff[i] = 0u;
#pragma unroll
                for (int i = 0; i < 4; ++i) wrow[i] = ~0u;
                tir = tis = tc = 0;
                return;
            }
            const int tile = xcd_remap(vt, ntiles);
            const int tm_ = tile / tiles_n;
            const int m0 = tm_ * kPM, n0 = (tile - tm_ * tiles_n) * BN;
#pragma unroll
            for (int i = 0; i < (HALO ? 0 : PA); ++i) {
                const int lrow = (i * 4 + pw) * 8 + rloc;              // row of the stage's pixel image
                const int lchunk = slot ^ ((lrow >> 1) & 7);
                const int m = m0 + lrow;
                const bool pv = m < M;
                const int mm = pv ? m : 0;
                const int b = fdiv(mm, div_img);
                const int rem = mm - b * rows_y * rows_x;
                const int iy = fdiv(rem, div_row);
                const int y = lat.h0 + lat.hstep * iy, xq = lat.w0 + lat.wstep * (rem - iy * rows_x);
                if (MODE == kModeDgrad) {
                    py[i] = y + g.pad;
                    px[i] = xq + g.pad;
                } else {
                    py[i] = y * g.stride - g.pad;
                    px[i] = xq * g.stride - g.pad;
                }
                const int oy = halved ? (py[i] >> 1) : py[i], ox = halved ? (px[i] >> 1) : px[i];
                rowoff[i] = (unsigned)(((b * src_h + oy) * src_w + ox) * ck + lchunk * 8) * 2u;
                unsigned mask = 0;
                if (pv) {
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
            for (int i = 0; i < 4; ++i) {
                const int lrow = (i * 4 + pw) * 8 + rloc;                  // row of the stage's weight image
                const int lchunk = slot ^ ((lrow >> 1) & 7);
                wrow[i] = (unsigned)((n0 + lrow) * ldw + lchunk * 8) * 2u;   // (N % 128 == 0: host-checked)
            }
            tir = tis = tc = 0;
        };
        unsigned is_soff_a = 0, is_soff_b = 0;
        unsigned char* is_dst = smem;
        bool need_setup = true;   // (setup() has ONE call site: it is a few hundred instructions per pixel piece)
        auto load_step = [&](int st) __attribute__((always_inline)) {   // all of this producer's loads of one K step -> stage st, then the loader moves on
            if (need_setup) {
                setup(ld_vt);
                need_setup = false;
            }
            if (HALO) {   // slice-major K order: tir = tap (0 .. 8), tc = first channel of the slice; only the weights ride the ring
                const int tr = tir / 3, ts = tir - tr * 3;
                is_soff_b = (unsigned)(((tr * g.S + ts) * ck + tc) * 2);
                if (++tir == 9) tir = 0, tc += kBK;
            } else {
                if (tc == 0) {   // entering a filter tap: its per-row offsets (an invalid tap gets ~0 -> the range check returns zeros)
                    const int tr = lat.r0 + lat.rstep * tir, ts = lat.s0 + lat.sstep * tis;
                    const int qr = halved ? (tr >> 1) : tr, qs = halved ? (ts >> 1) : ts;
                    const unsigned tapoff_b = (unsigned)(((MODE == kModeDgrad) ? -(qr * src_w + qs) : (qr * src_w + qs)) * ck) * 2u;
                    const int tap = tir * lat.ns + tis;
                    wtap = (unsigned)((tr * g.S + ts) * ck) * 2u;
#pragma unroll
                    for (int i = 0; i < PA; ++i) {
                        const unsigned ok = (vmask[i] >> tap) & 1u;
                        voff[i] = (rowoff[i] + tapoff_b) | (ok - 1u);
                    }
                }
                is_soff_a = (unsigned)tc * 2u;
                is_soff_b = wtap + is_soff_a;
                tc += kBK;
                if (tc >= ck) {
                    tc = 0;
                    if (++tis == lat.ns) {
                        tis = 0;
                        ++tir;
                    }
                }
            }
            is_dst = smem + st * kStage;
#pragma unroll
            for (int i = 0; i < (HALO ? 0 : PA); ++i) buf_load16_lds(rsrc_x, is_dst + (i * 4 + pw) * kSpecPieceB, voff[i], is_soff_a);
#pragma unroll
            for (int i = 0; i < 4; ++i) buf_load16_lds(rsrc_w, is_dst + kStageA + (i * 4 + pw) * kSpecPieceB, wrow[i], is_soff_b);
            if (++ld_kt == KT) {
                ld_kt = 0;
                ld_vt += gridDim.x;
                need_setup = true;
            }
        };
        // ---- HALO loader: halo image = 48 pieces of 8 rows; this producer's are pieces s = 8 t + 4 h + pw, two per K step t < NA
        unsigned hvoff[HALO ? 2 * NA : 1];
        int hl_vt = blockIdx.x, hl_slice = 0, hl_buf = 0;
        auto halo_setup = [&](int vt) __attribute__((always_inline)) {
            if (vt >= ntiles) {
#pragma unroll
                for (int i = 0; i < (HALO ? 2 * NA : 0); ++i) hvoff[i] = ~0u;
                return;
            }
            const int tile = xcd_remap(vt, ntiles);
            const int m0h = (tile / tiles_n) * kPM;
            const int pbase = padded(m0h) - (hW + 3), rr0 = padded_row(m0h) - 1;
#pragma unroll
            for (int i = 0; i < (HALO ? 2 * NA : 0); ++i) {
                const int j = ((i >> 1) * 8 + (i & 1) * 4 + pw) * 8 + rloc, pp = pbase + j, pc = pp < 0 ? 0 : pp;
                const int rr = fdiv(pc, hd.wp), xx = pc - rr * hWp;
                const int bb = fdiv(rr, hd.hp), yy = rr - bb * hHp;
                const bool ok = pp >= 0 && xx >= 1 && xx <= hW && yy >= 1 && yy <= hH && bb < g.B;
                const int u = j - 2 * (rr - rr0);   // swizzle key (conv_pipe.h: advances by exactly 1 from pixel to pixel across row ends)
                const int chunk = slot ^ ((u >> 1) & 7);
                hvoff[i] = ok ? (unsigned)((((bb * hH + yy - 1) * hW + xx - 1) * ck + chunk * 8) * 2) : ~0u;
            }
        };
        auto halo_issue = [&](int t) __attribute__((always_inline)) {   // (t is a compile-time constant at every call site): this producer's two pieces of K step t
#pragma unroll
            for (int h = 0; h < 2; ++h)
                buf_load16_lds(rsrc_x, tail0 + hl_buf * kHaloB + (t * 8 + h * 4 + pw) * kSpecPieceB, hvoff[HALO ? 2 * t + h : 0],
                               (unsigned)(hl_slice * (kBK * 2)));
        };
        auto halo_advance = [&]() __attribute__((always_inline)) {
            hl_buf ^= 1;
            if ((hl_slice + 1) * kBK >= ck) {
                hl_slice = 0;
                hl_vt += gridDim.x;
                halo_setup(hl_vt);
            } else {
                ++hl_slice;
            }
        };

        // ---- the handed-over store pass.  Lane l of producer pw holds, of every piece it owns, row 4 P + (l >> 4) and the 8 channels of
        // chunk cidx = (l & 15) ^ ((4 pw + (l >> 4)) & 15): the same channels in every piece of every tile of a column block
        const int prow = lane >> 4;
        const int cidx = (lane & 15) ^ ((4 * pw + prow) & 15);
        u16x8 rb[16];                       // the staged tile's pieces of this producer (read back at the next tile's first K step)
        u16x8 zb[bwd ? 8 : 1];              // data gradient: z at the offsets of half a staged tile (requested three K steps ahead of their use)
        float s0[8], s1[8], mu[8], sc[8], be[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) s0[q] = s1[q] = 0.f, mu[q] = sc[q] = be[q] = 0.f;
        int pm0 = 0, pn0 = 0;               // the staged tile
        bool pending = false;
        auto piece_off = [&](int i) __attribute__((always_inline)) -> unsigned {   // output element offset of this lane's 16 B of piece i (~0: a row past M)
            int mb = pm0;
            LP_OPAQUE(mb);   // (computed where it is used: hoisted, the 16 offsets - and the 64-bit addresses made of them - of a tile live in registers for its whole K loop)
            const int m = mb + (4 * i + pw) * 4 + prow;
            return m < M ? (unsigned)out_row(m, lat, div_img, div_row, full_h, full_w) * (unsigned)ep.ldo + (unsigned)(pn0 + cidx * 8) : ~0u;
        };
        auto readback = [&](int st, int hb) __attribute__((always_inline)) {
            const unsigned char* ra = region_a(st, hb) + lane * 16;
            const unsigned char* rbp = region_b(st) + lane * 16;
#pragma unroll
            for (int i = 0; i < 12; ++i) rb[i] = lds_read16_async(ra + (4 * i + pw) * kSpecPieceB);
#pragma unroll
            for (int i = 12; i < 16; ++i) rb[i] = lds_read16_async(rbp + (4 * (i - 12) + pw) * kSpecPieceB);
            LP_SPEC_TOUCH4(rb[0], rb[1], rb[2], rb[3]);
            LP_SPEC_TOUCH4(rb[4], rb[5], rb[6], rb[7]);
            LP_SPEC_TOUCH4(rb[8], rb[9], rb[10], rb[11]);
            LP_SPEC_TOUCH4(rb[12], rb[13], rb[14], rb[15]);
        };
        auto z_issue = [&](int half) __attribute__((always_inline)) {   // (compile-time half) data gradient: z at the offsets of pieces 8 half .. 8 half + 7
#pragma unroll
            for (int e = 0; e < (bwd ? 8 : 0); ++e) {
                const unsigned o = piece_off(8 * half + e);
                zb[bwd ? e : 0] = load8(ep.bn_z + (o != ~0u ? o : (unsigned)(pn0 + cidx * 8)));   // (rows past M: any valid address)
            }
        };
        auto store_group = [&](int grp) __attribute__((always_inline)) {   // (compile-time grp) pieces 4 grp .. 4 grp + 3: mask (data gradient), store, fused sums
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = 4 * grp + e;
                const unsigned o = piece_off(i);
                if (o == ~0u) continue;
                u16x8 w = rb[i];
                float zc[8];
                if (bwd) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        zc[q] = bf16_to_f32(zb[bwd ? (i & 7) : 0][q]) - mu[q];
                        // (the mask after the rounding: a masked value is exactly 0 either way - conv_pipe_kernel masks the fp32 value)
                        if (!(fmaf(zc[q], sc[q], be[q]) > 0x1p-134f)) w[q] = 0;
                    }
                }
                store8(ep.out_bf16 + o, w);
                if (want_stats) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const float vr = bf16_to_f32(w[q]);
                        s0[q] += vr;
                        s1[q] = fmaf(vr, bwd ? zc[q] : vr, s1[q]);   // backward: x invstd at the flush
                    }
                }
            }
        };
        int st_n0 = -1, st_seg_off = 0;
        auto stats_begin = [&](int n0, int seg_off) __attribute__((always_inline)) {   // the sums now belong to column block n0 of segment seg_off
            st_n0 = n0;
            st_seg_off = seg_off;
            if (bwd) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int n = n0 + cidx * 8 + q;
                    mu[q] = ep.bn_mean[seg_off + n];
                    sc[q] = ep.bn_invstd[seg_off + n] * ep.bn_gamma[n];
                    be[q] = ep.bn_beta[n];
                }
            }
        };
        // flush: lanes l ^ {0, 17, 34, 51} of a wave hold the same channels; the four producers' rows meet in `scratch` (4 KB, dead LDS)
        auto flush_write = [&](float* scratch) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                s0[q] += __shfl_xor(s0[q], 17, 64);
                s1[q] += __shfl_xor(s1[q], 17, 64);
                s0[q] += __shfl_xor(s0[q], 34, 64);
                s1[q] += __shfl_xor(s1[q], 34, 64);
            }
            if (lane < 16) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    scratch[(pw * 2 + 0) * BN + cidx * 8 + q] = s0[q];
                    scratch[(pw * 2 + 1) * BN + cidx * 8 + q] = s1[q];
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) s0[q] = s1[q] = 0.f;
        };
        auto flush_emit = [&](const float* scratch) __attribute__((always_inline)) {   // 256 producer threads = 2 sums x 128 channels, the producers' rows in a fixed order
            const int t = (pw << 6) | lane, comp = t >> 7, cl = t & 127;
            float v = 0.f;
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4) v += scratch[(w4 * 2 + comp) * BN + cl];
            if (bwd && comp == 1) v *= ep.bn_invstd[st_seg_off + st_n0 + cl];   // sum dy (z - mean)  ->  sum dy xhat
            stats_emit(ep, 2 * st_seg_off + comp * N + st_n0 + cl, v);
        };

        // ---- the walk
        if (HALO) {
            halo_setup(hl_vt);
#pragma unroll
            for (int t = 0; t < (HALO ? NA : 0); ++t) halo_issue(t);
            halo_advance();
        }
#pragma unroll 1
        for (int st = 0; st < 2; ++st) load_step(st);
        int cur = 0, mh_tap = 0, mh_buf = 0;
        int prev_n0 = -1, prev_seg = 0;
        for (int vt = blockIdx.x; vt < ntiles; vt += gridDim.x) {
            int m0, n0, seg_off;
            tile_of(vt, m0, n0, seg_off);
            for (int kt = 0; kt < KT; ++kt) {
                // this producer's loads of the current step have landed: the wait counts the YOUNGER loads - the next step's ring pieces, and
                // in the two steps behind each of a data gradient's z requests (8 loads, issued at steps 0 and 3 behind that step's ring
                // loads) those 8 as well (HALO: the two halo pieces a step may also have issued are left out of the count: conservative)
                if (!kFwd && pending && (kt == 1 || kt == 2 || kt == 4 || kt == 5)) {
                    if (HALO) LP_WAIT_VM(12);
                    else LP_WAIT_VM(20);
                } else {
                    if (HALO) LP_WAIT_VM(4);
                    else LP_WAIT_VM(12);
                }
                LP_RAW_BARRIER();
                const int nxt = cur == 0 ? 2 : cur - 1;   // (cur + 2) % 3: the stage consumed at the previous step
                if (pending && kt == 0) readback(nxt, mh_buf ^ 1);   // the staged tile sits where the next loads go: out of LDS first
                if (HALO) {
                    switch (mh_tap) {   // (workgroup-uniform) one eighth-KB-pair of the next halo image per K step, taps 0 .. NA - 1
                    case 0: halo_issue(0); break;
                    case 1: halo_issue(1); break;
                    case 2: halo_issue(2); break;
                    case 3: halo_issue(3); break;
                    case 4: halo_issue(4); break;
                    case 5: halo_issue(5); break;
                    default: break;
                    }
                }
                load_step(nxt);
                if (pending) {   // the store pass of the previous tile, spread over this tile's first K steps
                    if (kFwd) {
                        switch (kt) {
                        case 0: store_group(0); break;
                        case 1: store_group(1); break;
                        case 2: store_group(2); break;
                        case 3: store_group(3); pending = false; break;
                        default: break;
                        }
                    } else {
                        // z of the tile's first half is requested at step 0; of the second at step 3, once the first has been consumed (two
                        // register sets do not fit beside the staged tile) - each three K steps ahead of its use: the wait at the top of
                        // step 3 / 6 is for ring loads issued AFTER the request, so the data has landed by then (loads retire in order)
                        switch (kt) {
                        case 0: z_issue(0); break;
                        case 3: store_group(0); store_group(1); z_issue(1); break;
                        case 6: store_group(2); store_group(3); pending = false; break;
                        default: break;
                        }
                    }
                }
                cur = cur == 2 ? 0 : cur + 1;
                if (HALO && ++mh_tap == 9) {
                    mh_tap = 0;
                    mh_buf ^= 1;
                    halo_advance();
                }
            }
            LP_RAW_BARRIER();   // E: the consumers are done with the stage / halo image consumed last
            if (want_stats && prev_n0 >= 0 && (n0 != prev_n0 || seg_off != prev_seg)) {   // (workgroup-uniform) the sums of the tiles up to the previous one leave
                float* scratch = reinterpret_cast<float*>(region_b(cur == 0 ? 2 : cur - 1));
                flush_write(scratch);
                LP_RAW_BARRIER();   // F1
                flush_emit(scratch);
                LP_RAW_BARRIER();   // F2: the consumers may now stage the tile over the scratch
            }
            // (the consumers stage the tile now; it is read back at the next K step's barrier, or at the drain below)
            if (want_stats && (n0 != prev_n0 || seg_off != prev_seg)) stats_begin(n0, seg_off);
            else if (bwd && prev_n0 < 0) stats_begin(n0, seg_off);
            prev_n0 = n0, prev_seg = seg_off;
            pm0 = m0, pn0 = n0;
            pending = true;
        }
        // ---- drain: the last tile's store pass, then the last sums
        LP_RAW_BARRIER();   // X: the last staged tile is visible
        if (pending) {
            readback(cur == 0 ? 2 : cur - 1, mh_buf ^ 1);
            if (kFwd) {
                store_group(0);
                store_group(1);
                store_group(2);
                store_group(3);
            } else {
                z_issue(0);
                LP_WAIT_VM(0);
                store_group(0);
                store_group(1);
                z_issue(1);
                LP_WAIT_VM(0);
                store_group(2);
                store_group(3);
            }
        }
        if (want_stats) {
            float* scratch = reinterpret_cast<float*>(region_b(cur == 0 ? 2 : cur - 1));
            flush_write(scratch);
            LP_RAW_BARRIER();   // F1
            flush_emit(scratch);
        }
        LP_WAIT_VM(0);   // the ring's last (empty) loads still target this workgroup's LDS
        return;
    }

    // =========================================================================================================================
    // CONSUMER: fragment reads + MFMAs of a 128 x 64 wave tile, the staging of the finished tile, nothing else
    // =========================================================================================================================
    const int wm = wave & 1, wn = wave >> 1;
    const int fr = lane & 31, fg = lane >> 5;
    const int fsw = (fr >> 1) & 7;
    unsigned koff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) koff[kk] = (unsigned)(((kk * 2 + fg) ^ fsw) * 16);
    const unsigned a_row = (unsigned)((wm * 128 + fr) * kPRowB);
    const unsigned b_row = (unsigned)(kStageA + (wn * 64 + fr) * kPRowB);
    f32x16 acc[4][2];
    int ploc[4] = {0, 0, 0, 0}, uloc[4] = {0, 0, 0, 0};
    int cur = 0, mh_tap = 0, mh_buf = 0;
    int prev_n0 = -1, prev_seg = 0;
    LP_SETPRIO(1);   // (MI355X_MICROARCH.md: the matrix waves win the issue arbitration against their load partners)
    for (int vt = blockIdx.x; vt < ntiles; vt += gridDim.x) {
        int m0, n0, seg_off;
        tile_of(vt, m0, n0, seg_off);
        if (HALO) {
            const int pbase = padded(m0) - (hW + 3), rr0 = padded_row(m0) - 1;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int m = m0 + wm * 128 + mt * 32 + fr, mc = m < M ? m : M - 1;   // (rows past M: any staged row, the result is not stored)
                ploc[mt] = padded(mc) - pbase;
                uloc[mt] = ploc[mt] - 2 * (padded_row(mc) - rr0);
            }
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;
        for (int kt = 0; kt < KT; ++kt) {
            LP_RAW_BARRIER();
            const unsigned char* sb = smem + cur * kStage;
            const unsigned char* ha[4] = {smem, smem, smem, smem};
            unsigned hk[4][4] = {};
            if (HALO) {   // tap (r, s) of the slice: a constant row offset in padded raster coordinates
                const int tr = mh_tap / 3, ts = mh_tap - tr * 3;
                const int dp = (tr - 1) * hWp + (ts - 1), du = (tr - 1) * hW + (ts - 1);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int row = ploc[mt] + (MODE == kModeDgrad ? -dp : dp);
                    const int sw = ((uloc[mt] + (MODE == kModeDgrad ? -du : du)) >> 1) & 7;
                    ha[mt] = tail0 + mh_buf * kHaloB + row * kPRowB;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) hk[mt][kk] = (unsigned)(((kk * 2 + fg) ^ sw) * 16);
                }
            }
            bf16x8 a[2][4], b[2][2];
            auto fetch = [&](int kk, int set) __attribute__((always_inline)) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    if (HALO) a[set][mt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u16x8*>(ha[mt] + hk[mt][kk]));
                    else a[set][mt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u16x8*>(sb + a_row + mt * (32 * kPRowB) + koff[kk]));
                }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    b[set][nt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u16x8*>(sb + b_row + nt * (32 * kPRowB) + koff[kk]));
            };
            fetch(0, 0);
#pragma unroll
            for (int kk = 0; kk < kBK / 16; ++kk) {
                if (kk + 1 < kBK / 16) fetch(kk + 1, (kk + 1) & 1);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)   // roles swapped: D[channel][pixel]
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[kk & 1][nt], a[kk & 1][mt], acc[mt][nt], 0, 0, 0);
            }
            cur = cur == 2 ? 0 : cur + 1;
            if (HALO && ++mh_tap == 9) {
                mh_tap = 0;
                mh_buf ^= 1;
            }
        }
        LP_RAW_BARRIER();   // E: every consumer is done with the stage / halo image this tile consumed last
        if (want_stats && prev_n0 >= 0 && (n0 != prev_n0 || seg_off != prev_seg)) {   // the producers flush their sums through the dead LDS first
            LP_RAW_BARRIER();   // F1
            LP_RAW_BARRIER();   // F2
        }
        prev_n0 = n0, prev_seg = seg_off;
        // ---- stage the tile: lane (pixel fr, half fg) holds for block (mt, nt) the channels wn 64 + nt 32 + 8 j + 4 fg + (0..3) in acc[mt][nt][4 j .. 4 j + 3]
        {
            const int stl = cur == 0 ? 2 : cur - 1;
            unsigned char* ra = region_a(stl, mh_buf ^ 1);
            unsigned char* rbp = region_b(stl);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int row = wm * 128 + mt * 32 + fr;
                unsigned char* base = (wm == 1 && mt >= 2) ? rbp + ((row >> 2) - 48) * kSpecPieceB : ra + (row >> 2) * kSpecPieceB;
                base += (row & 3) * 256 + fg * 8;
                const int key = row & 15;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
                        const u32x2_t p = {pack_bf16x2(acc[mt][nt][4 * j], acc[mt][nt][4 * j + 1]),
                                           pack_bf16x2(acc[mt][nt][4 * j + 2], acc[mt][nt][4 * j + 3])};
                        *reinterpret_cast<u32x2_t*>(base + (((wn * 8 + nt * 4 + j) ^ key) << 4)) = p;
                    }
            }
            LP_WAIT_LGKM0();   // the staged tile is in LDS before this wave arrives at the barrier that publishes it
        }
    }
    LP_RAW_BARRIER();   // X: the last staged tile is visible to the producers
    if (want_stats) LP_RAW_BARRIER();   // F1 of the final flush
}

}  // namespace lp
