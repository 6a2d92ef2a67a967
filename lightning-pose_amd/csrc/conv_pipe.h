// Pipelined implicit-GEMM convolution (round 3): forward and data-gradient of the ResNet-50 trunk on a direct-to-LDS operand ring.
//
// Why a second kernel next to conv_igemm_kernel (conv.hip): that kernel fetches ONE K step ahead into registers, so every
// K step of every tile waits a full (loaded) memory latency - the 1x1 layers ran at ~3 TB/s of a 6.3 TB/s part and their
// data gradients spent 53 % of their wave cycles parked on memory (profiles/archive/r02_pmc_mfma.json).  Here the operands go
// global -> LDS with `buffer_load_dwordx4 ... lds` (no staging registers, no ds_write pass) into a ring of THREE stages:
// while the MFMAs of K step g run, the loads of steps g+1 and g+2 are in flight (96 KB per CU; profiles/probe/glds_probe.hip
// measures 6.0 TB/s read-only and 5.4-5.5 TB/s with an output stream for exactly this ring), and the ring never drains between
// tiles - the loader walks the persistent tile list two K steps ahead of the MFMAs, across tile boundaries.
//
//   workgroup   512 threads = 8 waves (4 along M x 2 along N), ONE per CU (LDS: 3 x 48 KB), tile 256 x BN (BN = 128 or 64), K step 64
//   operand LDS image  [rows][64 k] bf16 = 128-B rows without padding (the LDS side of a direct load is lane-linear); the 16-B chunk
//               c of row r sits at position c ^ ((r >> 1) & 7), applied on the SOURCE address of the load - ds_read_b128 fragment
//               reads are then conflict-free (every 16-lane service group sees 8 distinct positions per row parity)
//   MFMA        v_mfma_f32_32x32x16_bf16 with the operand ROLES swapped (weights as A, pixels as B): a lane then owns ONE pixel and
//               runs of 4 consecutive output channels, so the store pass packs 8-B pieces instead of 64 scalar LDS writes per lane
//   store pass  per WAVE through a private corner of the stage the tile consumed last (no workgroup barrier inside it): 32 pixels x
//               its channels, bf16 (forward) or fp32 (data gradient: addend, ReLU masks and BatchNorm terms are applied before the
//               single rounding, exactly as conv_igemm_kernel does), read back as 16-B row pieces for full-line global stores
//   BatchNorm sums  per-THREAD running sums over all tiles a workgroup walks (same columns every tile), folded across lanes and
//               waves only when the column block / segment changes or the walk ends: no per-tile LDS reduction, no barriers
//
// Numerics: the K order of the accumulation and every rounding point equal conv_igemm_kernel's, so outputs are bit-identical to it
// (tests/test_emu_conv_pipe.py); only the fp32 summation order of the fused BatchNorm sums differs.
//
// Synchronisation (one raw s_barrier per K step): a wave waits for ITS loads of step g with a counted `s_waitcnt vmcnt(n)` (n =
// the loads of step g+1, which stay in flight; loads retire in order, and stores still pending only make the wait conservative),
// then the barrier publishes the stage to all waves AND certifies that everyone is done reading stage (g-1) % 3, which the loads
// of step g+2 issued right after the barrier overwrite.
#pragma once

#include <type_traits>

namespace lp {

#ifndef LP_HALO_KEY_ROW
#define LP_HALO_KEY_ROW 0   // (A/B builds: 1 = round 3's swizzle key of the HALO form, the halo row itself)
#endif
#ifndef LP_HALO_STAGES
#define LP_HALO_STAGES 3   // weight-ring stages of the HALO form (A/B builds: 4 = three K steps of weight loads in flight, 160 KB of LDS)
#endif
#ifndef LP_FWD_CORNER_PAD
#define LP_FWD_CORNER_PAD 0
#endif
#ifndef LP_PIPE_SPREAD
#define LP_PIPE_SPREAD 1   // (A/B builds: 0 issues a K step's loads in one burst after the barrier)
#endif
constexpr bool kSpread = LP_PIPE_SPREAD != 0;
constexpr int kPM = 256;   // tile rows (pixels)
constexpr int kPRowB = 128;  // bytes per staged operand row (64 k x bf16)

#if defined(__HIP_DEVICE_COMPILE__)
#define LP_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define LP_RAW_BARRIER()                   \
    do {                                   \
        asm volatile("" ::: "memory");     \
        __builtin_amdgcn_s_barrier();      \
        asm volatile("" ::: "memory");     \
    } while (0)
// wait until at most n LDS / scalar-memory operations of this wave are outstanding, and make the fragments named after it depend on the
// wait (they were filled by lds_read_tr16_async: without the "+v" ties the compiler could move their first use above the wait)
#define LP_WAIT_LGKM_TOUCH3(n, a, b, c) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a), "+v"(b), "+v"(c)::"memory")
#define LP_WAIT_LGKM_TOUCH4(n, a, b, c, d) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory")
#else
#define LP_WAIT_VM(n) ((void)0)
#define LP_RAW_BARRIER() __syncthreads()
#define LP_WAIT_LGKM_TOUCH3(n, a, b, c) ((void)0)
#define LP_WAIT_LGKM_TOUCH4(n, a, b, c, d) ((void)0)
#endif

// 16-B store (a non-temporal form was measured in round 3 and dropped: profiles/archive/r03b_bench_ntstore_1.json.log)
__device__ __forceinline__ void store8(unsigned short* p, u16x8 v) { *reinterpret_cast<u16x8*>(p) = v; }

// EK = what the data gradient's store pass reads back (its own instantiation each, so the read-back registers of one form are not
// allocated in the others; anything else goes to conv_igemm_kernel):
//   kEkNone  forward
//   kEkZ     BatchNorm backward sums + ReLU mask recomputed from the pre-normalisation tensor z (conv3 / conv2 of every block)
//   kEkAZB   addend + z (sums) + 1-bit ReLU mask (conv1 of the blocks behind an identity shortcut)
//   kEkPlain no BatchNorm fused: optional addend, optional bf16 activation as the ReLU mask (conv1 / projection shortcut of the first
//            block of a layer, whose input gradient has two writers)
//   kEkInfer forward with + residual and ReLU in the store pass (inference, BatchNorm folded into weights and bias: lp_conv_fwd_act)
//   kEkPB    kEkPlain with the ReLU mask at 1 bit per element (lp_conv_dgrad_bits, round 5: the two writers of a layer's first block read
//            57 MB of mask bits instead of re-reading the block input's 906 MB activation)
//   kEkGeluBwd  forward-mode GEMM whose store pass multiplies the (bf16-rounded) product by GELU'(u), u read at the output's own offsets
//            as `addend` (lp_gemm_nt_gelu_bwd, round 6: the data gradient of a ViT block's fc2 leaves as the gradient of fc1's OUTPUT - the
//            stand-alone GELU backward's write and read of the activation gradient are gone; the column sums = fc1's bias gradient)
//   kEkGeluFwd  forward GEMM that also writes GELU of its (bias-added, bf16-rounded) output to ep.out2_bf16 (lp_gemm_nt_gelu_fwd: a ViT block's
//            fc1 leaves with its activation; the stand-alone GELU pass's read of the pre-activation is gone)
enum { kEkNone = 0, kEkZ = 1, kEkAZB = 2, kEkPlain = 3, kEkInfer = 4, kEkPB = 5, kEkGeluBwd = 6, kEkGeluFwd = 7 };

// HALO (3x3, stride 1, pad 1 - conv2 of every identity-stride block, forward and data gradient): the pixel operand is not fetched per
// filter tap.  The ring above re-reads every activation row 9 times from L2 (once per tap: 27.7 us per tap on layer1's 64-channel layers,
// DESIGN.md section 10.6); here a tile's input neighbourhood - its 256 pixels plus one image row and one pixel either side, in PADDED
// raster coordinates P = ((b (H + 2) + y + 1) (W + 2) + x + 1), so that the zero border is part of the image and a tap is a constant row
// offset - is staged ONCE per 64-channel slice ("halo image", kHaloRows x 128 B, two of them: the next slice / the next tile's first slice
// loads while this one is consumed) and the 9 taps read it at row offsets (r - 1) (W + 2) + (s - 1).  Only the weights still ride the
// 3-stage ring (BN x 128 B per K step).  The K loop runs slice-major (all 9 taps of a channel slice, then the next slice): for more than
// 64 channels the fp32 accumulation order differs from conv_igemm_kernel's tap-major order (outputs agree to fp32 reassociation, not
// bit for bit; with 64 channels they are bit-identical).
struct HaloDivs {
    FastDiv h, wp, hp;   // image height, padded width W + 2, padded height H + 2
};

template <int BN, int MODE, int EK, bool HALO = false>
__global__ __launch_bounds__(512) void conv_pipe_kernel(const unsigned short* __restrict__ X, const unsigned short* __restrict__ Wt,
                                                        unsigned x_bytes, unsigned w_bytes, ConvGeom g, Lattice lat, FastDiv div_img,
                                                        FastDiv div_row, int M, int N, int K, int tiles_n, int ntiles, ConvEpilogue ep,
                                                        HaloDivs hd) {
    static_assert((MODE == kModeFwd && (EK == kEkNone || EK == kEkInfer || EK == kEkGeluBwd || EK == kEkGeluFwd)) ||
                      (MODE == kModeDgrad && EK != kEkNone && EK != kEkInfer && EK != kEkGeluBwd && EK != kEkGeluFwd),
                  "trunk convolutions only");
    constexpr int NT = BN / 64;                  // 32-channel MFMA blocks per wave along N (wave tile 64 pixels x NT*32 channels)
    constexpr int NBL = BN / 64;                 // weight rows each thread stages per K step
    constexpr int kStageA = HALO ? 0 : kPM * kPRowB, kStageB = BN * kPRowB, kStage = kStageA + kStageB;
    constexpr int kHaloRows = BN == 64 ? 512 : 384;   // rows of a halo image (host-checked against the geometry: pipe_halo_rows)
    constexpr int kHaloB = kHaloRows * kPRowB, NA = kHaloRows / 64;   // bytes; direct-to-LDS loads per thread and halo image
    // Ring depth.  Per-tap ring: 3 stages of 48 KB (two K steps of loads in flight) is what LDS holds.  HALO form: a stage is only the weights
    // (16 / 8 KB), so FOUR stages would fit beside the two halo images (160 KB exactly), three K steps of weight loads in flight: measured in
    // round 5 (-DLP_HALO_STAGES=4, profiles/r05h_halo_stages.txt) - no difference in any 3x3 layer, so their K step is not waiting for its weights.
    constexpr int NST = HALO ? LP_HALO_STAGES : 3;
    static_assert(NST == 3 || NST == 4, "ring depth");
    __shared__ __attribute__((aligned(16))) unsigned char smem[NST * kStage + (HALO ? 2 * kHaloB : 0)];
    unsigned char* const halo0 = smem + NST * kStage;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;

    const buf_rsrc rsrc_x = make_buf_rsrc(X, x_bytes), rsrc_w = make_buf_rsrc(Wt, w_bytes);
    const int rows_y = lat.nh, rows_x = lat.nw;
    const int full_h = (MODE == kModeDgrad) ? g.Hi : g.Ho;
    const int full_w = (MODE == kModeDgrad) ? g.Wi : g.Wo;
    const int ck = (MODE == kModeDgrad) ? g.Co : g.Ci;   // channels of the gathered tensor
    const int src_h = (MODE == kModeDgrad) ? g.Ho : g.Hi;
    const int src_w = (MODE == kModeDgrad) ? g.Wo : g.Wi;
    const bool halved = (MODE == kModeDgrad) && g.stride == 2;
    const int KT = K / kBK;
    const int ldw = g.R * g.S * ck;              // elements per weight row (the FULL filter)

    // ---- loader: this thread's 4 pixel rows and NBL weight rows of a stage; lane -> (row in the wave's 8-row piece, 16-B slot)
    const int rloc = lane >> 3, slot = lane & 7;
    const int lrow = wave * 8 + rloc;                      // row inside a 64-row pass
    const int lchunk = slot ^ ((lrow >> 1) & 7);           // the K chunk this lane fetches (position `slot` holds chunk slot ^ swizzle)
    int py[4], px[4];
    unsigned rowoff[4], vmask[4], voff[4], wrow[NBL];
    int tir = 0, tis = 0, tc = 0;
    unsigned wtap = 0;
    int ld_vt = blockIdx.x, ld_kt = 0;

    auto setup = [&](int vt) {
        if (vt >= ntiles) {   // past the end of the walk: the ring keeps turning on loads that fetch nothing (zeros), so the counted waits stay valid
#pragma unroll
            for (int i = 0; i < 4; ++i) vmask[i] = 0u, rowoff[i] = 0u;
#pragma unroll
            for (int i = 0; i < NBL; ++i) wrow[i] = ~0u;
            tir = tis = tc = 0;
            return;
        }
        const int tile = xcd_remap(vt, ntiles);
        const int tm_ = tile / tiles_n;
        const int m0 = tm_ * kPM, n0 = (tile - tm_ * tiles_n) * BN;
#pragma unroll
        for (int i = 0; i < (HALO ? 0 : 4); ++i) {   // (HALO: the pixel operand comes from the halo image, halo_setup below)
            const int m = m0 + lrow + 64 * i;
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
        for (int i = 0; i < NBL; ++i) wrow[i] = (unsigned)((n0 + lrow + 64 * i) * ldw + lchunk * 8) * 2u;   // (N % BN == 0: host-checked)
        tir = tis = tc = 0;
    };

    // one K step of operands -> stage `st`: 4 + NBL direct-to-LDS loads per thread, each wave instruction fills 8 rows x 128 B.
    // prep_step() does the arithmetic (and moves the loader on, into the next tile if need be); issue_load(i) is load i of that step -
    // the MFMA loop spreads them between its k-slices, because all 8 waves issuing 6 loads each right after the barrier queue up at the
    // CU's one address path (48 KB at 64 B/clk = 768 cycles) before any MFMA starts
    unsigned is_soff_a = 0, is_soff_b = 0, is_w[NBL];
    unsigned char* is_dst = smem;
    auto prep_step = [&](int st) {
        if (HALO) {      // slice-major K order: tir = tap (0 .. 8), tc = first channel of the slice; only the weights are fetched per step
            const int tr = tir / 3, ts = tir - tr * 3;
            is_soff_b = (unsigned)(((tr * g.S + ts) * ck + tc) * 2);
#pragma unroll
            for (int i = 0; i < NBL; ++i) is_w[i] = wrow[i];
            is_dst = smem + st * kStage + wave * (8 * kPRowB);
            if (++tir == 9) tir = 0, tc += kBK;
            return;
        }
        if (tc == 0) {   // entering a filter tap: its per-row offsets (an invalid tap gets ~0 -> the range check returns zeros)
            const int tr = lat.r0 + lat.rstep * tir, ts = lat.s0 + lat.sstep * tis;
            const int qr = halved ? (tr >> 1) : tr, qs = halved ? (ts >> 1) : ts;
            const unsigned tapoff_b = (unsigned)(((MODE == kModeDgrad) ? -(qr * src_w + qs) : (qr * src_w + qs)) * ck) * 2u;
            const int tap = tir * lat.ns + tis;
            wtap = (unsigned)((tr * g.S + ts) * ck) * 2u;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned ok = (vmask[i] >> tap) & 1u;
                voff[i] = (rowoff[i] + tapoff_b) | (ok - 1u);
            }
        }
        is_soff_a = (unsigned)tc * 2u;
        is_soff_b = wtap + is_soff_a;
#pragma unroll
        for (int i = 0; i < NBL; ++i) is_w[i] = wrow[i];
        is_dst = smem + st * kStage + wave * (8 * kPRowB);
        tc += kBK;
        if (tc >= ck) {
            tc = 0;
            if (++tis == lat.ns) {
                tis = 0;
                ++tir;
            }
        }
    };
    auto issue_load = [&](int i) {   // (i is a compile-time constant at every call site)
        if (HALO) buf_load16_lds(rsrc_w, is_dst + (i % NBL) * (64 * kPRowB), is_w[i % NBL], is_soff_b);
        else if (i < 4) buf_load16_lds(rsrc_x, is_dst + i * (64 * kPRowB), voff[i], is_soff_a);
        else buf_load16_lds(rsrc_w, is_dst + kStageA + (i - 4) * (64 * kPRowB), is_w[i - 4], is_soff_b);
    };
    auto advance_tile = [&]() {      // after the step's last load has been issued: the loader may cross into the next tile of the walk
        if (++ld_kt == KT) {
            ld_kt = 0;
            ld_vt += gridDim.x;
            setup(ld_vt);
        }
    };
    auto load_step = [&](int st) {
        prep_step(st);
#pragma unroll
        for (int i = 0; i < (HALO ? NBL : 4 + NBL); ++i) issue_load(i);
        advance_tile();
    };

    // ---- HALO loader: one halo image (a tile's padded-raster neighbourhood x one 64-channel slice) ahead of the MFMAs
    const int hW = g.Wi, hH = g.Hi, hWp = g.Wi + 2, hHp = g.Hi + 2;   // (stride 1, "same": input and output grids coincide)
    unsigned hvoff[HALO ? NA : 1];
    int hl_vt = blockIdx.x, hl_slice = 0, hl_buf = 0;
    auto padded = [&](int m) {   // padded raster coordinate of pixel row m
        const int rm = fdiv(m, div_row), bm = fdiv(rm, hd.h);
        return (rm + 1 + 2 * bm) * hWp + (m - rm * hW) + 1;
    };
    auto padded_row = [&](int m) {   // ... and the padded image row it lies in
        const int rm = fdiv(m, div_row), bm = fdiv(rm, hd.h);
        return rm + 1 + 2 * bm;
    };
    // Swizzle key of a halo row (round 4).  A wave's 32-pixel fragment window crosses an image-row end every W pixels, where the padded
    // raster skips the two border columns (+3 instead of +1): keyed on the halo row itself - (row >> 1) & 7, conflict-free for CONSECUTIVE
    // rows - the ds_read_b128 lane groups {0-3, 12-15, 20-27} ... then meet 2-way bank conflicts after every jump (W = 24 / 12: one to three
    // per window; SQ_LDS_BANK_CONFLICT was 8 - 10 % of the HALO launches' cycles, profiles/archive/r03_pmc_mfma.json).  Keyed on
    // u = row - 2 x (padded image rows since the tile's first), u advances by exactly 1 from pixel to pixel across row ends, and row - u is
    // even, so (bank half, chunk position) = u mod 16 stays distinct over any 16 pixels in a row.  A tap (r, s) moves u by (r - 1) W + (s - 1).
    auto halo_setup = [&](int vt) {
        if (vt >= ntiles) {
#pragma unroll
            for (int i = 0; i < (HALO ? NA : 0); ++i) hvoff[i] = ~0u;
            return;
        }
        const int tile = xcd_remap(vt, ntiles);
        const int m0h = (tile / tiles_n) * kPM;
        const int pbase = padded(m0h) - (hW + 3), rr0 = padded_row(m0h) - 1;   // (pbase lies in the padded row above the tile's first pixel)
#pragma unroll
        for (int i = 0; i < (HALO ? NA : 0); ++i) {
            const int j = i * 64 + lrow, pp = pbase + j, pc = pp < 0 ? 0 : pp;
            const int rr = fdiv(pc, hd.wp), xx = pc - rr * hWp;
            const int bb = fdiv(rr, hd.hp), yy = rr - bb * hHp;
            const bool ok = pp >= 0 && xx >= 1 && xx <= hW && yy >= 1 && yy <= hH && bb < g.B;
            const int u = LP_HALO_KEY_ROW ? j : j - 2 * (rr - rr0);
            const int chunk = slot ^ ((u >> 1) & 7);
            hvoff[i] = ok ? (unsigned)((((bb * hH + yy - 1) * hW + xx - 1) * ck + chunk * 8) * 2) : ~0u;   // border / beyond the batch: zeros
        }
    };
    auto halo_issue = [&](int i) {   // (i is a compile-time constant at every call site)
        buf_load16_lds(rsrc_x, halo0 + hl_buf * kHaloB + i * (64 * kPRowB) + wave * (8 * kPRowB), hvoff[HALO ? i : 0], (unsigned)(hl_slice * (kBK * 2)));
    };
    auto halo_advance = [&]() {      // the next slice of the tile, or the first slice of the workgroup's next tile
        hl_buf ^= 1;
        if ((hl_slice + 1) * kBK >= ck) {
            hl_slice = 0;
            hl_vt += gridDim.x;
            halo_setup(hl_vt);
        } else {
            ++hl_slice;
        }
    };

    // ---- MFMA side: fragment addresses inside a stage (16-B chunk (2 kk + g) of row r sits at position chunk ^ ((r >> 1) & 7))
    const int fr = lane & 31, fg = lane >> 5;
    const int fsw = (fr >> 1) & 7;
    unsigned koff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) koff[kk] = (unsigned)(((kk * 2 + fg) ^ fsw) * 16);
    const unsigned a_row = (unsigned)((wm * 64 + fr) * kPRowB);                   // pixel rows of this wave
    const unsigned b_row = (unsigned)(kStageA + (wn * (NT * 32) + fr) * kPRowB);  // weight rows of this wave

    f32x16 acc[2][NT];
    // `spread`: issue the prepared step's loads between the k-slices (2 after the first slice's MFMAs have been queued, then 2, 1, 1)
    // `spread`: issue the prepared step's loads between the k-slices (2 after the first slice's MFMAs have been queued, then 2, 1, 1).
    // (Measured and dropped, profiles/archive/r03k_stagger.txt: letting the two waves of a SIMD do their per-step address arithmetic at different
    // points - one before slice 0, the other after slice 1 with its loads in slices 2 - 3 - made every forward layer 15 - 25 % slower.)
    // fragment sets in flight: the forward kernel reads TWO k-slices ahead of its MFMAs (3 register sets; 16 more VGPRs it has), the data
    // gradient - at the register cap because of its read-back batches - one slice ahead (2 sets)
#ifndef LP_PIPE_FRAG_SETS
#define LP_PIPE_FRAG_SETS 2   // (3 = two slices ahead: measured equal within noise, profiles/archive/r03n_fragsets.txt, at 16 more VGPRs)
#endif
    constexpr int NS = (MODE == kModeFwd) ? LP_PIPE_FRAG_SETS : 2;
    // HALO: rows of the tile's pixels inside the halo image (per tile), the halo image the MFMAs read, its tap, and whether the
    // loader still has pieces of the next halo image to issue during this K step
    int ploc[2] = {0, 0}, uloc[2] = {0, 0}, mh_buf = 0, mh_tap = 0;
    auto mma_stage = [&](int st, const bool spread) {
        const unsigned char* sb = smem + st * kStage;
        bf16x8 a[NS][2], b[NS][NT];
        const unsigned char* ha[2] = {smem, smem};
        unsigned hk[2][4] = {};
        if (HALO) {   // tap (r, s) of the slice: a constant row offset in padded raster coordinates (the zero border is part of the image)
            const int tr = mh_tap / 3, ts = mh_tap - tr * 3;
            const int dp = (tr - 1) * hWp + (ts - 1), du = (tr - 1) * hW + (ts - 1);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int row = ploc[mt] + (MODE == kModeDgrad ? -dp : dp);
                const int sw = LP_HALO_KEY_ROW ? (row >> 1) & 7 : ((uloc[mt] + (MODE == kModeDgrad ? -du : du)) >> 1) & 7;
                ha[mt] = halo0 + mh_buf * kHaloB + row * kPRowB;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) hk[mt][kk] = (unsigned)(((kk * 2 + fg) ^ sw) * 16);
            }
        }
        auto fetch = [&](int kk, int set) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                if (HALO) a[set][mt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u16x8*>(ha[mt] + hk[mt][kk]));
                else a[set][mt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u16x8*>(sb + a_row + mt * (32 * kPRowB) + koff[kk]));
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                b[set][nt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u16x8*>(sb + b_row + nt * (32 * kPRowB) + koff[kk]));
        };
#pragma unroll
        for (int kk = 0; kk < NS - 1; ++kk) fetch(kk, kk);
#pragma unroll
        for (int kk = 0; kk < kBK / 16; ++kk) {
            if (kk + NS - 1 < kBK / 16) fetch(kk + NS - 1, (kk + NS - 1) % NS);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)   // roles swapped: D[channel][pixel]
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[kk % NS][nt], a[kk % NS][mt], acc[mt][nt], 0, 0, 0);
            if (spread && HALO) {   // one piece of the next halo image (taps 0 .. NA - 1 of a slice), then this step's weight rows
                if (kk == 0) {
                    switch (mh_tap) {   // (workgroup-uniform)
                    case 0: halo_issue(0); break;
                    case 1: halo_issue(1); break;
                    case 2: halo_issue(2); break;
                    case 3: halo_issue(3); break;
                    case 4: halo_issue(4); break;
                    case 5: halo_issue(5); break;
                    case 6: if (NA > 6) halo_issue(NA > 6 ? 6 : 0); break;
                    case 7: if (NA > 7) halo_issue(NA > 7 ? 7 : 0); break;
                    default: break;
                    }
                } else if (kk == 1) {
                    issue_load(0);
                } else if (kk == 2) {
                    if (NBL == 2) issue_load(1);
                }
            } else if (spread) {
                constexpr int NL = 4 + NBL;
                if (kk == 0) {
                    issue_load(0);
                    issue_load(1);
                } else if (kk == 1) {
                    issue_load(2);
                    if (NL == 6) issue_load(3);
                } else if (kk == 2) {
                    issue_load(NL == 6 ? 4 : 3);
                } else {
                    issue_load(NL - 1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- fused BatchNorm sums: per-THREAD running sums of the 8 channels this thread stores (piece `pc` of the wave's NT*32 channels -
    // the same in every tile of one column block), folded across lanes and waves only when the column block / segment changes
    constexpr bool kFwd = (MODE == kModeFwd);
    constexpr int CP = NT * 4;                  // 8-channel pieces per row of a wave's store chunk
    constexpr int RP = 64 / CP;                 // rows per read-back pass
    const int pc = lane % CP, prow = lane / CP;
    float s0[8], s1[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) s0[q] = s1[q] = 0.f;
    int st_n0 = -1, st_seg_off = 0;
    const bool want_stats = ep.stats_sums != nullptr;
    constexpr bool bwd = (EK == kEkZ || EK == kEkAZB);   // the store pass takes BatchNorm's two backward sums (host: ep.bn_z set)
    // `scratch`: LDS nobody else touches right now (the stage the last finished tile consumed last).  Workgroup-uniform call.
    auto stats_flush = [&](float* scratch) {
        if (st_n0 >= 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
#pragma unroll
                for (int msk = CP; msk < 64; msk <<= 1) {
                    s0[q] += __shfl_xor(s0[q], msk, 64);
                    s1[q] += __shfl_xor(s1[q], msk, 64);
                }
            }
            __syncthreads();   // every wave's staging corner is dead
            if (lane < CP) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    scratch[(wave * 2 + 0) * (NT * 32) + lane * 8 + q] = s0[q];
                    scratch[(wave * 2 + 1) * (NT * 32) + lane * 8 + q] = s1[q];
                }
            }
            __syncthreads();
            if (tid < 2 * BN) {
                const int comp = tid / BN, cl = tid % BN;
                const int wn_ = cl / (NT * 32), c = cl % (NT * 32);
                float t = 0.f;
#pragma unroll
                for (int w4 = 0; w4 < 4; ++w4) t += scratch[((wn_ * 4 + w4) * 2 + comp) * (NT * 32) + c];
                if (bwd && comp == 1) t *= ep.bn_invstd[st_seg_off + st_n0 + cl];   // sum dy (z - mean)  ->  sum dy xhat
                stats_emit(ep, 2 * st_seg_off + comp * N + st_n0 + cl, t);
            }
            __syncthreads();
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) s0[q] = s1[q] = 0.f;
    };

    // ---- forward store pass of the tile at (m0, n0): per wave, 2 chunks of 32 pixels through a private bf16 corner; lane (pixel fr,
    // half fg) holds for block nt the channels nt*32 + 8 j + 4 fg + (0..3) in acc[mt][nt][4 j .. 4 j + 3]
    // (Measured and dropped, profiles/archive/r03al_store_pass_cost.txt + r03am_lazy_layers.txt.  In the open this pass costs ~1600 cycles per wave
    // and tile whatever K is: 45 % on top of a K = 256 tile (timing builds, hooks kept in profiles/retired/r04_conv_pipe_timing_hooks.txt: 7.32 ms of forward launches per step, 6.53
    // without the global stores, 7.08 without the sums, 5.74 without the pass).  Converting the accumulators at the end of a tile and running
    // the rest - a 2-KB corner per wave outside the ring, 16 units: write / read / store + sums twice per 16-pixel chunk - one unit after
    // the MFMAs of each k-slice of the NEXT tile came out SLOWER in every layer (l3.c3 106 -> 123..132 us, l1.c3 246 -> 355, step 44.6 ->
    // 45.4 ms): two waves' units are ~400 VALU cycles per k-slice against 256 cycles of matrix pipe, and they share the issue port.)
    auto epilogue_fwd = [&](const int m0, const int n0, unsigned char* stg_all) {
        const int nbase = n0 + wn * (NT * 32);   // first channel of this wave
        constexpr bool infer = EK == kEkInfer;   // lp_conv_fwd_act: its own instantiations, so the training kernels' store pass carries none of it
        constexpr bool gelub = EK == kEkGeluBwd; // lp_gemm_nt_gelu_bwd: `addend` is u, the pre-activation the product's gradient passes through
        // The wave's corner: 32 pixel rows of NT*64 B, UNPADDED, the 8-B slot s of row r at position s ^ key(r) (key = r & 15 for 128-B rows,
        // (r >> 1) & 7 for 64-B rows): the 16 lanes of a ds_write_b64 group - 16 consecutive pixels, one slot - then hit 16 different slots,
        // and the ds_read_b128 service groups of the read-back (4 rows x 4 chunks each) 16 different 16-bank ranges.  (Until round 5 the rows
        // were padded to NT*64 + 16 B instead: conflict-free writes, but 2-way conflicts in the read-back - SQ_LDS_BANK_CONFLICT was 32 % of this
        // kernel's busy cycles, 54 % on the K <= 128 launches where the store pass is most of the tile, profiles/r05_pmc_mfma_spec1.json.)
        constexpr bool kPad = LP_FWD_CORNER_PAD != 0;   // (A/B builds: 1 = rounds 3 - 4's padded rows)
        constexpr int ROWB = NT * 64 + (kPad ? 16 : 0);
        unsigned char* stg = stg_all + wave * (32 * ROWB);
        const int wkey = kPad ? 0 : NT == 2 ? (fr & 15) : ((fr >> 1) & 7);
        u16x8 radd[2][32 / RP] = {};   // inference: this lane's pieces of the residual, all 8 requested before the conversion / staging work
        if ((infer || gelub) && ep.addend != nullptr) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int ps = 0; ps < 32 / RP; ++ps) {
                    const int m = m0 + wm * 64 + mt * 32 + ps * RP + prow;
                    if (m < M)
                        radd[mt][ps] = load8(ep.addend + (unsigned)out_row(m, lat, div_img, div_row, full_h, full_w) * (unsigned)ep.ldo +
                                             (unsigned)(nbase + pc * 8));
                }
        }
        if (ep.bias != nullptr) {   // (workgroup-uniform; Linear layers through lp_gemm_nt, inference with the BatchNorm folded)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(ep.bias + nbase + nt * 32 + 8 * j + 4 * fg);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[mt][nt][4 * j + e] += bv[e];
                }
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
                    const u32x2_t p = {pack_bf16x2(acc[mt][nt][4 * j], acc[mt][nt][4 * j + 1]),
                                       pack_bf16x2(acc[mt][nt][4 * j + 2], acc[mt][nt][4 * j + 3])};
                    *reinterpret_cast<u32x2_t*>(stg + fr * ROWB + ((((nt * 4 + j) * 2 + fg) ^ wkey) << 3)) = p;
                }
            __builtin_amdgcn_wave_barrier();   // (lanes exchange through the wave's private corner: lock step on the device)
#pragma unroll
            for (int ps = 0; ps < 32 / RP; ++ps) {
                const int row = ps * RP + prow;
                const int rkey = kPad ? 0 : NT == 2 ? (row & 15) : ((row >> 1) & 7);
                u16x8 w;
                {   // chunk pc = slots 2 pc, 2 pc + 1 -> positions (2 pc) ^ key and its neighbour: one aligned 16-B chunk, halves swapped for an odd key
                    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
                    const u32x4_t q = *reinterpret_cast<const u32x4_t*>(stg + row * ROWB + ((pc ^ (rkey >> 1)) << 4));
                    const bool sw = (rkey & 1) != 0;
                    const u32x4_t o = {sw ? q[2] : q[0], sw ? q[3] : q[1], sw ? q[0] : q[2], sw ? q[1] : q[3]};
                    w = __builtin_bit_cast(u16x8, o);
                }
                const int m = m0 + wm * 64 + mt * 32 + row;
                if (m < M) {
                    const unsigned off = (unsigned)out_row(m, lat, div_img, div_row, full_h, full_w) * (unsigned)ep.ldo + (unsigned)(nbase + pc * 8);
                    if (infer) {   // lp_conv_fwd_act: + residual, ReLU (on the bf16 value of accumulator + bias: one more rounding than kModeInfer)
                        float v[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = bf16_to_f32(w[q]);
                        if (ep.addend != nullptr) {
#pragma unroll
                            for (int q = 0; q < 8; ++q) v[q] += bf16_to_f32(radd[mt][ps][q]);
                        }
                        if (ep.relu_fwd) {
#pragma unroll
                            for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], 0.f);
                        }
                        w = pack_bf16x8(v);
                    }
                    if (gelub) {   // d u = bf16(d a) * GELU'(u): the arithmetic of the stand-alone pass (vit.hip: gelu_bwd_kernel) on the value it would have read
                        float v[8], u8[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = bf16_to_f32(w[q]), u8[q] = bf16_to_f32(radd[mt][ps][q]);
                        gelu8_bwd(v, u8);
                        w = pack_bf16x8(v);
                    }
                    store8(ep.out_bf16 + off, w);
                    if (EK == kEkGeluFwd) {   // the activation beside it: the arithmetic of vit.hip's gelu_fwd_kernel on the value just stored
                        float v[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = bf16_to_f32(w[q]);
                        gelu8(v);
                        store8(ep.out2_bf16 + off, pack_bf16x8(v));
                    }
                    if (want_stats) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const float vr = bf16_to_f32(w[q]);
                            s0[q] += vr;
                            s1[q] = fmaf(vr, vr, s1[q]);
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    };

    // ---- data-gradient store pass.  The tensors it reads back (addend = the gradient arriving over the residual branch, the
    // pre-normalisation tensor z of the BatchNorm whose backward sums are fused here, the ReLU mask in one of its three forms) are 2 - 3x
    // the bytes of the result, so their loads are issued a whole 32-pixel chunk at a time and EARLY: chunk 0's before the MFMAs of the
    // tile's last K step, chunk 1's before chunk 0 is processed - the store pass then waits for memory once, behind work, instead of
    // twice in the open.  fp32 through LDS, 16 pixels x the wave's NT*32 channels at a time (full 128-B lines per row for BN = 128);
    // everything is applied before the single rounding to bf16, as in conv_igemm_kernel.
    constexpr int PH = 16 / RP;          // read-back passes per 16-pixel half
    constexpr int NPC = 2 * PH;          // pieces (8 channels of one row) per lane and 32-pixel chunk
    constexpr bool kLa = (EK == kEkAZB || EK == kEkPlain || EK == kEkPB), kLz = (EK != kEkNone && EK != kEkInfer && EK != kEkPB && EK != kEkGeluBwd && EK != kEkGeluFwd),
                   kLb = (EK == kEkAZB || EK == kEkPB);
    // (kEkAZB - the hottest data gradient, at the register cap - does not keep the output offsets of its pieces: its launches cover the full
    //  pixel lattice (host-checked), so an offset is two multiply-adds away and is recomputed in rb_process: 8 VGPRs, the 7 it used to spill)
    constexpr bool kKeepOff = EK != kEkAZB;
    struct ReadBack {
        unsigned off[kKeepOff ? NPC : 1];
        u16x8 la[kLa ? NPC : 1];   // addend
        u16x8 lz[kLz ? NPC : 1];   // z (kEkZ, kEkAZB) or the activation used as ReLU mask (kEkPlain)
        unsigned lb[kLb ? NPC : 1];
    };
    float mu[8], sc[8], be[8];           // this thread's channels of the fused BatchNorm (reloaded when the column block / segment changes)
    auto rb_issue = [&](ReadBack& rb, const int mt, const int m0, const int n0) {
        const int nbase = n0 + wn * (NT * 32);
#pragma unroll
        for (int p = 0; p < NPC; ++p) {
            const int m = m0 + wm * 64 + mt * 32 + (p / PH) * 16 + (p % PH) * RP + prow;
            const bool rv = m < M;
            const unsigned off_p = rv ? (unsigned)(kKeepOff ? out_row(m, lat, div_img, div_row, full_h, full_w) : m) * (unsigned)ep.ldo + (unsigned)(nbase + pc * 8) : ~0u;
            if (kKeepOff) rb.off[kKeepOff ? p : 0] = off_p;
            const unsigned o = rv ? off_p : (unsigned)(nbase + pc * 8);   // (rows past M: any valid address, the result is not stored)
            if (EK == kEkAZB && ep.addend_half) {
                // the shortcut's gradient lives on the half-resolution grid: pixel (b, y, x) of this (full) lattice takes row (b, y / 2, x / 2) of
                // it if y and x are both even, nothing otherwise
                const int mm = rv ? m : 0;
                const int bi = fdiv(mm, div_img), rem = mm - bi * div_img.d;
                const int yy = fdiv(rem, div_row), xx = rem - yy * div_row.d;
                const bool on = rv && !((yy | xx) & 1);
                const unsigned oa = (unsigned)((bi * ((full_h + 1) >> 1) + (yy >> 1)) * ((full_w + 1) >> 1) + (xx >> 1)) * (unsigned)ep.ldo + (unsigned)(nbase + pc * 8);
                rb.la[kLa ? p : 0] = on ? load8_stream(ep.addend + oa) : zero8();
            } else if (EK == kEkAZB || ((EK == kEkPlain || EK == kEkPB) && ep.addend)) rb.la[kLa ? p : 0] = load8_stream(ep.addend + o);
            if (bwd) rb.lz[kLz ? p : 0] = load8(ep.bn_z + o);
            if (EK == kEkPlain && ep.relu_mask) rb.lz[kLz ? p : 0] = load8_stream(ep.relu_mask + o);
            if (EK == kEkAZB || EK == kEkPB) rb.lb[kLb ? p : 0] = ep.relu_bits[o >> 3];
        }
    };
    auto rb_process = [&](const ReadBack& rb, const int mt, unsigned char* stg_all, const int m0, const int n0) {
        // fp32 row of the wave's channels + pad.  (The XOR-swizzled unpadded layout of the forward's corner was tried here too, round 5: the
        // padded rows' read-back groups have 2-way conflicts - SQ_LDS_BANK_CONFLICT 9 - 16 % of these kernels' busy cycles - but the data gradients
        // sit at the register cap, and the swizzle's address arithmetic made three of them spill: tests/test_kernel_resources.py.)
        constexpr int ROWF = NT * 128 + 16;
        unsigned char* stg = stg_all + wave * (16 * ROWF);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if ((fr >> 4) == h) {   // the 16 pixels of this half write their 4-channel runs
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 p4 = {acc[mt][nt][4 * j], acc[mt][nt][4 * j + 1], acc[mt][nt][4 * j + 2], acc[mt][nt][4 * j + 3]};
                        *reinterpret_cast<f32x4*>(stg + (fr & 15) * ROWF + (nt * 32 + 8 * j + 4 * fg) * 4) = p4;
                    }
            }
            __builtin_amdgcn_wave_barrier();
            f32x4 lo_[PH], hi_[PH];
#pragma unroll
            for (int ps = 0; ps < PH; ++ps) {
                lo_[ps] = *reinterpret_cast<const f32x4*>(stg + (ps * RP + prow) * ROWF + pc * 32);
                hi_[ps] = *reinterpret_cast<const f32x4*>(stg + (ps * RP + prow) * ROWF + pc * 32 + 16);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ps = 0; ps < PH; ++ps) {
                const int p = h * PH + ps;
                unsigned off_p;
                if (kKeepOff) {
                    off_p = rb.off[kKeepOff ? p : 0];
                } else {
                    const int m = m0 + wm * 64 + mt * 32 + h * 16 + ps * RP + prow;
                    off_p = m < M ? (unsigned)m * (unsigned)ep.ldo + (unsigned)(n0 + wn * (NT * 32) + pc * 8) : ~0u;
                }
                if (off_p != ~0u) {
                    float v[8] = {lo_[ps][0], lo_[ps][1], lo_[ps][2], lo_[ps][3], hi_[ps][0], hi_[ps][1], hi_[ps][2], hi_[ps][3]};
                    if (EK == kEkAZB || ((EK == kEkPlain || EK == kEkPB) && ep.addend)) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] += bf16_to_f32(rb.la[kLa ? p : 0][q]);
                    }
                    float zc[8];
                    if (bwd) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) zc[q] = bf16_to_f32(rb.lz[kLz ? p : 0][q]) - mu[q];
                        if (EK == kEkZ) {
#pragma unroll
                            for (int q = 0; q < 8; ++q)
                                if (!(fmaf(zc[q], sc[q], be[q]) > 0x1p-134f)) v[q] = 0.f;
                        }
                    }
                    if (EK == kEkPlain && ep.relu_mask) {
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                            if (!bf16_positive(rb.lz[kLz ? p : 0][q])) v[q] = 0.f;
                    }
                    if (EK == kEkAZB || EK == kEkPB) {
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                            if (!((rb.lb[kLb ? p : 0] >> q) & 1u)) v[q] = 0.f;
                    }
                    const u16x8 w = pack_bf16x8(v);
                    store8(ep.out_bf16 + off_p, w);
                    if (want_stats) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const float vr = bf16_to_f32(w[q]);
                            s0[q] += vr;
                            s1[q] = fmaf(vr, bwd ? zc[q] : vr, s1[q]);   // backward: x invstd at the flush
                        }
                    }
                }
            }
        }
    };

    // ---- the walk: loader two K steps ahead of the MFMAs, across tile boundaries; the ring never drains
    if (HALO) {   // the first halo image, whole; the loader then stays one image ahead
        halo_setup(hl_vt);
#pragma unroll
        for (int i = 0; i < (HALO ? NA : 0); ++i) halo_issue(i);
        halo_advance();
    }
    setup(ld_vt);
#pragma unroll
    for (int st = 0; st < NST - 1; ++st) load_step(st);
    int cur = 0;   // stage of the K step the MFMAs are about to consume
    ReadBack rb0, rb1;
    // LDS nobody else touches between a tile's last K step and the next tile's first barrier: the stage (HALO: the halo image) consumed last
    auto free_lds = [&]() -> unsigned char* { return HALO ? halo0 + (mh_buf ^ 1) * kHaloB : smem + (cur == 0 ? NST - 1 : cur - 1) * kStage; };
    for (int vt = blockIdx.x; vt < ntiles; vt += gridDim.x) {
        const int tile = xcd_remap(vt, ntiles);
        const int tm_ = tile / tiles_n;
        const int m0 = tm_ * kPM, n0 = (tile - tm_ * tiles_n) * BN;
        const int seg_off = (ep.seg_images > 0 && m0 >= ep.seg_images * rows_y * rows_x) ? N : 0;
        if (HALO) {
            const int pbase = padded(m0) - (hW + 3), rr0 = padded_row(m0) - 1;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int m = m0 + wm * 64 + mt * 32 + fr, mc = m < M ? m : M - 1;   // (rows past M: any staged row, the result is not stored)
                ploc[mt] = padded(mc) - pbase;
                uloc[mt] = ploc[mt] - 2 * (padded_row(mc) - rr0);
            }
        }
        if (n0 != st_n0 || seg_off != st_seg_off) {   // (workgroup-uniform) new column block / BatchNorm segment
            // the stage the previous tile consumed last is free until this tile's first K step has passed its barrier
            if (want_stats) stats_flush(reinterpret_cast<float*>(free_lds()));
            st_n0 = n0;
            st_seg_off = seg_off;
            if (bwd) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int n = n0 + wn * (NT * 32) + pc * 8 + q;
                    mu[q] = ep.bn_mean[seg_off + n];
                    if (EK == kEkZ) {
                        sc[q] = ep.bn_invstd[seg_off + n] * ep.bn_gamma[n];
                        be[q] = ep.bn_beta[n];
                    }
                }
            }
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;
        for (int kt = 0; kt < KT; ++kt) {
            // this wave's loads of the current step have landed; the next step's (4 + NBL; HALO: its NBL weight pieces - the halo piece
            // issued before them is the oldest of that step and is waited for a step early, which costs nothing: it has 8 steps to land) stay in flight
            // (Measured and dropped, profiles/archive/r03p_vmcnt_layers.txt: right behind a tile boundary the needed loads are OLDER than the
            // previous tile's output stores, so the count could leave those stores in flight instead of waiting for them to reach
            // memory - every layer came out 0 - 7 % SLOWER, forward and data gradient: the wait paces the workgroups' store bursts.)
            // (the count = the weight pieces of the NST - 2 younger steps; the halo pieces those steps may also have issued are left out: conservative)
            if (HALO && NBL == 2 && NST == 4) LP_WAIT_VM(4);
            else if (HALO && NST == 4) LP_WAIT_VM(2);
            else if (HALO && NBL == 2) LP_WAIT_VM(2);
            else if (HALO) LP_WAIT_VM(1);
            else if (NBL == 2) LP_WAIT_VM(6);
            else LP_WAIT_VM(5);
            LP_RAW_BARRIER();              // ... everyone's have, and everyone is done reading the stage refilled next
            if (!kFwd && kt == KT - 1) rb_issue(rb0, 0, m0, n0);   // the first chunk's read-backs travel under the last K step
            const int nxt = cur == 0 ? NST - 1 : cur - 1;   // (cur + NST - 1) % NST: the stage consumed at the previous step
            if (kSpread || HALO) {
                prep_step(nxt);
                mma_stage(cur, true);
                advance_tile();
            } else {
                load_step(nxt);
                mma_stage(cur, false);
            }
            cur = cur == NST - 1 ? 0 : cur + 1;
            if (HALO && ++mh_tap == 9) {   // the slice is consumed: the MFMAs move to the other halo image, the loader to the one after it
                mh_tap = 0;
                mh_buf ^= 1;
                halo_advance();
            }
        }
        unsigned char* stg_all = free_lds();   // the stage consumed last: free until the next K step's barrier
        LP_RAW_BARRIER();                  // every wave is done reading its fragments
        if (kFwd) {
            epilogue_fwd(m0, n0, stg_all);
        } else {
            rb_issue(rb1, 1, m0, n0);
            rb_process(rb0, 0, stg_all, m0, n0);
            rb_process(rb1, 1, stg_all, m0, n0);
        }
    }
    if (want_stats) stats_flush(reinterpret_cast<float*>(free_lds()));
    LP_WAIT_VM(0);   // the ring's last (empty) loads still target this workgroup's LDS
}

// ------------------------------------------------------------------------------------------------------------
// Weight gradient on the same ring:  D[a][b] = sum over pixels m of  Pa[m][a] * Qb[m][b]
//   ordinary   a = (r, s, ci) of the gathered activations x (filter taps in the address), b = co of dy     -> dW[b][a]
//   swapped    (1x1, few input channels) a = co of dy, b = ci of x                                            -> dW[a][b]
// Both operands keep their memory orientation in LDS ([pixel][channel]; the contraction index is the ROW): a K step is 64 pixels,
// the A image 64 rows x 256 channels (512-B rows), the B image 64 rows x BN channels, filled by direct-to-LDS loads; the MFMA
// fragments (8 consecutive pixels of one channel per lane) come out of ds_read_b64_tr_b16 as in conv_wgrad_kernel.  Without row
// padding (the LDS side of a direct load is lane-linear) the four pixel rows of a transpose read would sit on the same banks, so the
// 16-B chunk c of pixel row r is stored at c ^ ((r & 3) << 2) (BN = 64: c ^ (((r >> 1) & 1) << 2), two 128-B rows share a bank
// row) - applied on the source address, and arranged so that a THREAD always fetches the same chunk (its filter tap is a constant).
// One (tile, pixel slice) per workgroup, 512 threads = 8 waves (4 along a x 2 along b, 64 x BN/2 each), tile 256 x BN: 48 KB per
// K step for 1024 MFMA cycles (conv_wgrad_kernel: 32 KB per 512 - at the CU's 64 B/clk address path that alone caps it).  The
// slices' partial tiles go to the workspace in accumulator order; wgrad_pipe_reduce_kernel adds them up in a fixed order.
// ------------------------------------------------------------------------------------------------------------
struct WgradPipeGeom {
    int B, Hi, Wi, Ho, Wo;     // source image of the gathered operand, pixel grid of the contraction (rows m = (b, ho, wo))
    int Ca, R, S, stride, pad; // channels per filter tap of the gathered operand, the taps
    int Ka, Cb;                // a-extent (R * S * Ca) and b-extent (= row pitch of Qb)
    int plain;                 // 1x1 / stride 1 / no padding: row m of Pa is pixel m
};

template <int BN>
__global__ __launch_bounds__(512) void conv_wgrad_pipe_kernel(const unsigned short* __restrict__ Pa, const unsigned short* __restrict__ Qb,
                                                              unsigned pa_bytes, unsigned qb_bytes, WgradPipeGeom g, int M, int tiles,
                                                              int tiles_b, int m_per_split, FastDiv div_hw, FastDiv div_wo,
                                                              float* __restrict__ ws) {
    constexpr int NT = BN / 64;
    constexpr int kRowA = 512, kRowB = BN * 2;            // bytes per pixel row of the two LDS images
    constexpr int kStageA = kBK * kRowA, kStageB = kBK * kRowB, kStage = kStageA + kStageB;
    constexpr int NLB = BN / 64;                           // B loads per thread and K step (4 A loads)
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * kStage];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const int work = xcd_remap(blockIdx.x, gridDim.x);    // slice slow: an XCD owns whole slices (all tiles of a slice share its L2)
    const int slice = work / tiles, tile = work - slice * tiles;
    const int a0 = (tile / tiles_b) * 256, b0 = (tile % tiles_b) * BN;
    const int m_begin = slice * m_per_split, m_end = min(M, m_begin + m_per_split);
    const int KT = (m_end - m_begin + kBK - 1) / kBK;
    const buf_rsrc rsrc_a = make_buf_rsrc(Pa, pa_bytes), rsrc_b = make_buf_rsrc(Qb, qb_bytes);

    // ---- loader.  A: wave instruction p covers pixel rows p*16 + wave*2 + (lane >> 5) (two 512-B rows); B (BN = 128): rows
    // q*32 + wave*4 + (lane >> 4) (four 256-B rows); B (BN = 64): rows wave*8 + (lane >> 3) (eight 128-B rows)
    const int arow = wave * 2 + (lane >> 5);                               // + 16 p
    const int achunk = (lane & 31) ^ ((arow & 3) << 2);                    // the chunk this thread always fetches
    const int ja = a0 + achunk * 8;
    const bool ja_ok = ja < g.Ka;
    const int tap = (ja_ok ? ja : 0) / g.Ca;
    const int tcn = (ja_ok ? ja : 0) - tap * g.Ca;
    const int tr = tap / g.S, ts = tap - tr * g.S;
    const int brow = (BN == 128) ? wave * 4 + (lane >> 4) : wave * 8 + (lane >> 3);   // + 32 q
    const int bchunk = (BN == 128) ? ((lane & 15) ^ ((brow & 3) << 2)) : ((lane & 7) ^ (((brow >> 1) & 1) << 2));
    const unsigned boff = (unsigned)(b0 + bchunk * 8) * 2u;
    const int hw = g.Ho * g.Wo;

    unsigned is_a[4], is_b[NLB];
    unsigned char* is_dst = smem;
    // Stride-1 "same" convolutions (every 3x3 of the trunk but the three stride-2 ones): the source pixel of output pixel m under tap
    // (tr, ts) is m + const, so a row's byte offset is linear in m and only the padding decision needs (ho, wo) - tracked per row and
    // advanced by 64 pixels per K step (64 = qb images + qh rows + qw pixels) instead of two divisions per row and step.
    const bool same = !g.plain && g.stride == 1 && g.Hi == g.Ho && g.Wi == g.Wo && hw >= kBK;
    const int qb64 = kBK / hw, qh64 = (kBK - qb64 * hw) / g.Wo, qw64 = kBK - qb64 * hw - qh64 * g.Wo;
    const int lo_h = max(0, g.pad - tr), span_h = min(g.Ho, g.Hi + g.pad - tr) - lo_h;   // valid output rows / columns for this thread's tap
    const int lo_w = max(0, g.pad - ts), span_w = min(g.Wo, g.Wi + g.pad - ts) - lo_w;
    const unsigned tap_off = (unsigned)(((tr - g.pad) * g.Wi + (ts - g.pad)) * g.Ca + tcn) * 2u;   // (may wrap: added mod 2^32)
    int rho[4], rwo[4];
    if (same) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int m = m_begin + p * 16 + arow;
            const int mm = m < M ? m : 0;
            const int rem = mm - fdiv(mm, div_hw) * hw;
            rho[p] = fdiv(rem, div_wo);
            rwo[p] = rem - rho[p] * g.Wo;
        }
    }
    auto prep_step = [&](int st, int kt) {
        const int mk = m_begin + kt * kBK;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int m = mk + p * 16 + arow;
            unsigned off;
            bool ok = ja_ok && m < m_end && kt < KT;
            if (g.plain) {
                off = (unsigned)(m * g.Ca + tcn) * 2u;
            } else if (same) {
                ok = ok && (unsigned)(rho[p] - lo_h) < (unsigned)span_h && (unsigned)(rwo[p] - lo_w) < (unsigned)span_w;
                off = (unsigned)m * (unsigned)(g.Ca * 2) + tap_off;
                rwo[p] += qw64;                       // this row's pixel of the NEXT step (prep_step is called with consecutive kt)
                const int c = rwo[p] >= g.Wo ? 1 : 0;
                rwo[p] -= c ? g.Wo : 0;
                rho[p] += qh64 + c;
                rho[p] -= rho[p] >= g.Ho ? g.Ho : 0;
            } else {
                const int mm = ok ? m : 0;
                const int b = fdiv(mm, div_hw), rem = mm - b * hw;
                const int ho = fdiv(rem, div_wo), wo = rem - ho * g.Wo;
                const int hi = ho * g.stride - g.pad + tr, wi = wo * g.stride - g.pad + ts;
                ok = ok && hi >= 0 && hi < g.Hi && wi >= 0 && wi < g.Wi;
                off = (unsigned)(((b * g.Hi + hi) * g.Wi + wi) * g.Ca + tcn) * 2u;
            }
            is_a[p] = ok ? off : ~0u;
        }
#pragma unroll
        for (int q = 0; q < NLB; ++q) {
            const int m = mk + q * 32 + brow;
            is_b[q] = (m < m_end && kt < KT) ? (unsigned)m * (unsigned)(g.Cb * 2) + boff : ~0u;
        }
        is_dst = smem + st * kStage;
    };
    auto issue_load = [&](int i) {   // (compile-time i)
        if (i < 4) buf_load16_lds(rsrc_a, is_dst + (i * 16 + wave * 2) * kRowA, is_a[i], 0u);
        else if (BN == 128) buf_load16_lds(rsrc_b, is_dst + kStageA + ((i - 4) * 32 + wave * 4) * kRowB, is_b[i - 4], 0u);
        else buf_load16_lds(rsrc_b, is_dst + kStageA + (wave * 8) * kRowB, is_b[0], 0u);
    };

    // ---- fragments: lane of 16-lane group fq supplies pixel row 8 (fq / 2) + (fi / 4) (+ 4 for the second read), channels
    // 16 (fq % 2) + 4 (fi % 4) .. + 3 of a 32-channel block, and receives channel lane % 32, pixels 8 (lane / 32) .. + 7
    const int fq = lane >> 4, fi = lane & 15;
    const int frow = (fq >> 1) * 8 + (fi >> 2), fcol = (fq & 1) * 16 + (fi & 3) * 4;
    const int swa = (frow & 3) << 2;                                                       // (the same for row + 4, + 8, + 16 kk)
    const int swb = (BN == 128) ? ((frow & 3) << 2) : (((frow >> 1) & 1) << 2);
    unsigned foa[2], fob[NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int e0 = wm * 64 + mt * 32 + fcol;
        foa[mt] = (unsigned)(frow * kRowA + (((e0 >> 3) ^ swa) << 4) + (e0 & 7) * 2);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int e0 = wn * (NT * 32) + nt * 32 + fcol;
        fob[nt] = (unsigned)(kStageA + frow * kRowB + (((e0 >> 3) ^ swb) << 4) + (e0 & 7) * 2);
    }
    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;

    auto mma_stage = [&](int st) {
        const unsigned char* sb = smem + st * kStage;
        bf16x8 a[2][2], b[2][NT];
        // per stage one base address per fragment column block; the k-slice and the second pixel quad ride in the instruction's offset
        const unsigned char* pa[2];
        const unsigned char* pb[NT];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) pa[mt] = sb + foa[mt];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) pb[nt] = sb + fob[nt];
        auto fetch = [&](auto kk_c, int set) {
            constexpr int kk = decltype(kk_c)::value;
            typedef __attribute__((ext_vector_type(8))) short s16x8_t;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const s16x4_t lo = lds_read_tr16_async_off<kk * 16 * kRowA>(pa[mt]), hi = lds_read_tr16_async_off<kk * 16 * kRowA + 4 * kRowA>(pa[mt]);
                const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                a[set][mt] = __builtin_bit_cast(bf16x8, v);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const s16x4_t lo = lds_read_tr16_async_off<kk * 16 * kRowB>(pb[nt]), hi = lds_read_tr16_async_off<kk * 16 * kRowB + 4 * kRowB>(pb[nt]);
                const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                b[set][nt] = __builtin_bit_cast(bf16x8, v);
            }
        };
        // the transpose reads are asm (see lds_read_tr16_async): slice kk's 2 (2 + NT) reads must have returned before its MFMAs, slice
        // kk + 1's (just issued) may still be in flight - LDS operations return in order
        auto ready = [&](int set, bool last) {
            if (NT == 2) {
                if (last) LP_WAIT_LGKM_TOUCH4(0, a[set][0], a[set][1], b[set][0], b[set][NT - 1]);
                else LP_WAIT_LGKM_TOUCH4(8, a[set][0], a[set][1], b[set][0], b[set][NT - 1]);
            } else {
                if (last) LP_WAIT_LGKM_TOUCH3(0, a[set][0], a[set][1], b[set][0]);
                else LP_WAIT_LGKM_TOUCH3(6, a[set][0], a[set][1], b[set][0]);
            }
        };
        auto slice = [&](auto kk_c) {
            constexpr int kk = decltype(kk_c)::value;
            if constexpr (kk + 1 < kBK / 16) fetch(std::integral_constant<int, kk + 1>{}, (kk + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            ready(kk & 1, kk + 1 == kBK / 16);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kk & 1][mt], b[kk & 1][nt], acc[mt][nt], 0, 0, 0);
            constexpr int NL = 4 + NLB;   // the next-but-one step's loads, spread between the k-slices
            if (kk == 0) {
                issue_load(0);
                issue_load(1);
            } else if (kk == 1) {
                issue_load(2);
                if (NL == 6) issue_load(3);
            } else if (kk == 2) {
                issue_load(NL == 6 ? 4 : 3);
            } else {
                issue_load(NL - 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        fetch(std::integral_constant<int, 0>{}, 0);
        slice(std::integral_constant<int, 0>{});
        slice(std::integral_constant<int, 1>{});
        slice(std::integral_constant<int, 2>{});
        slice(std::integral_constant<int, 3>{});
    };

    prep_step(0, 0);
#pragma unroll
    for (int i = 0; i < 4 + NLB; ++i) issue_load(i);
    prep_step(1, 1);
#pragma unroll
    for (int i = 0; i < 4 + NLB; ++i) issue_load(i);
    int cur = 0;
    for (int kt = 0; kt < KT; ++kt) {
        if (NLB == 2) LP_WAIT_VM(6);
        else LP_WAIT_VM(5);
        LP_RAW_BARRIER();
        prep_step(cur == 0 ? 2 : cur - 1, kt + 2);   // (steps past the slice fetch nothing: the ring keeps its count)
        mma_stage(cur);
        cur = cur == 2 ? 0 : cur + 1;
    }
    LP_WAIT_VM(0);
    // partial tile -> workspace[slice][tile][wave][mt][nt][e][lane] (accumulator order: every store is a full 256-B line per wave)
    float* dst = ws + ((((size_t)slice * tiles + tile) * 8 + wave) * (2 * NT * 16)) * 64 + lane;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) dst[((mt * NT + nt) * 16 + e) * 64] = acc[mt][nt][e];
}

// dW[a * sa + b * sb] += sum over slices of the partial tiles, slices in order (deterministic).  One workgroup per 64 consecutive
// accumulator elements of a tile; its 4 waves stride over the slices and combine through LDS.
template <int BN>
__global__ __launch_bounds__(256) void wgrad_pipe_reduce_kernel(const float* __restrict__ ws, int slices, int tiles, int tiles_b, int Ka, int Cb,
                                                                int sa, int sb, float* __restrict__ dW) {
    constexpr int NT = BN / 64;
    constexpr int PER_TILE = 8 * 2 * NT * 16 * 64;
    __shared__ float part[4][64];
    const size_t total = (size_t)tiles * PER_TILE;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t i = (size_t)blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f;
    int sl = w;
    for (; sl + 4 < slices; sl += 8) {
        s0 += ws[(size_t)sl * total + i];
        s1 += ws[(size_t)(sl + 4) * total + i];
    }
    for (; sl < slices; sl += 4) s0 += ws[(size_t)sl * total + i];
    part[w][lane] = s0 + s1;
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
        const int wave = (int)(t & 7);
        const int tile = (int)(t >> 3);
        const int wm = wave & 3, wn = wave >> 2;
        const int a = (tile / tiles_b) * 256 + wm * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        const int b = (tile % tiles_b) * BN + wn * (NT * 32) + nt * 32 + (lane & 31);
        if (a < Ka && b < Cb) dW[(size_t)a * sa + (size_t)b * sb] += sum;
    }
}

}  // namespace lp
