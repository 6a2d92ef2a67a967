// Pipelined implicit-GEMM convolution (round 3): forward and data-gradient of the ResNet-50 trunk on a direct-to-LDS operand ring.
//
// Why a second kernel next to conv_igemm_kernel (conv.hip): that kernel fetches ONE K step ahead into registers, so every
// K step of every tile waits a full (loaded) memory latency - the 1x1 layers ran at ~3 TB/s of a 6.3 TB/s part and their
// data gradients spent 53 % of their wave cycles parked on memory (profiles/r02_pmc_mfma.json).  Here the operands go
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

namespace lp {

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
#else
#define LP_WAIT_VM(n) ((void)0)
#define LP_RAW_BARRIER() __syncthreads()
#endif

// 16-B store; `stream`: non-temporal (the lines are not kept in L2, which the weight panel of the deep 1x1 layers needs)
__device__ __forceinline__ void store8(unsigned short* p, u16x8 v, bool stream) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (stream) __builtin_nontemporal_store(v, reinterpret_cast<u16x8*>(p));
    else *reinterpret_cast<u16x8*>(p) = v;
#else
    (void)stream;
    *reinterpret_cast<u16x8*>(p) = v;
#endif
}

// EK = what the data gradient's store pass reads back (its own instantiation each, so the read-back registers of one form are not
// allocated in the others; anything else goes to conv_igemm_kernel):
//   kEkNone  forward
//   kEkZ     BatchNorm backward sums + ReLU mask recomputed from the pre-normalisation tensor z (conv3 / conv2 of every block)
//   kEkAZB   addend + z (sums) + 1-bit ReLU mask (conv1 of the blocks behind an identity shortcut)
//   kEkPlain no BatchNorm fused: optional addend, optional bf16 activation as the ReLU mask (conv1 / projection shortcut of the first
//            block of a layer, whose input gradient has two writers)
enum { kEkNone = 0, kEkZ = 1, kEkAZB = 2, kEkPlain = 3 };

template <int BN, int MODE, int EK>
__global__ __launch_bounds__(512) void conv_pipe_kernel(const unsigned short* __restrict__ X, const unsigned short* __restrict__ Wt,
                                                        unsigned x_bytes, unsigned w_bytes, ConvGeom g, Lattice lat, FastDiv div_img,
                                                        FastDiv div_row, int M, int N, int K, int tiles_n, int ntiles, ConvEpilogue ep, int flags) {
    static_assert((MODE == kModeFwd && EK == kEkNone) || (MODE == kModeDgrad && EK != kEkNone), "trunk convolutions only");
    constexpr int NT = BN / 64;                  // 32-channel MFMA blocks per wave along N (wave tile 64 pixels x NT*32 channels)
    constexpr int NBL = BN / 64;                 // weight rows each thread stages per K step
    constexpr int kStageA = kPM * kPRowB, kStageB = BN * kPRowB, kStage = kStageA + kStageB;
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * kStage];

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
        for (int i = 0; i < 4; ++i) {
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
        if (i < 4) buf_load16_lds(rsrc_x, is_dst + i * (64 * kPRowB), voff[i], is_soff_a);
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
        for (int i = 0; i < 4 + NBL; ++i) issue_load(i);
        advance_tile();
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
    auto mma_stage = [&](int st, const bool spread) {
        const unsigned char* sb = smem + st * kStage;
        bf16x8 a[2][2], b[2][NT];
        auto fetch = [&](int kk, int set) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                a[set][mt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u16x8*>(sb + a_row + mt * (32 * kPRowB) + koff[kk]));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                b[set][nt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u16x8*>(sb + b_row + nt * (32 * kPRowB) + koff[kk]));
        };
        fetch(0, 0);
#pragma unroll
        for (int kk = 0; kk < kBK / 16; ++kk) {
            if (kk + 1 < kBK / 16) fetch(kk + 1, (kk + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)   // roles swapped: D[channel][pixel]
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[kk & 1][nt], a[kk & 1][mt], acc[mt][nt], 0, 0, 0);
            if (spread) {
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
                atomicAdd(&ep.stats_sums[2 * st_seg_off + comp * N + st_n0 + cl], t);
                float* accp = comp == 0 ? ep.stats_acc0 : ep.stats_acc1;
                if (accp != nullptr) atomicAdd(&accp[st_n0 + cl], t);
            }
            __syncthreads();
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) s0[q] = s1[q] = 0.f;
    };

    // ---- forward store pass of the tile at (m0, n0): per wave, 2 chunks of 32 pixels through a private bf16 corner; lane (pixel fr,
    // half fg) holds for block nt the channels nt*32 + 8 j + 4 fg + (0..3) in acc[mt][nt][4 j .. 4 j + 3]
    auto epilogue_fwd = [&](const int m0, const int n0, unsigned char* stg_all) {
        const int nbase = n0 + wn * (NT * 32);   // first channel of this wave
        constexpr int ROWB = NT * 64 + 16;       // bf16 chunk row + pad (16-B aligned)
        unsigned char* stg = stg_all + wave * (32 * ROWB);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
                    const u32x2_t p = {pack_bf16x2(acc[mt][nt][4 * j], acc[mt][nt][4 * j + 1]),
                                       pack_bf16x2(acc[mt][nt][4 * j + 2], acc[mt][nt][4 * j + 3])};
                    *reinterpret_cast<u32x2_t*>(stg + fr * ROWB + (nt * 32 + 8 * j + 4 * fg) * 2) = p;
                }
            __builtin_amdgcn_wave_barrier();   // (lanes exchange through the wave's private corner: lock step on the device)
#pragma unroll
            for (int ps = 0; ps < 32 / RP; ++ps) {
                const int row = ps * RP + prow;
                const u16x8 w = *reinterpret_cast<const u16x8*>(stg + row * ROWB + pc * 16);
                const int m = m0 + wm * 64 + mt * 32 + row;
                if (m < M) {
                    const unsigned off = (unsigned)out_row(m, lat, div_img, div_row, full_h, full_w) * (unsigned)ep.ldo + (unsigned)(nbase + pc * 8);
                    store8(ep.out_bf16 + off, w, (flags & 2) != 0);
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
    constexpr bool kLa = (EK == kEkAZB || EK == kEkPlain), kLz = (EK != kEkNone), kLb = (EK == kEkAZB);
    struct ReadBack {
        unsigned off[NPC];
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
            rb.off[p] = rv ? (unsigned)out_row(m, lat, div_img, div_row, full_h, full_w) * (unsigned)ep.ldo + (unsigned)(nbase + pc * 8) : ~0u;
            const unsigned o = rv ? rb.off[p] : (unsigned)(nbase + pc * 8);   // (rows past M: any valid address, the result is not stored)
            if (EK == kEkAZB || (EK == kEkPlain && ep.addend)) rb.la[kLa ? p : 0] = load8_stream(ep.addend + o);
            if (bwd) rb.lz[p] = load8(ep.bn_z + o);
            if (EK == kEkPlain && ep.relu_mask) rb.lz[p] = load8_stream(ep.relu_mask + o);
            if (EK == kEkAZB) rb.lb[kLb ? p : 0] = ep.relu_bits[o >> 3];
        }
    };
    auto rb_process = [&](const ReadBack& rb, const int mt, unsigned char* stg_all) {
        constexpr int ROWF = NT * 128 + 16;   // fp32 row of the wave's channels + pad
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
                if (rb.off[p] != ~0u) {
                    float v[8] = {lo_[ps][0], lo_[ps][1], lo_[ps][2], lo_[ps][3], hi_[ps][0], hi_[ps][1], hi_[ps][2], hi_[ps][3]};
                    if (EK == kEkAZB || (EK == kEkPlain && ep.addend)) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] += bf16_to_f32(rb.la[kLa ? p : 0][q]);
                    }
                    float zc[8];
                    if (bwd) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) zc[q] = bf16_to_f32(rb.lz[p][q]) - mu[q];
                        if (EK == kEkZ) {
#pragma unroll
                            for (int q = 0; q < 8; ++q)
                                if (!(fmaf(zc[q], sc[q], be[q]) > 0x1p-134f)) v[q] = 0.f;
                        }
                    }
                    if (EK == kEkPlain && ep.relu_mask) {
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                            if (!bf16_positive(rb.lz[p][q])) v[q] = 0.f;
                    }
                    if (EK == kEkAZB) {
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                            if (!((rb.lb[kLb ? p : 0] >> q) & 1u)) v[q] = 0.f;
                    }
                    const u16x8 w = pack_bf16x8(v);
                    store8(ep.out_bf16 + rb.off[p], w, (flags & 2) != 0);
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
    setup(ld_vt);
    load_step(0);
    load_step(1);
    int cur = 0;   // stage of the K step the MFMAs are about to consume
    ReadBack rb0, rb1;
    for (int vt = blockIdx.x; vt < ntiles; vt += gridDim.x) {
        const int tile = xcd_remap(vt, ntiles);
        const int tm_ = tile / tiles_n;
        const int m0 = tm_ * kPM, n0 = (tile - tm_ * tiles_n) * BN;
        const int seg_off = (ep.seg_images > 0 && m0 >= ep.seg_images * rows_y * rows_x) ? N : 0;
        if (n0 != st_n0 || seg_off != st_seg_off) {   // (workgroup-uniform) new column block / BatchNorm segment
            // the stage the previous tile consumed last is free until this tile's first K step has passed its barrier
            if (want_stats) stats_flush(reinterpret_cast<float*>(smem + (cur == 0 ? 2 : cur - 1) * kStage));
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
            if (NBL == 2) LP_WAIT_VM(6);   // this wave's loads of the current step have landed; the next step's 4 + NBL stay in flight
            else LP_WAIT_VM(5);
            LP_RAW_BARRIER();              // ... everyone's have, and everyone is done reading the stage refilled next
            if (!kFwd && kt == KT - 1) rb_issue(rb0, 0, m0, n0);   // the first chunk's read-backs travel under the last K step
            const int nxt = cur == 0 ? 2 : cur - 1;   // (cur + 2) % 3
            if (kSpread) {
                prep_step(nxt);
                mma_stage(cur, true);
                advance_tile();
            } else {
                load_step(nxt);
                mma_stage(cur, false);
            }
            cur = cur == 2 ? 0 : cur + 1;
        }
        unsigned char* stg_all = smem + (cur == 0 ? 2 : cur - 1) * kStage;   // the stage consumed last: free until the next K step's barrier
        LP_RAW_BARRIER();                  // every wave is done reading its fragments
        if (kFwd) {
            epilogue_fwd(m0, n0, stg_all);
        } else {
            rb_issue(rb1, 1, m0, n0);
            rb_process(rb0, 0, stg_all);
            rb_process(rb1, 1, stg_all);
        }
    }
    if (want_stats) stats_flush(reinterpret_cast<float*>(smem + (cur == 0 ? 2 : cur - 1) * kStage));
    LP_WAIT_VM(0);   // the ring's last (empty) loads still target this workgroup's LDS
}

}  // namespace lp
