// Fused soft-max attention forward for the ViT path (reference: HF ViTSelfAttention's eager attention behind
// lightning_pose/models/backbones/vit.py:38-43):  P = softmax(scale * Q K^T),  O = P V  per (image, head), head dimension 64.
//
// The composition it replaces (lp_gemm_nt -> lp_softmax_rows_fwd -> lp_gemm_nt) writes the scores, reads and rewrites them as
// probabilities and reads those again: 4 passes over a [B*heads][T][T] tensor (850 MB per layer at C4).  Here the scores never
// leave the chip; the only large write is P itself, which the backward pass wants (lp_attn_dscores, lp_gemm_tn).
//
// One workgroup = one (image, head) and 128 query rows (a wave owns 32 of them); keys / values stream through LDS in tiles of
// 64.  Everything is computed TRANSPOSED, S^T = K Q^T and O^T = V^T P^T, because of how the 32x32 MFMA lays out its result:
// lane l holds column l%32 and 16 of the 32 rows.  With queries as columns a lane owns ONE query and 32 of a tile's 64 keys, so
//   * the soft-max row reductions are per-lane loops plus one exchange with lane l^32 (no cross-lane butterflies), and
//   * the probabilities a lane just computed are, register for register, the B operand of the O^T product: the contraction
//     index of an MFMA may be visited in any order as long as both operands agree, so V^T is simply fetched in the order the
//     accumulator layout dictates (ds_read_b64_tr_b16 picks its 4 source rows per lane group freely).
// Two passes over the keys: pass 1 finds each query's running maximum and exp-sum (online, fp32), pass 2 recomputes the scores,
// writes the normalised probabilities (staged through LDS so they leave as full 128-B lines) and accumulates O^T.  Recomputing
// Q K^T costs 8 MFMAs per tile and wave; storing the scores instead would cost the traffic this kernel exists to avoid.
#include "lp_common.h"

namespace lp {

constexpr int kAQ = 128;          // query rows per workgroup
constexpr int kAK = 64;           // keys per tile
constexpr int kAD = 64;           // head dimension
constexpr int kLDK = kAD + 8;     // K tile pitch: 144 B, conflict-free 16-B fragment reads
constexpr int kLDV = kAD + 32;    // V tile pitch: +64 B so the 4 rows of a transpose read fall in distinct bank quarters
constexpr int kLDP = kAK + 8;     // P staging pitch

// 2^x as ONE instruction (v_exp_f32); the soft-max below works in base 2 - scores x (scale log2 e), maxima and sums in that unit - so an
// exponential is a subtraction (or one FMA from the raw score) and a v_exp, not multiply + subtract + multiply + v_exp (round 6)
__device__ __forceinline__ float exp2_fast(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_exp2f(x);
#else
    return exp2f(x);
#endif
}

struct AttnArgs {
    const unsigned short* qkv;  // [B*T][ld] bf16 token rows; Q at column h*64, K at k_off + h*64, V at v_off + h*64
    int ld, k_off, v_off;
    int nh, T, qtiles;
    float scale;
    unsigned short* p;          // [B*nh][T][ldp] bf16 probabilities (pad columns [T, ldp) zeroed)
    int ldp;
    unsigned short* out;        // [B*T][ldo] bf16, head h at column h*64
    int ldo;
};

// WRITE_P = false (inference: a.p == nullptr): the probabilities are used for O and dropped - nothing of size T x T is written.
template <bool WRITE_P>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned short sK[kAK * kLDK];
    __shared__ __attribute__((aligned(16))) unsigned short sV[kAK * kLDV];
    __shared__ __attribute__((aligned(16))) unsigned short sP[kAQ * kLDP];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, half = lane >> 5;
    const int z = blockIdx.x / a.qtiles, qt = blockIdx.x - z * a.qtiles;
    const int b = z / a.nh, h = z - b * a.nh;
    const int T = a.T;
    const int q0 = qt * kAQ;
    const float sc2 = a.scale * 1.4426950408889634f;   // scores in base-2 units
    const unsigned short* Qp = a.qkv + (size_t)b * T * a.ld + h * kAD;
    const unsigned short* Kp = Qp + a.k_off;
    const unsigned short* Vp = Qp + a.v_off;
    const bool active = q0 + wave * 32 < T;  // wave-uniform: a wave whose 32 queries are all past T only helps with the staging

    // this lane's query row as the B operand of S^T = K Q^T: 8 consecutive d per k-slice (rows past T alias the last row)
    bf16x8 qf[kAD / 16];
    {
        int q = q0 + wave * 32 + col;
        if (q >= T) q = T - 1;
        const unsigned short* qr = Qp + (size_t)q * a.ld + half * 8;
#pragma unroll
        for (int kk = 0; kk < kAD / 16; ++kk) qf[kk] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u16x8*>(qr + kk * 16));
    }

    // K / V tile staging: thread -> 16-B chunk (tid & 7) of rows (tid >> 3) and (tid >> 3) + 32
    const int srow = tid >> 3, schunk = tid & 7;
    u16x8 rk[2], rv[2];
    auto fetch = [&](const unsigned short* base, int t, u16x8 (&r)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int kv = t * kAK + srow + 32 * i;
            u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (kv < T) v = *reinterpret_cast<const u16x8*>(base + (size_t)kv * a.ld + schunk * 8);
            r[i] = v;
        }
    };
    auto stage = [&](unsigned short* dst, int ldd, const u16x8 (&r)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<u16x8*>(dst + (srow + 32 * i) * ldd + schunk * 8) = r[i];
    };

    // S^T tile of this wave: [64 keys][32 queries] = 2 accumulator blocks; reg e of block blk is key blk*32 + (e&3) + 8*(e>>2) + 4*half
    f32x16 s[2];
    auto scores = [&]() {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
            for (int e = 0; e < 16; ++e) s[blk][e] = 0.f;
#pragma unroll
            for (int kk = 0; kk < kAD / 16; ++kk) {
                const bf16x8 kf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u16x8*>(&sK[(blk * 32 + col) * kLDK + kk * 16 + half * 8]));
                s[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], s[blk], 0, 0, 0);
            }
        }
    };

    const int n_kv = (T + kAK - 1) / kAK;
    // ---- pass 1: running maximum m and exp-sum l of this lane's half of every key tile
    float m = -INFINITY, l = 0.f;
    fetch(Kp, 0, rk);
    for (int t = 0; t < n_kv; ++t) {
        __syncthreads();  // the previous tile's fragment reads are done
        stage(sK, kLDK, rk);
        __syncthreads();
        if (t + 1 < n_kv) fetch(Kp, t + 1, rk);
        if (active) {
            scores();
            float tmax = -INFINITY;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int kv = t * kAK + blk * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                    s[blk][e] = kv < T ? s[blk][e] * sc2 : -INFINITY;
                    tmax = fmaxf(tmax, s[blk][e]);
                }
            const float mn = fmaxf(m, tmax);
            if (mn > -INFINITY) {  // (a lane's half of the ragged last tile may hold no valid key at all)
                float sum = 0.f;
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int e = 0; e < 16; ++e) sum += exp2_fast(s[blk][e] - mn);
                l = l * exp2_fast(m - mn) + sum;
                m = mn;
            }
        }
    }
    {   // merge the two halves of each query (lanes l and l^32)
        const float mo = __shfl_xor(m, 32, 64), lo = __shfl_xor(l, 32, 64);
        const float mt = fmaxf(m, mo);
        l = (m > -INFINITY ? l * exp2_fast(m - mt) : 0.f) + (mo > -INFINITY ? lo * exp2_fast(mo - mt) : 0.f);
        m = mt;
    }
    const float inv_l = active ? 1.f / l : 0.f;

    // ---- pass 2: probabilities (out through LDS) and O^T = V^T P^T
    f32x16 o[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[blk][e] = 0.f;
    const int fq = lane >> 4, fi = lane & 15;
    const int vrow = 4 * (fq >> 1) + (fi >> 2), vcol = 16 * (fq & 1) + 4 * (fi & 3);  // transpose-read source of this lane
    // tiles that only hold pad columns of the stored P get zeros
    const int n_p = (WRITE_P && (a.ldp + kAK - 1) / kAK > n_kv) ? (a.ldp + kAK - 1) / kAK : n_kv;
    __syncthreads();
    fetch(Kp, 0, rk);
    fetch(Vp, 0, rv);
    for (int t = 0; t < n_p; ++t) {
        stage(sK, kLDK, rk);
        stage(sV, kLDV, rv);
        __syncthreads();
        if (t + 1 < n_kv) {
            fetch(Kp, t + 1, rk);
            fetch(Vp, t + 1, rv);
        }
        unsigned pk[2][8];  // the tile's probabilities as packed bf16 pairs: pk[blk][e/2] = regs e, e+1
        if (active && t < n_kv) {
            scores();
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const int kv = t * kAK + blk * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                    const float p0 = kv < T ? exp2_fast(fmaf(s[blk][e], sc2, -m)) * inv_l : 0.f;
                    const float p1 = kv + 1 < T ? exp2_fast(fmaf(s[blk][e + 1], sc2, -m)) * inv_l : 0.f;
                    pk[blk][e >> 1] = pack_bf16x2(p0, p1);
                }
            // O^T += V^T P^T: k-slab j contracts the 16 keys held in regs 8*(j&1) .. +7 of block j>>1 (both halves)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
                const u32x4_t pw = {pk[j >> 1][4 * (j & 1)], pk[j >> 1][4 * (j & 1) + 1], pk[j >> 1][4 * (j & 1) + 2], pk[j >> 1][4 * (j & 1) + 3]};
                const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
                const int kvb = 32 * (j >> 1) + 16 * (j & 1);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const unsigned short* vp = &sV[(kvb + vrow) * kLDV + db * 32 + vcol];
                    const s16x4_t lo = lds_read_tr16(vp), hi = lds_read_tr16(vp + 8 * kLDV);
                    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
                    const s16x8_t vv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vv), pf, o[db], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int e = 0; e < 8; ++e) pk[blk][e] = 0u;
        }
        if (WRITE_P) {
            // stage P[query][key]: regs e .. e+3 are 4 consecutive keys (8 B)
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
                    const u32x2_t w = {pk[blk][2 * g4], pk[blk][2 * g4 + 1]};
                    *reinterpret_cast<u32x2_t*>(&sP[(wave * 32 + col) * kLDP + blk * 32 + 8 * g4 + 4 * half]) = w;
                }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = i * 32 + srow, q = q0 + row, c = t * kAK + schunk * 8;
                if (q < T && c < a.ldp)
                    *reinterpret_cast<u16x8*>(a.p + ((size_t)z * T + q) * a.ldp + c) = *reinterpret_cast<const u16x8*>(&sP[row * kLDP + schunk * 8]);
            }
        }
        __syncthreads();  // sP, sK, sV are free again
    }

    // O[q][h*64 + d]: reg e of block db is d = db*32 + (e&3) + 8*(e>>2) + 4*half -> 4 consecutive d (8 B) per lane and piece.  A lane owns a
    // query ROW, so written straight from the registers every store instruction touched 32 rows with 16 B each: 66 of the kernel's 407 us for
    // a tensor a tenth of P's size (timing builds, profiles/r06x_attn_probe.txt).  The wave's 32 rows go through its own rows of sP (free
    // since the last tile's closing barrier) and leave as 16-B pieces, 8 lanes per 128-B row (round 6).
    if (active) {
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
                const u32x2_t w = {pack_bf16x2(o[db][4 * g4], o[db][4 * g4 + 1]), pack_bf16x2(o[db][4 * g4 + 2], o[db][4 * g4 + 3])};
                *reinterpret_cast<u32x2_t*>(&sP[(wave * 32 + col) * kLDP + db * 32 + 8 * g4 + 4 * half]) = w;
            }
        __builtin_amdgcn_wave_barrier();   // (lanes exchange through the wave's own rows: lock step on the device)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = j * 8 + (lane >> 3), q = q0 + wave * 32 + row;
            if (q < T)
                *reinterpret_cast<u16x8*>(a.out + ((size_t)b * T + q) * a.ldo + h * kAD + (lane & 7) * 8) =
                    *reinterpret_cast<const u16x8*>(&sP[(wave * 32 + row) * kLDP + (lane & 7) * 8]);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Backward, key / value side:  dS = scale * P o (dO V^T - D)  (written: the dQ product reads it),  dV = P^T dO,  dK = dS^T Q
// in ONE pass over the stored probabilities (the composition lp_attn_dscores + 2 x lp_gemm_tn reads P twice and dS twice).
//
// Mirror image of the forward kernel: a workgroup owns 128 KEYS of one (image, head) (a wave 32 of them) and streams the queries
// through LDS in tiles of 64.  dP = dO V^T is computed untransposed, so in the MFMA result layout a lane owns ONE key and 32 of a
// tile's 64 queries; the stored P tile comes out of LDS in exactly that layout through ds_read_b64_tr_b16 (4 consecutive queries
// of one key per read), dS is formed in registers, and both P and dS are - register for register - the B operands of
// dV^T += dO^T P and dK^T += Q^T dS (contraction over the queries, visited in accumulator order; dO^T / Q^T are fetched in that
// order by transpose reads of the [query][d] tiles).  dS replaces P in the LDS tile in place (a wave only touches its own 32
// columns) and leaves as full rows.  The next tile's global loads are in flight (in registers) while the current one is computed.
constexpr int kBQ = 64;            // queries per tile
constexpr int kBK2 = 128;          // keys per workgroup
constexpr int kLDT = kAD + 32;     // dO / Q tile pitch (transpose-read source: +64 B)
constexpr int kLDP2 = kBK2 + 32;   // P / dS tile pitch

struct AttnBwdArgs {
    const unsigned short* qkv;   // token rows: Q at column h*64, V at v_off + h*64
    int ld, v_off;
    const unsigned short* d_out; // [B*T][ld_do], head h at column h*64
    int ld_do;
    const unsigned short* p;     // [B*nh][T][ldp]
    int ldp;
    const float* d_rows;         // [B*T][nh]: rowsum(dO o O)
    int nh, T, ktiles;
    float scale;
    unsigned short* ds;          // [B*nh][T][ldp] (pad columns zeroed)
    unsigned short* dqkv;        // token rows of pitch ld_dqkv: dK at dk_off + h*64, dV at dv_off + h*64
    int ld_dqkv, dk_off, dv_off;
};

// (Round 3 also built this kernel WITHOUT the stored probabilities - rebuilt per tile as exp(scale * Q K^T - m) / l from per-query statistics
// the forward pass kept: 8 more MFMAs + 32 exponentials per tile and wave in a kernel already at 256 VGPRs.  Forward 433 -> 361 us per layer,
// this kernel 476 -> 571 us, C4 3790 -> 3745 frames/s in one call (profiles/archive/r03x_vit_kernel_stats.txt): slower, so it left the tree in round 4;
// the source is kept in profiles/retired/r04_attn_with_lse_recompute.hip.txt.)
__global__ __launch_bounds__(256, 2) void attn_bwd_kv_kernel(AttnBwdArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned short sDO[kBQ * kLDT];
    __shared__ __attribute__((aligned(16))) unsigned short sQ[kBQ * kLDT];
    __shared__ __attribute__((aligned(16))) unsigned short sP[kBQ * kLDP2];
    __shared__ __attribute__((aligned(16))) float sD[kBQ];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, half = lane >> 5;
    const int z = blockIdx.x / a.ktiles, kt = blockIdx.x - z * a.ktiles;
    const int b = z / a.nh, h = z - b * a.nh;
    const int T = a.T;
    const int kv0 = kt * kBK2;
    const unsigned short* Qp = a.qkv + (size_t)b * T * a.ld + h * kAD;
    const unsigned short* Vp = Qp + a.v_off;
    const unsigned short* DOp = a.d_out + (size_t)b * T * a.ld_do + h * kAD;
    const unsigned short* Pp = a.p + (size_t)z * T * a.ldp;
    const bool active = kv0 + wave * 32 < T;

    // this lane's key row of V as the B operand of dP = dO V^T (rows past T alias the last row: their P is zero)
    bf16x8 vf[kAD / 16];
    {
        int kv = kv0 + wave * 32 + col;
        if (kv >= T) kv = T - 1;
        const unsigned short* vr = Vp + (size_t)kv * a.ld + half * 8;
#pragma unroll
        for (int kk = 0; kk < kAD / 16; ++kk) vf[kk] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u16x8*>(vr + kk * 16));
    }

    // staging: dO / Q tiles -> thread owns chunk (tid & 7) of rows (tid >> 3) + 32 i; P tile -> chunk (tid & 15) of rows (tid >> 4) + 16 i
    const int srow = tid >> 3, schunk = tid & 7, prow = tid >> 4, pchunk = tid & 15;
    const bool pcol_ok = kv0 + pchunk * 8 < a.ldp;
    u16x8 rdo[2], rq[2], rp[4];
    float rd = 0.f;
    auto fetch = [&](int t) {
        const u16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = t * kBQ + srow + 32 * i;
            rdo[i] = q < T ? *reinterpret_cast<const u16x8*>(DOp + (size_t)q * a.ld_do + schunk * 8) : zero;
            rq[i] = q < T ? *reinterpret_cast<const u16x8*>(Qp + (size_t)q * a.ld + schunk * 8) : zero;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = t * kBQ + prow + 16 * i;
            rp[i] = (q < T && pcol_ok) ? *reinterpret_cast<const u16x8*>(Pp + (size_t)q * a.ldp + kv0 + pchunk * 8) : zero;
        }
        if (tid < kBQ) {
            const int q = t * kBQ + tid;
            rd = q < T ? a.d_rows[((size_t)b * T + q) * a.nh + h] : 0.f;
        }
    };
    auto stage = [&](int t) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<u16x8*>(&sDO[(srow + 32 * i) * kLDT + schunk * 8]) = rdo[i];
            *reinterpret_cast<u16x8*>(&sQ[(srow + 32 * i) * kLDT + schunk * 8]) = rq[i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u16x8*>(&sP[(prow + 16 * i) * kLDP2 + pchunk * 8]) = rp[i];
        if (tid < kBQ) sD[tid] = rd;
    };

    f32x16 dv[2], dk[2];  // dV^T, dK^T: [d = db*32 + rows][key = this lane's]
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) dv[db][e] = dk[db][e] = 0.f;
    const int fq = lane >> 4, fi = lane & 15;
    const int trow = 4 * (fq >> 1) + (fi >> 2), tcol = 16 * (fq & 1) + 4 * (fi & 3);  // transpose-read source of this lane

    const int n_q = (T + kBQ - 1) / kBQ;
    fetch(0);
    for (int t = 0; t < n_q; ++t) {
        stage(t);
        __syncthreads();
        if (t + 1 < n_q) fetch(t + 1);
        if (active) {
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
            typedef __attribute__((ext_vector_type(8))) short s16x8_t;
            unsigned pk[2][8];   // dS of this tile as packed bf16 pairs (regs e, e+1 = two consecutive queries)
            s16x4_t pr[2][4];    // P: [block][e >> 2] = 4 consecutive queries of this lane's key
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                f32x16 dp;
#pragma unroll
                for (int e = 0; e < 16; ++e) dp[e] = 0.f;
#pragma unroll
                for (int kk = 0; kk < kAD / 16; ++kk) {
                    const bf16x8 of = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u16x8*>(&sDO[(blk * 32 + col) * kLDT + kk * 16 + half * 8]));
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(of, vf[kk], dp, 0, 0, 0);
                }
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int q0 = blk * 32 + 8 * g4;  // + 4 * half + (0..3)
                    pr[blk][g4] = lds_read_tr16(&sP[(q0 + trow) * kLDP2 + wave * 32 + tcol]);
                    const f32x4 d4 = *reinterpret_cast<const f32x4*>(&sD[q0 + 4 * half]);
                    float s4[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        s4[i] = a.scale * bf16_to_f32((unsigned short)pr[blk][g4][i]) * (dp[4 * g4 + i] - d4[i]);
                    pk[blk][2 * g4] = pack_bf16x2(s4[0], s4[1]);
                    pk[blk][2 * g4 + 1] = pack_bf16x2(s4[2], s4[3]);
                }
            }
            // dV^T += dO^T P and dK^T += Q^T dS: k-slab j = the 16 queries in regs 8*(j&1) .. +7 of block j>>1 (both halves)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const s16x8_t pv = {pr[j >> 1][2 * (j & 1)][0], pr[j >> 1][2 * (j & 1)][1], pr[j >> 1][2 * (j & 1)][2], pr[j >> 1][2 * (j & 1)][3],
                                    pr[j >> 1][2 * (j & 1) + 1][0], pr[j >> 1][2 * (j & 1) + 1][1], pr[j >> 1][2 * (j & 1) + 1][2],
                                    pr[j >> 1][2 * (j & 1) + 1][3]};
                const bf16x8 pf = __builtin_bit_cast(bf16x8, pv);
                const u32x4_t sw = {pk[j >> 1][4 * (j & 1)], pk[j >> 1][4 * (j & 1) + 1], pk[j >> 1][4 * (j & 1) + 2], pk[j >> 1][4 * (j & 1) + 3]};
                const bf16x8 sf = __builtin_bit_cast(bf16x8, sw);
                const int qb = 32 * (j >> 1) + 16 * (j & 1);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const unsigned short* op = &sDO[(qb + trow) * kLDT + db * 32 + tcol];
                    const s16x4_t olo = lds_read_tr16(op), ohi = lds_read_tr16(op + 8 * kLDT);
                    const s16x8_t ov = {olo[0], olo[1], olo[2], olo[3], ohi[0], ohi[1], ohi[2], ohi[3]};
                    dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ov), pf, dv[db], 0, 0, 0);
                    const unsigned short* qp = &sQ[(qb + trow) * kLDT + db * 32 + tcol];
                    const s16x4_t qlo = lds_read_tr16(qp), qhi = lds_read_tr16(qp + 8 * kLDT);
                    const s16x8_t qv = {qlo[0], qlo[1], qlo[2], qlo[3], qhi[0], qhi[1], qhi[2], qhi[3]};
                    dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, qv), sf, dk[db], 0, 0, 0);
                }
            }
            // dS over P, in place: this wave's 32 columns only
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int q = blk * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                    sP[q * kLDP2 + wave * 32 + col] = (unsigned short)((pk[blk][e >> 1] >> (16 * (e & 1))) & 0xffffu);
                }
        } else {  // keys past T: the probabilities there are zero, so is dS
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int e = 0; e < 16; ++e) sP[(blk * 32 + (e & 3) + 8 * (e >> 2) + 4 * half) * kLDP2 + wave * 32 + col] = 0;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = t * kBQ + prow + 16 * i;
            if (q < T && pcol_ok)
                *reinterpret_cast<u16x8*>(a.ds + ((size_t)z * T + q) * a.ldp + kv0 + pchunk * 8) =
                    *reinterpret_cast<const u16x8*>(&sP[(prow + 16 * i) * kLDP2 + pchunk * 8]);
        }
        __syncthreads();  // the tiles are free for the next staging
    }

    // dV / dK rows: a lane owns a KEY row (8-B pieces of it), so written straight from the registers every store instruction touches 32 rows
    // with 16 B each - the forward kernel's O store in the same form cost a sixth of that kernel (profiles/r06x_attn_probe.txt).  With 16-B
    // aligned rows the wave's 32 rows of each tensor go through its own slice of sP (free since the loop's closing barrier; pitch 72) and leave as
    // 16-B pieces, 8 lanes per 128-B row (round 6); other pitches keep the direct form.
    if (active && ((a.ld_dqkv | a.dk_off | a.dv_off) & 7) == 0) {
        unsigned short* stg = sP + wave * (32 * 72);
#pragma unroll
        for (int which = 0; which < 2; ++which) {
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
                    const f32x16& src = which == 0 ? dv[db] : dk[db];
                    const u32x2_t w = {pack_bf16x2(src[4 * g4], src[4 * g4 + 1]), pack_bf16x2(src[4 * g4 + 2], src[4 * g4 + 3])};
                    *reinterpret_cast<u32x2_t*>(&stg[col * 72 + db * 32 + 8 * g4 + 4 * half]) = w;
                }
            __builtin_amdgcn_wave_barrier();   // (lanes exchange through the wave's own rows: lock step on the device)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = j * 8 + (lane >> 3), kvr = kv0 + wave * 32 + r;
                if (kvr < T)
                    *reinterpret_cast<u16x8*>(a.dqkv + ((size_t)b * T + kvr) * a.ld_dqkv + h * kAD + (which == 0 ? a.dv_off : a.dk_off) + (lane & 7) * 8) =
                        *reinterpret_cast<const u16x8*>(&stg[r * 72 + (lane & 7) * 8]);
            }
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }
    const int kv = kv0 + wave * 32 + col;
    if (active && kv < T) {
        unsigned short* row = a.dqkv + ((size_t)b * T + kv) * a.ld_dqkv + h * kAD;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
                const u32x2_t wv = {pack_bf16x2(dv[db][4 * g4], dv[db][4 * g4 + 1]), pack_bf16x2(dv[db][4 * g4 + 2], dv[db][4 * g4 + 3])};
                const u32x2_t wk = {pack_bf16x2(dk[db][4 * g4], dk[db][4 * g4 + 1]), pack_bf16x2(dk[db][4 * g4 + 2], dk[db][4 * g4 + 3])};
                *reinterpret_cast<u32x2_t*>(row + a.dv_off + db * 32 + 8 * g4 + 4 * half) = wv;
                *reinterpret_cast<u32x2_t*>(row + a.dk_off + db * 32 + 8 * g4 + 4 * half) = wk;
            }
    }
}

}  // namespace lp

extern "C" int lp_attn_fwd(const void* qkv_bf16, int ld_qkv, int k_off, int v_off, int B, int nh, int T, float scale, void* p_bf16, int ldp,
                           void* out_bf16, int ldo, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(qkv_bf16 && out_bf16 && B > 0 && nh > 0 && T > 0 && (p_bf16 == nullptr || ldp >= T) && ldo >= nh * kAD && k_off >= 0 &&
               v_off >= 0 && ld_qkv >= nh * kAD);
    if (p_bf16 == nullptr) ldp = 8;  // unused
    if (ld_qkv % 8 != 0 || k_off % 8 != 0 || v_off % 8 != 0 || ldp % 8 != 0 || ldo % 4 != 0) return LP_ERR_UNSUPPORTED;
    const int qtiles = (T + kAQ - 1) / kAQ;
    const long long wgs = (long long)B * nh * qtiles;
    if (wgs >= (1LL << 31)) return LP_ERR_UNSUPPORTED;
    AttnArgs a{(const unsigned short*)qkv_bf16, ld_qkv, k_off, v_off, nh, T, qtiles, scale, (unsigned short*)p_bf16, ldp,
               (unsigned short*)out_bf16, ldo};
    if (p_bf16) hipLaunchKernelGGL(attn_fwd_kernel<true>, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(attn_fwd_kernel<false>, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, a);
    return launch_status();
}

extern "C" int lp_attn_bwd_kv(const void* qkv_bf16, int ld_qkv, int v_off, const void* d_out_bf16, int ld_do, const void* p_bf16, int ldp,
                              const float* d_rows, int B, int nh, int T, float scale, void* ds_bf16, void* dqkv_bf16, int ld_dqkv, int dk_off,
                              int dv_off, lp_stream_t stream) {
    using namespace lp;
    LP_REQUIRE(qkv_bf16 && d_out_bf16 && p_bf16 && d_rows && ds_bf16 && dqkv_bf16 && B > 0 && nh > 0 && T > 0 && ldp >= T &&
               ld_qkv >= nh * kAD && ld_do >= nh * kAD && ld_dqkv >= nh * kAD && v_off >= 0 && dk_off >= 0 && dv_off >= 0);
    if (ld_qkv % 8 != 0 || v_off % 8 != 0 || ld_do % 8 != 0 || ldp % 8 != 0 || ld_dqkv % 4 != 0 || dk_off % 4 != 0 || dv_off % 4 != 0)
        return LP_ERR_UNSUPPORTED;
    const int ktiles = (ldp + kBK2 - 1) / kBK2;  // every column of dS up to its pitch is written (pad columns: zeros)
    const long long wgs = (long long)B * nh * ktiles;
    if (wgs >= (1LL << 31)) return LP_ERR_UNSUPPORTED;
    AttnBwdArgs a{(const unsigned short*)qkv_bf16, ld_qkv, v_off, (const unsigned short*)d_out_bf16, ld_do, (const unsigned short*)p_bf16, ldp,
                  d_rows, nh, T, ktiles, scale, (unsigned short*)ds_bf16, (unsigned short*)dqkv_bf16, ld_dqkv, dk_off, dv_off};
    hipLaunchKernelGGL(attn_bwd_kv_kernel, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, a);
    return launch_status();
}

