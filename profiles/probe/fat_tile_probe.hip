// Round 4, last probe: a DIFFERENT loop structure for round 5 to start from.  loop_probe.hip showed that the 256 x 128 tile of 8 waves x (64 x 64)
// tops out near 1100 - 1200 TFLOP/s: per K step of 64 it moves 48 KB through the direct-to-LDS path and reads 128 KB of fragments for 1031 cycles of
// MFMA per SIMD.  Here ONE workgroup of FOUR waves owns a 256 x 256 tile, a wave 128 x 128 = 4 x 4 MFMA blocks (256 accumulator registers): per K step
// of 64 the CU loads 64 KB (+33 %) and reads 128 KB of fragments (the same) for 2048 MFMA cycles per SIMD (2 x) - twice the arithmetic intensity on
// both LDS ports.  Two stages of 64 KB (K step 64) or four stages of 32 KB (K step 32).  Synthetic operands, L2-resident (every `reuse` tiles share
// their rows), no im2col; store pass: none, or packed 8-B stores straight from the accumulators (the cheapest possible form).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o fat_tile_probe profiles/probe/fat_tile_probe.hip && ./fat_tile_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((address_space(3))) void* lds_ptr;

#define RAW_BARRIER()                  \
    do {                               \
        asm volatile("" ::: "memory"); \
        __builtin_amdgcn_s_barrier();  \
        asm volatile("" ::: "memory"); \
    } while (0)

__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t r, void* lds_wave_base, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)lds_wave_base, 16, voff, 0u, 0, 0);
}
__device__ __forceinline__ unsigned pack2(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// KS = K step (64: rows of 128 B, 2 stages; 32: rows of 64 B, 4 stages); LOADS / LDSRD as in loop_probe; STORE 0 / 1
template <int KS, bool LOADS, bool LDSRD, bool STORE>
__global__ __launch_bounds__(256, 1) void fat_tile(const unsigned short* __restrict__ X, const unsigned short* __restrict__ W, unsigned short* __restrict__ Y,
                                                   float* __restrict__ sink, int K, int tiles, int reuse, unsigned x_rows) {
    constexpr int ROWB = KS * 2, NST = KS == 64 ? 2 : 4, kStageA = 256 * ROWB, kStage = 512 * ROWB;   // 64 KB / 32 KB per stage
    constexpr int KSL = KS / 16;                                                                      // k-slices of 16 per K step
    constexpr int RPI = 1024 / ROWB;                                                                  // rows a wave instruction fills (8 / 16)
    constexpr int CH = ROWB / 16;                                                                     // 16-B chunks per row (8 / 4)
    __shared__ __attribute__((aligned(16))) unsigned char smem[NST * kStage];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const unsigned x_bytes = x_rows * (unsigned)K * 2u, w_bytes = 256u * (unsigned)K * 2u;
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(X), 0, x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(W), 0, w_bytes, 0x00020000);
    const int KT = K / KS;
    const int lrow = lane / CH, slot = lane % CH;
    auto swz = [](int row) { return (row >> 1) & (CH - 1); };
    int ld_tile = 0, ld_kt = 0;
    auto issue = [&](int st) {
        if (!LOADS) return;
        unsigned char* dst = smem + st * kStage;
        const bool live = ld_tile < tiles;
        // reuse > 0: `reuse` CONSECUTIVE workgroups share their rows - consecutive workgroups sit on different XCDs, so the rows come from MALL / HBM
        // through the fabric, not from a shared L2; reuse < 0: -reuse workgroups OF THE SAME XCD (blockIdx % 8) share them: one L2 miss, the rest hits
        const unsigned grp = reuse > 0 ? (unsigned)((blockIdx.x * tiles + ld_tile) / reuse)
                                       : (unsigned)((((blockIdx.x & 7) * 64 + (blockIdx.x >> 3) / (-reuse)) * tiles) + ld_tile);
        const unsigned a_row0 = live ? (grp * 256u) % x_rows : 0u;
        constexpr int PER = 256 / RPI / 4;   // wave instructions per operand and wave (8 / 4)
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int r = (i * 4 + wave) * RPI + lrow;
            const unsigned ka = (unsigned)((slot ^ swz(r)) * 8 + ld_kt * KS);
            glds16(rx, dst + (i * 4 + wave) * 1024, live ? ((a_row0 + r) * (unsigned)K + ka) * 2u : ~0u);
            glds16(rw, dst + kStageA + (i * 4 + wave) * 1024, live ? ((unsigned)r * (unsigned)K + ka) * 2u : ~0u);
        }
        if (++ld_kt == KT) ld_kt = 0, ++ld_tile;
    };
    const int fr = lane & 31, fg = lane >> 5;
    f32x16 acc[4][4];
    float keep = 0.f;
    for (int s = 0; s < NST - 1; ++s) issue(s);
    int st = 0;
    for (int t = 0; t < tiles; ++t) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int kt = 0; kt < KT; ++kt) {
            // the oldest stage's loads have landed (this wave's; the barrier makes it everyone's) and everyone is done with the stage refilled next
            if (LOADS) {
                if (NST == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // 4 stages: two younger steps (2 x 8 instructions) stay in flight
            }
            RAW_BARRIER();
            issue((st + NST - 1) % NST);
            const unsigned char* sa = smem + st * kStage;
            const unsigned char* sb = sa + kStageA;
            bf16x8 af[4], bfv[4];
#pragma unroll
            for (int kk = 0; kk < KSL; ++kk) {
                const unsigned c = (unsigned)(kk * 2 + fg);
                if (LDSRD || (kt == 0 && kk == 0)) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int ra = wm * 128 + i * 32 + fr, rb = wn * 128 + i * 32 + fr;
                        af[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u16x8*>(sa + ra * ROWB + ((c ^ (unsigned)swz(ra)) << 4)));
                        bfv[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u16x8*>(sb + rb * ROWB + ((c ^ (unsigned)swz(rb)) << 4)));
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfv[j], af[i], acc[i][j], 0, 0, 0);
            }
            st = (st + 1) % NST;
        }
        if (STORE) {   // D[channel][pixel] (roles swapped): lane (pixel fr, half fg) holds channels 8 q + 4 fg + (0..3) of a 32 x 32 block in acc[4 q .. 4 q + 3]
            const size_t row0 = ((size_t)blockIdx.x * tiles + t) * 256;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const u32x2 pk = {pack2(acc[i][j][4 * q], acc[i][j][4 * q + 1]), pack2(acc[i][j][4 * q + 2], acc[i][j][4 * q + 3])};
                        *reinterpret_cast<u32x2*>(Y + (row0 + wm * 128 + i * 32 + fr) * 256 + wn * 128 + j * 32 + 8 * q + 4 * fg) = pk;
                    }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) keep += acc[i][j][0] + acc[i][j][15];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (keep == 12345.678f) sink[0] = keep;
}

// ---- producer / consumer wave specialisation on today's 256 x 128 tile -------------------------------------------------------------------
// What the first table shows: with the loads on, BOTH shapes stop at ~1100 - 1190 TFLOP/s, whatever the arithmetic intensity - a direct-to-LDS
// load costs the wave that ISSUES it ~110 cycles, during which that wave issues no MFMA, and waves in the same phase do not cover for each other.
// Here 8 waves = 4 CONSUMERS (one per SIMD: 128 x 64 each, 128 accumulator registers, fragment reads + MFMAs only) + 4 PRODUCERS (one per SIMD:
// they issue all 48 direct-to-LDS loads of a K step, 12 each, wait for them and meet the consumers at the one barrier per K step).  A kernel has
// ONE register allocation for all its waves, so two waves per SIMD cap the consumer at 256 registers: 128 x 64 is the largest tile that fits.
// 3-stage ring of 48 KB as in conv_pipe_kernel (loads of step g + 2 issued during step g).
template <bool LOADS, bool LDSRD, bool STORE>
__global__ __launch_bounds__(512, 1) void spec_tile(const unsigned short* __restrict__ X, const unsigned short* __restrict__ W, unsigned short* __restrict__ Y,
                                                    float* __restrict__ sink, int K, int tiles, int reuse, unsigned x_rows) {
    constexpr int KS = 64, ROWB = 128, kStageA = 256 * ROWB, kStage = (256 + 128) * ROWB, NST = 3;
    __shared__ __attribute__((aligned(16))) unsigned char smem[NST * kStage];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned x_bytes = x_rows * (unsigned)K * 2u, w_bytes = 128u * (unsigned)K * 2u;
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(X), 0, x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(W), 0, w_bytes, 0x00020000);
    const int KT = K / KS, steps = tiles * KT;
    auto swz = [](int row) { return (row >> 1) & 7; };
    if (wave >= 4) {   // ---- producer
        const int pw = wave - 4, lrow = lane >> 3, slot = lane & 7;
        int ld_tile = 0, ld_kt = 0;
        auto issue = [&](int st) {
            if (!LOADS) return;
            unsigned char* dst = smem + st * kStage;
            const bool live = ld_tile < tiles;
            // reuse > 0: `reuse` CONSECUTIVE workgroups share their rows - consecutive workgroups sit on different XCDs, so the rows come from MALL / HBM
        // through the fabric, not from a shared L2; reuse < 0: -reuse workgroups OF THE SAME XCD (blockIdx % 8) share them: one L2 miss, the rest hits
        const unsigned grp = reuse > 0 ? (unsigned)((blockIdx.x * tiles + ld_tile) / reuse)
                                       : (unsigned)((((blockIdx.x & 7) * 64 + (blockIdx.x >> 3) / (-reuse)) * tiles) + ld_tile);
        const unsigned a_row0 = live ? (grp * 256u) % x_rows : 0u;
#pragma unroll
            for (int i = 0; i < 8; ++i) {   // pixel rows: 32 wave instructions of 8 rows, 8 per producer
                const int r = (i * 4 + pw) * 8 + lrow;
                glds16(rx, dst + (i * 4 + pw) * 1024, live ? ((a_row0 + r) * (unsigned)K + (unsigned)((slot ^ swz(r)) * 8 + ld_kt * KS)) * 2u : ~0u);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {   // weight rows: 16 wave instructions, 4 per producer
                const int r = (i * 4 + pw) * 8 + lrow;
                glds16(rw, dst + kStageA + (i * 4 + pw) * 1024, live ? ((unsigned)r * (unsigned)K + (unsigned)((slot ^ swz(r)) * 8 + ld_kt * KS)) * 2u : ~0u);
            }
            if (++ld_kt == KT) ld_kt = 0, ++ld_tile;
        };
        issue(0);
        issue(1);
        int st = 0;
        for (int g = 0; g < steps; ++g) {
            if (LOADS) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // step g's loads have landed (step g + 1's 12 may still fly)
            RAW_BARRIER();
            issue((st + 2) % NST);
            st = (st + 1) % NST;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    // ---- consumer
    const int wm = wave & 1, wn = wave >> 1, fr = lane & 31, fg = lane >> 5;
    f32x16 acc[4][2];
    float keep = 0.f;
    int st = 0;
    for (int t = 0; t < tiles; ++t) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int kt = 0; kt < KT; ++kt) {
            RAW_BARRIER();
            const unsigned char* sa = smem + st * kStage;
            const unsigned char* sb = sa + kStageA;
            bf16x8 af[4], bfv[2];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const unsigned c = (unsigned)(kk * 2 + fg);
                if (LDSRD || (kt == 0 && kk == 0)) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int ra = wm * 128 + i * 32 + fr;
                        af[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u16x8*>(sa + ra * ROWB + ((c ^ (unsigned)swz(ra)) << 4)));
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int rb = wn * 64 + j * 32 + fr;
                        bfv[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u16x8*>(sb + rb * ROWB + ((c ^ (unsigned)swz(rb)) << 4)));
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfv[j], af[i], acc[i][j], 0, 0, 0);
            }
            st = (st + 1) % NST;
        }
        if (STORE) {
            const size_t row0 = ((size_t)blockIdx.x * tiles + t) * 256;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const u32x2 pk = {pack2(acc[i][j][4 * q], acc[i][j][4 * q + 1]), pack2(acc[i][j][4 * q + 2], acc[i][j][4 * q + 3])};
                        *reinterpret_cast<u32x2*>(Y + (row0 + wm * 128 + i * 32 + fr) * 128 + wn * 64 + j * 32 + 8 * q + 4 * fg) = pk;
                    }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) keep += acc[i][j][0] + acc[i][j][15];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (keep == 12345.678f) sink[0] = keep;
}

template <bool LOADS, bool LDSRD, bool STORE>
static double run_spec(const unsigned short* X, const unsigned short* W, unsigned short* Y, float* sink, int K, int tiles, int reuse, unsigned x_rows, int cus) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((spec_tile<LOADS, LDSRD, STORE>), dim3(cus), dim3(512), 0, 0, X, W, Y, sink, K, tiles, reuse, x_rows);
    hipEventRecord(e0);
    const int reps = 5;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((spec_tile<LOADS, LDSRD, STORE>), dim3(cus), dim3(512), 0, 0, X, W, Y, sink, K, tiles, reuse, x_rows);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    if (hipGetLastError() != hipSuccess) return -1.0;
    return ms * 1e3 / reps / tiles;
}

template <int KS, bool LOADS, bool LDSRD, bool STORE>
static double run(const unsigned short* X, const unsigned short* W, unsigned short* Y, float* sink, int K, int tiles, int reuse, unsigned x_rows, int cus) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((fat_tile<KS, LOADS, LDSRD, STORE>), dim3(cus), dim3(256), 0, 0, X, W, Y, sink, K, tiles, reuse, x_rows);
    hipEventRecord(e0);
    const int reps = 5;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((fat_tile<KS, LOADS, LDSRD, STORE>), dim3(cus), dim3(256), 0, 0, X, W, Y, sink, K, tiles, reuse, x_rows);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    if (hipGetLastError() != hipSuccess) return -1.0;
    return ms * 1e3 / reps / tiles;   // us per tile and CU
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const unsigned x_rows = 1u << 19;   // (x_rows * K * 2 must stay below 2^32: the buffer descriptor's range)
    unsigned short *X, *W, *Y;
    float* sink;
    hipMalloc(&X, (size_t)x_rows * 2304 * 2);
    hipMalloc(&W, (size_t)256 * 2304 * 2);
    hipMalloc(&Y, (size_t)cus * 64 * 256 * 256 * 2);
    hipMalloc(&sink, 64);
    hipMemset(X, 0x11, (size_t)x_rows * 2304 * 2);
    hipMemset(W, 0x22, (size_t)256 * 2304 * 2);
    printf("%d CUs; 256 x 256 x K tile, 4 waves x (128 x 128); us per tile and CU (TFLOP/s over the chip)\n", cus);
    for (int K : {256, 576, 1024, 2304}) {
        const int tiles = K <= 576 ? 48 : 16;
        const double fl = 2.0 * 256 * 256 * K * cus;
        auto tf = [&](double us) { return fl / us * 1e-6; };
        for (int reuse : {8}) {
            const double a = run<64, true, true, false>(X, W, Y, sink, K, tiles, reuse, x_rows, cus);
            const double b = run<64, false, true, false>(X, W, Y, sink, K, tiles, reuse, x_rows, cus);
            const double c = run<64, false, false, false>(X, W, Y, sink, K, tiles, reuse, x_rows, cus);
            const double d = run<64, true, true, true>(X, W, Y, sink, K, tiles, reuse, x_rows, cus);
            const double e = run<32, true, true, false>(X, W, Y, sink, K, tiles, reuse, x_rows, cus);
            const double f = run<32, true, true, true>(X, W, Y, sink, K, tiles, reuse, x_rows, cus);
            const double g = run<64, true, true, false>(X, W, Y, sink, K, tiles, -8, x_rows, cus);
            const double h = run<32, true, true, false>(X, W, Y, sink, K, tiles, -8, x_rows, cus);
            const double i1 = run<64, true, true, false>(X, W, Y, sink, K, tiles, 1, x_rows, cus);
            printf("K %4d reuse %d | K step 64, 2 stages: full %7.2f (%4.0f)  no loads %7.2f (%4.0f)  MFMA + barrier only %7.2f (%4.0f)  with stores %7.2f (%4.0f)"
                   " | K step 32, 4 stages: full %7.2f (%4.0f)  with stores %7.2f (%4.0f) | rows shared inside an XCD (8 CUs): K step 64 %7.2f (%4.0f)  K step 32 %7.2f (%4.0f)"
                   " | no sharing (HBM): %7.2f (%4.0f)\n",
                   K, reuse, a, tf(a), b, tf(b), c, tf(c), d, tf(d), e, tf(e), f, tf(f), g, tf(g), h, tf(h), i1, tf(i1));
        }
    }
    printf("\n256 x 128 x K tile, 4 consumer waves x (128 x 64) + 4 producer waves; us per tile and CU (TFLOP/s over the chip)\n");
    for (int K : {256, 576, 1024, 2304}) {
        const int tiles = K <= 576 ? 48 : 16;
        const double fl = 2.0 * 256 * 128 * K * cus;
        auto tf = [&](double us) { return fl / us * 1e-6; };
        const double a = run_spec<true, true, false>(X, W, Y, sink, K, tiles, 8, x_rows, cus);
        const double b = run_spec<false, true, false>(X, W, Y, sink, K, tiles, 8, x_rows, cus);
        const double c = run_spec<false, false, false>(X, W, Y, sink, K, tiles, 8, x_rows, cus);
        const double d = run_spec<true, true, true>(X, W, Y, sink, K, tiles, 8, x_rows, cus);
        const double e = run_spec<true, true, false>(X, W, Y, sink, K, tiles, 1, x_rows, cus);
        const double g = run_spec<true, true, false>(X, W, Y, sink, K, tiles, -8, x_rows, cus);
        const double h = run_spec<true, true, true>(X, W, Y, sink, K, tiles, -8, x_rows, cus);
        printf("K %4d | full %7.2f (%4.0f)  no loads %7.2f (%4.0f)  MFMA + barrier only %7.2f (%4.0f)  with stores %7.2f (%4.0f)  full, reuse 1 (operands from HBM) %7.2f (%4.0f)"
               " | rows shared inside an XCD (8 CUs): full %7.2f (%4.0f)  with stores %7.2f (%4.0f)\n",
               K, a, tf(a), b, tf(b), c, tf(c), d, tf(d), e, tf(e), g, tf(g), h, tf(h));
    }
    return 0;
}
