// Go / no-go probe for round 4's first kernel item: does a store pass hide under matrix work when TWO
// independent 4-wave workgroups share a CU, where one 8-wave workgroup (all waves in the same phase) leaves the matrix pipe idle?
//
// Synthetic tile loop with the instruction mix and the memory pattern of conv_pipe_kernel's forward on a 1x1 layer - direct-to-LDS operand
// ring with counted vmcnt + raw barriers, swizzled ds_read_b128 fragments, v_mfma_f32_32x32x16_bf16 with swapped roles, and a store pass
// (convert, stage through LDS, full-line global stores, per-thread BatchNorm-like sums) - in two shapes of the same 256 x 128 tile:
//   A  (today)    512 threads, 8 waves of  64 x 64, K steps of 64: 3 x 48 KB ring, one workgroup per CU
//   B  (proposed) 256 threads, 4 waves of 128 x 64, K steps of 32: 3 x 24 KB ring, two workgroups per CU
// The numbers it produces are arithmetic nonsense (operands are whatever the buffers hold); only the time per tile matters.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tile_probe profiles/probe/tile_probe.hip && ./tile_probe
// prints, per shape and K in {64, 256, 1024}, microseconds per tile and CU with and without the store pass, and the TFLOP/s that implies.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((address_space(3))) void* lds_ptr;

#define RAW_BARRIER()                      \
    do {                                   \
        asm volatile("" ::: "memory");     \
        __builtin_amdgcn_s_barrier();      \
        asm volatile("" ::: "memory");     \
    } while (0)
#define WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t r, void* lds_wave_base, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)lds_wave_base, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ unsigned pack2(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// X: [rows][K] bf16 (pixels), W: [128][K] bf16, Y: [rows][128] bf16.  Every workgroup walks `tiles` tiles of 256 rows; `reuse` consecutive
// tiles read the same rows of X (the N tiles of a wide layer: L2 hits), x_rows = rows the X buffer holds (the walk wraps around it).
template <int WAVES, int KB, bool EPI>
__global__ __launch_bounds__(WAVES * 64, WAVES == 4 ? 2 : 1) void tile_probe(const unsigned short* __restrict__ X, const unsigned short* __restrict__ W,
                                                         unsigned short* __restrict__ Y, float* __restrict__ sums, int K, int tiles, int reuse,
                                                         unsigned x_rows) {
    constexpr int MT = WAVES == 8 ? 2 : 4, NT = 2;            // 32-row / 32-column MFMA blocks per wave
    constexpr int ROWB = KB * 2;                               // bytes per operand row in a stage
    constexpr int CH = ROWB / 16;                              // 16-B chunks per row (8 or 4)
    constexpr int RPI = 1024 / ROWB;                           // rows one wave load instruction fills (8 or 16)
    constexpr int kStageA = 256 * ROWB, kStage = (256 + 128) * ROWB;
    constexpr int NLA = 256 / (RPI * WAVES), NLB = 128 / (RPI * WAVES);   // 4 and 2 in both shapes
    static_assert(NLA == 4 && NLB == 2, "six loads per thread and K step");
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * kStage];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = WAVES == 8 ? (wave & 3) : (wave & 1), wn = WAVES == 8 ? (wave >> 2) : (wave >> 1);
    const unsigned x_bytes = x_rows * (unsigned)K * 2u, w_bytes = 128u * (unsigned)K * 2u;
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(X), 0, x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(W), 0, w_bytes, 0x00020000);
    const int KT = K / KB;

    // loader: lane -> (row inside the instruction's rows, 16-B slot); the chunk it fetches is slot ^ swizzle(row)
    const int lrow = lane / CH, slot = lane % CH;
    auto swz = [](int row) { return KB == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3); };
    int ld_tile = 0, ld_kt = 0;                                // the loader's position in the walk (two K steps ahead of the MFMAs)
    auto issue = [&](int st) {
        unsigned char* dst = smem + st * kStage;
        const bool live = ld_tile < tiles;
        const unsigned a_row0 = live ? (unsigned)((((blockIdx.x * tiles + ld_tile) / reuse) * 256) % x_rows) : 0u;
#pragma unroll
        for (int i = 0; i < NLA; ++i) {
            const int r = (i * WAVES + wave) * RPI + lrow;                       // row of the tile
            const unsigned voff = live ? ((a_row0 + r) * (unsigned)K + (unsigned)((slot ^ swz(r)) * 8 + ld_kt * KB)) * 2u : ~0u;
            glds16(rx, dst + (i * WAVES + wave) * 1024, voff, 0u);
        }
#pragma unroll
        for (int i = 0; i < NLB; ++i) {
            const int r = (i * WAVES + wave) * RPI + lrow;
            const unsigned voff = live ? ((unsigned)r * (unsigned)K + (unsigned)((slot ^ swz(r)) * 8 + ld_kt * KB)) * 2u : ~0u;
            glds16(rw, dst + kStageA + (i * WAVES + wave) * 1024, voff, 0u);
        }
        if (++ld_kt == KT) ld_kt = 0, ++ld_tile;
    };

    // fragments
    const int fr = lane & 31, fg = lane >> 5;
    f32x16 acc[MT][NT];
    auto mma_stage = [&](int st) {
        const unsigned char* sb = smem + st * kStage;
#pragma unroll
        for (int kk = 0; kk < KB / 16; ++kk) {
            bf16x8 a[MT], b[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int row = wm * (MT * 32) + mt * 32 + fr;
                a[mt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u16x8*>(sb + row * ROWB + (((kk * 2 + fg) ^ swz(row)) << 4)));
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int row = wn * (NT * 32) + nt * 32 + fr;
                b[nt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u16x8*>(sb + kStageA + row * ROWB + (((kk * 2 + fg) ^ swz(row)) << 4)));
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[nt], a[mt], acc[mt][nt], 0, 0, 0);
        }
    };

    // store pass (as conv_pipe_kernel's epilogue_fwd): per 32-row block, convert, stage through the wave's corner of the stage consumed
    // last, read back 8 rows x 128 B per instruction, store full lines, accumulate two sums per channel
    const int pc = lane & 7, prow = lane >> 3;
    float s0[8], s1[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) s0[q] = s1[q] = 0.f;
    auto store_pass = [&](int tile_row0, unsigned char* stg_all) {
        constexpr int SROW = NT * 64 + 16;
        unsigned char* stg = stg_all + wave * (32 * SROW);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const u32x2 p = {pack2(acc[mt][nt][4 * j], acc[mt][nt][4 * j + 1]), pack2(acc[mt][nt][4 * j + 2], acc[mt][nt][4 * j + 3])};
                    *reinterpret_cast<u32x2*>(stg + fr * SROW + (nt * 32 + 8 * j + 4 * fg) * 2) = p;
                }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int row = ps * 8 + prow;
                const u16x8 w = *reinterpret_cast<const u16x8*>(stg + row * SROW + pc * 16);
                const unsigned m = (unsigned)tile_row0 + (unsigned)(wm * (MT * 32) + mt * 32 + row);
                *reinterpret_cast<u16x8*>(Y + (size_t)m * 128 + wn * 64 + pc * 8) = w;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float v = __uint_as_float(((unsigned)w[q]) << 16);
                    s0[q] += v;
                    s1[q] = fmaf(v, v, s1[q]);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    };

    issue(0);
    issue(1);
    int cur = 0;
    for (int t = 0; t < tiles; ++t) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;
        for (int kt = 0; kt < KT; ++kt) {
            WAIT_VM(6);                       // this step's loads have landed, the next step's six stay in flight
            RAW_BARRIER();
            issue(cur == 0 ? 2 : cur - 1);    // (steps past the walk fetch nothing: the count stays valid)
            mma_stage(cur);
            cur = cur == 2 ? 0 : cur + 1;
        }
        if (EPI) {
            RAW_BARRIER();                    // every wave is done reading the stage consumed last: it is the staging area until the next step's loads
            store_pass((blockIdx.x * tiles + t) * 256, smem + (cur == 0 ? 2 : cur - 1) * kStage);
        }
    }
    WAIT_VM(0);
    // keep the accumulators and the sums alive
    float t0 = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) t0 += s0[q] + s1[q];
    if (!EPI) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) t0 += acc[mt][nt][lane & 15];
    }
    if (t0 == 12345.678f) sums[tid] = t0;
}

// ---- LDS read rate per CU: ds_read_b128 (forward / data-gradient fragments) against 2 x ds_read_b64_tr_b16 (weight-gradient fragments) -----------
// 8 waves, every lane reads ITER fragments of 16 B from a 32-KB image with the kernels' row swizzle; bytes / cycle / CU = 8 x 64 x 16 x ITER / cycles.
// conv_wgrad_pipe_kernel issues 128 KB of transpose reads per 1024 MFMA cycles and K step: if this probe shows ~128 B/clk for the transpose
// form, that kernel is LDS-bound at half the matrix rate and only a bigger wave tile (fewer fragments per MFMA) can lift it.
typedef __attribute__((ext_vector_type(4))) short s16x4;
template <int MODE>
__global__ __launch_bounds__(512) void lds_rate(unsigned* out, long long* cycles, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char img[32 * 1024];
    for (int i = threadIdx.x; i < 32 * 1024 / 4; i += 512) reinterpret_cast<unsigned*>(img)[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned acc = 0;
    const long long t0 = clock64();
    if (MODE == 0) {   // fragment of 32 rows x 16 B: lane -> (row fr, half fg), chunk (2 kk + fg) ^ ((row >> 1) & 7), 128-B rows
        const int fr = lane & 31, fg = lane >> 5;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int row = wave * 32 + fr;
                const u16x8 v = *reinterpret_cast<const volatile u16x8*>(img + row * 128 + (((kk * 2 + fg) ^ ((row >> 1) & 7)) << 4));
                acc ^= v[0] ^ v[7];
            }
        }
    } else {           // transpose fragment: 2 reads of 8 B per lane, rows of 512 B (the weight gradient's [pixel][256 channels] image)
        const int fq = lane >> 4, fi = lane & 15;
        const int frow = (fq >> 1) * 8 + (fi >> 2), fcol = (fq & 1) * 16 + (fi & 3) * 4;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int e0 = (wave & 3) * 64 + fcol;
                const unsigned char* pa = img + (kk * 16 + frow) * 512 + (((e0 >> 3) ^ ((frow & 3) << 2)) << 4) + (e0 & 7) * 2;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)pa);
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pa + 4 * 512));
                acc ^= (unsigned)lo[0] ^ (unsigned)hi[3];
            }
        }
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_));                 \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

template <int WAVES, int KB, bool EPI>
static double run(const unsigned short* X, const unsigned short* W, unsigned short* Y, float* sums, int K, int tiles, int reuse, unsigned x_rows,
                  int grid) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {   // the first launch warms up
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((tile_probe<WAVES, KB, EPI>), dim3(grid), dim3(WAVES * 64), 0, 0, X, W, Y, sums, K, tiles, reuse, x_rows);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
    }
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3;   // us
}

int main() {
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const size_t x_bytes = 1536ull << 20, y_rows = (size_t)cus * 2 * 64 * 256;   // X: 1.5 GB (beyond the MALL), Y: room for 64 tiles per workgroup
    unsigned short *X, *W, *Y;
    float* sums;
    CK(hipMalloc(&X, x_bytes));
    CK(hipMalloc(&W, 128 * 1024 * 2));
    CK(hipMalloc(&Y, y_rows * 128 * 2));
    CK(hipMalloc(&sums, 4096));
    CK(hipMemset(X, 0x3c, x_bytes));      // bf16 0x3c3c ~ 0.0115: finite operands
    CK(hipMemset(W, 0x3c, 128 * 1024 * 2));
    printf("%d CUs.  us per tile and CU (256 x 128 x K), TFLOP/s over the chip\n", cus);
    printf("%-34s %6s %6s %10s %10s %9s %9s\n", "shape", "K", "reuse", "no store", "with", "TF/s no", "TF/s with");
    const int Ks[3] = {64, 256, 1024};
    for (int ki = 0; ki < 3; ++ki) {
        const int K = Ks[ki];
        for (int reuse = 1; reuse <= 8; reuse *= 8) {
            const unsigned x_rows = (unsigned)(x_bytes / ((size_t)K * 2));
            const int tiles_a = K >= 1024 ? 16 : 48;                       // per workgroup, shape A (one workgroup per CU)
            const double flop_tile = 2.0 * 256 * 128 * K;
            {
                const double u0 = run<8, 64, false>(X, W, Y, sums, K, tiles_a, reuse, x_rows, cus);
                const double u1 = run<8, 64, true>(X, W, Y, sums, K, tiles_a, reuse, x_rows, cus);
                printf("%-34s %6d %6d %10.2f %10.2f %9.0f %9.0f\n", "A: 8 waves x 64x64, kstep 64, 1 WG", K, reuse, u0 / tiles_a, u1 / tiles_a,
                       flop_tile * tiles_a * cus / u0 * 1e-6, flop_tile * tiles_a * cus / u1 * 1e-6);
            }
            {
                const int tiles_b = tiles_a / 2;                           // two workgroups per CU share the same number of tiles per CU
                const double u0 = run<4, 32, false>(X, W, Y, sums, K, tiles_b, reuse, x_rows, 2 * cus);
                const double u1 = run<4, 32, true>(X, W, Y, sums, K, tiles_b, reuse, x_rows, 2 * cus);
                printf("%-34s %6d %6d %10.2f %10.2f %9.0f %9.0f\n", "B: 4 waves x 128x64, kstep 32, 2 WG", K, reuse, u0 / tiles_a, u1 / tiles_a,
                       flop_tile * tiles_a * cus / u0 * 1e-6, flop_tile * tiles_a * cus / u1 * 1e-6);
            }
        }
    }
    // LDS read rates (one workgroup of 8 waves per CU; HIP events around the second launch)
    {
        long long* cyc;
        CK(hipMalloc(&cyc, cus * sizeof(long long)));
        const int iters = 16384;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        for (int mode = 0; mode < 2; ++mode) {
            float ms = 0.f;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0, 0));
                if (mode == 0) hipLaunchKernelGGL(lds_rate<0>, dim3(cus), dim3(512), 0, 0, (unsigned*)sums, cyc, iters);
                else hipLaunchKernelGGL(lds_rate<1>, dim3(cus), dim3(512), 0, 0, (unsigned*)sums, cyc, iters);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
            }
            const double bytes_cu = 512.0 * 16 * 4 * iters;
            printf("LDS %-26s %7.1f GB/s per CU = %6.1f B/clk at 2.4 GHz (%.1f us)\n", mode == 0 ? "ds_read_b128 fragments" : "2 x ds_read_b64_tr_b16",
                   bytes_cu / (ms * 1e-3) * 1e-9, bytes_cu / (ms * 1e-3) / 2.4e9, ms * 1e3);
        }
    }
    return 0;
}
