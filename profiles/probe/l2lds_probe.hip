// How fast does L2 feed LDS?  The per-tap ring form of conv_pipe_kernel moves 48 KB per K step and CU from L2 (mostly hits: every pixel row is
// read by N / 128 workgroups, every weight row by M / 256) into LDS with `buffer_load_dwordx4 ... lds`; its K step takes ~1950 cycles where the
// MFMAs need 1024 (profiles/r05k_conv_timing.txt).  This probe runs the ring's load pattern alone - one 512-thread workgroup per CU, 3 stages
// of 48 KB, two stages in flight, counted vmcnt + one s_barrier per step - over a buffer small enough to stay in every XCD's 4 MB L2 (all
// workgroups walk the same `ws` bytes from different phases), and reports bytes per clock and CU:
//   mode 0  LDS-DMA loads only                       mode 1  ... plus the fragment reads of the step (16 ds_read_b128 per wave)
//   mode 2  ... plus the step's 16 MFMAs per wave    mode 3  plain global_load_dwordx4 into registers (no LDS), same addresses
//   hipcc --offload-arch=gfx950 -O3 -o l2lds_probe profiles/probe/l2lds_probe.hip && ./l2lds_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void* lds_ptr;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kStage = 48 * 1024;
template <int MODE>
__global__ __launch_bounds__(512) void ring(const unsigned char* x, unsigned ws, int steps, unsigned long long* clk, float* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * kStage];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(x), 0, ws, 0x00020000);
    const unsigned nst = ws / kStage;
    unsigned pos = (blockIdx.x * 7u) % nst;   // phase of this workgroup in the shared working set
    u32x4 racc = {0, 0, 0, 0};
    auto issue = [&](int s) {
        unsigned char* dst = smem + (s % 3) * kStage;
        const unsigned so = pos * kStage;
        pos = pos + 1 == nst ? 0 : pos + 1;
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            const unsigned vo = (unsigned)((p * 8 + wave) * 1024 + lane * 16);
            if (MODE == 3) racc ^= __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(dst + (p * 8 + wave) * 1024), 16, vo, so, 0, 0);
        }
    };
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    issue(0);
    issue(1);
    for (int s = 0; s < steps; ++s) {
        if (MODE != 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue(s + 2);
        if (MODE == 1 || MODE == 2) {
            const unsigned char* src = smem + (s % 3) * kStage;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                bf16x8 a[2], b[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[i] = *reinterpret_cast<const bf16x8*>(src + ((wave & 3) * 64 + i * 32 + (lane & 31)) * 128 + (((kk * 2 + (lane >> 5)) ^ ((lane >> 1) & 7)) * 16));
                    b[i] = *reinterpret_cast<const bf16x8*>(src + 32768 + ((wave >> 2) * 64 + i * 32 + (lane & 31)) * 128 + (((kk * 2 + (lane >> 5)) ^ ((lane >> 1) & 7)) * 16));
                }
                if (MODE == 2) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i * 2 + j], 0, 0, 0);
                } else {
                    asm volatile("" ::"v"(a[0]), "v"(a[1]), "v"(b[0]), "v"(b[1]));
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) clk[blockIdx.x * 2] = t1 - t0, clk[blockIdx.x * 2 + 1] = r1 - r0;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) v += acc[i][0] + acc[i][7];
    if (v == 12345.f || (racc[0] ^ racc[1]) == 0x1234567u) sink[tid] = v;
}

template <int MODE>
static void run(const unsigned char* x, unsigned ws, int steps, int wgs, unsigned long long* clk, float* sink) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 3; ++it) {
        if (it == 2) CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((ring<MODE>), dim3(wgs), dim3(512), 0, 0, x, ws, steps, clk, sink);
    }
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[512];
    CK(hipMemcpy(h, clk, sizeof(unsigned long long) * 2 * wgs, hipMemcpyDeviceToHost));
    double cyc = 0, ticks = 0;
    for (int i = 0; i < wgs; ++i) cyc += h[2 * i], ticks += h[2 * i + 1];
    cyc /= wgs, ticks /= wgs;
    const double bytes = (double)(steps + 2) * kStage;
    printf("mode %d  working set %5.1f MB  wgs %3d: %8.1f us (events)  %8.0f cycles/WG at %.2f GHz  %6.0f cycles/step  %5.1f B/clk/CU  %6.2f TB/s in-kernel\n", MODE,
           ws / 1048576.0, wgs, ms * 1e3, cyc, cyc / ticks * 0.1, cyc / (steps + 2), bytes / cyc, bytes * wgs / (ticks * 1e-8) / 1e12);
}

int main() {
    const size_t cap = 1536ull << 20;
    unsigned char* x;
    unsigned long long* clk;
    float* sink;
    CK(hipMalloc(&x, cap)); CK(hipMalloc(&clk, 8192)); CK(hipMalloc(&sink, 4096));
    CK(hipMemset(x, 1, cap));
    const int steps = 3000;
    const unsigned sets[5] = {3u * kStage * 7, 2u << 20, 24u << 20, 192u << 20, 1536u << 20};   // L2 of an XCD: 4 MB; memory-side cache: 256 MB
    for (int si = 0; si < 5; ++si) {
        const unsigned ws = sets[si] / kStage * kStage;
        run<0>(x, ws, steps, 256, clk, sink);
        run<1>(x, ws, steps, 256, clk, sink);
        run<2>(x, ws, steps, 256, clk, sink);
        run<3>(x, ws, steps, 256, clk, sink);
    }
    run<0>(x, (2u << 20) / kStage * kStage, steps, 32, clk, sink);   // an eighth of the CUs: is the limit per CU or shared?
    run<0>(x, (2u << 20) / kStage * kStage, steps, 8, clk, sink);
    return 0;
}
