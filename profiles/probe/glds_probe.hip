// Probe for the direct-to-LDS pipeline the round-3 convolution kernel is built on (gfx950):
//  1. semantics of `buffer_load_dwordx4 ... lds` for an out-of-range vector offset (does the LDS slot receive zeros or stay stale?)
//  2. streaming rate HBM -> LDS of a 3-buffer ring with counted vmcnt + raw s_barrier, 1 workgroup of 512 threads per CU, with
//     1 or 2 stages in flight, with and without an output stream (mimics the HBM-bound 1x1 layers)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) void* lds_ptr;

__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t r, void* lds_wave_base, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)lds_wave_base, 16, voff, soff, 0, 0);
}

__global__ void sem_probe(const unsigned* x, unsigned bytes, unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned smem[64 * 4 * 2];
    for (int i = threadIdx.x; i < 64 * 4 * 2; i += 64) smem[i] = 0xAAAAAAAAu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(x), 0, bytes, 0x00020000);
    unsigned voff = threadIdx.x * 16;
    if (threadIdx.x & 1) voff = 0xffffffffu;          // out of range
    glds16(r, smem, voff, 0);
    unsigned voff2 = threadIdx.x * 16;                 // second half: in range + scalar offset of 1 KB
    glds16(r, smem + 256, voff2, 1024);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 4 * 2; i += 64) out[i] = smem[i];
}

constexpr int kStageBytes = 48 * 1024;                 // 512 threads x 6 x 16 B
template <int INFLIGHT, int STORE_KB>
__global__ __launch_bounds__(512) void stream_probe(const unsigned char* x, size_t per_wg_bytes, unsigned char* y, unsigned* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * kStageBytes];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned char* base = x + (size_t)blockIdx.x * per_wg_bytes;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(base), 0, (unsigned)per_wg_bytes, 0x00020000);
    const int nst = (int)(per_wg_bytes / kStageBytes);
    auto issue = [&](int s) {
        unsigned char* dst = smem + (s % 3) * kStageBytes;
#pragma unroll
        for (int p = 0; p < 6; ++p) glds16(r, dst + (p * 8 + wave) * 1024, (unsigned)((p * 8 + wave) * 1024 + lane * 16), (unsigned)s * kStageBytes);
    };
    u32x4 acc = {0, 0, 0, 0};
    unsigned char* yb = y + (size_t)blockIdx.x * (per_wg_bytes / kStageBytes) * (STORE_KB * 1024);
    for (int s = 0; s < INFLIGHT && s < nst; ++s) issue(s);
    for (int s = 0; s < nst; ++s) {
        if (INFLIGHT == 2) {
            if (s + 1 < nst) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (s + INFLIGHT < nst) issue(s + INFLIGHT);
        const unsigned char* src = smem + (s % 3) * kStageBytes;
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(src + (p * 8 + wave) * 1024 + lane * 16);
            acc ^= v;
        }
        if (STORE_KB > 0) {
#pragma unroll
            for (int p = 0; p < STORE_KB / 8; ++p)
                *reinterpret_cast<u32x4*>(yb + (size_t)s * (STORE_KB * 1024) + (p * 8 + wave) * 1024 + lane * 16) = acc;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

template <int INFLIGHT, int STORE_KB>
static void run_stream(const unsigned char* x, unsigned char* y, unsigned* sink, size_t total, int wgs) {
    const size_t per = total / wgs / kStageBytes * kStageBytes;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int it = 0; it < 4; ++it) {
        if (it == 1) hipEventRecord(e0, 0);
        hipLaunchKernelGGL((stream_probe<INFLIGHT, STORE_KB>), dim3(wgs), dim3(512), 0, 0, x, per, y, sink);
    }
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 3;
    const double rd = (double)per * wgs, wr = rd / kStageBytes * STORE_KB * 1024;
    printf("stream inflight=%d store=%2d KB/stage wgs=%d: %.3f ms  read %.2f TB/s  write %.2f TB/s  total %.2f TB/s  (%s)\n", INFLIGHT, STORE_KB, wgs, ms,
           rd / ms / 1e9, wr / ms / 1e9, (rd + wr) / ms / 1e9, hipGetErrorString(hipGetLastError()));
}

int main() {
    {
        unsigned *dx, *dout;
        std::vector<unsigned> hx(2048), ho(512);
        for (int i = 0; i < 2048; ++i) hx[i] = 0x1000u + i;
        hipMalloc(&dx, 8192);
        hipMalloc(&dout, 2048);
        hipMemcpy(dx, hx.data(), 8192, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(sem_probe, dim3(1), dim3(64), 0, 0, dx, 8192u, dout);
        hipMemcpy(ho.data(), dout, 2048, hipMemcpyDeviceToHost);
        printf("glds semantics (lane: 4 dwords; odd lanes out of range):\n");
        for (int l = 0; l < 8; ++l) printf("  lane %d: %08x %08x %08x %08x\n", l, ho[l * 4], ho[l * 4 + 1], ho[l * 4 + 2], ho[l * 4 + 3]);
        printf("with soffset 1024 (expect 0x1100 + 4*lane ...):\n");
        for (int l = 0; l < 4; ++l) printf("  lane %d: %08x %08x %08x %08x\n", l, ho[256 + l * 4], ho[256 + l * 4 + 1], ho[256 + l * 4 + 2], ho[256 + l * 4 + 3]);
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int q = 0; q < 4; ++q) {
                const unsigned want = (l & 1) ? 0u : 0x1000u + l * 4 + q;
                bad += ho[l * 4 + q] != want;
                bad += ho[256 + l * 4 + q] != 0x1100u + l * 4 + q;
            }
        printf("OOB->zero and soffset semantics: %s (%d mismatches)\n", bad ? "UNEXPECTED" : "as assumed", bad);
    }
    const size_t total = (size_t)3 << 30;
    unsigned char *x, *y;
    unsigned* sink;
    hipMalloc(&x, total);
    hipMalloc(&y, total / 48 * 64 + (1 << 20));
    hipMalloc(&sink, 64);
    hipMemset(x, 1, total);
    run_stream<1, 0>(x, y, sink, total, 256);
    run_stream<2, 0>(x, y, sink, total, 256);
    run_stream<2, 0>(x, y, sink, total, 512);
    run_stream<2, 16>(x, y, sink, total, 256);
    run_stream<2, 64>(x, y, sink, total, 256);
    run_stream<1, 64>(x, y, sink, total, 256);
    return 0;
}
