// Probe of ds_read_b64_tr_b16 lane semantics on gfx950: LDS holds element index = its own 16-bit position; lane l reads at
// byte address l*8 (4 consecutive b16) with the transpose read; print what each lane receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void probe(unsigned short* out, int stride_bytes) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    // lane l -> row (l % 16), 8-byte column group (l / 16): address = row * stride + (l/16)*8
    const int l = threadIdx.x;
    const unsigned addr = (unsigned)(size_t)lds + (l % 16) * stride_bytes + (l / 16) * 8;
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = (unsigned short)v[e];
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    for (int stride : {8, 32, 128}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, stride);
        unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("stride %d bytes (row r at element r*%d):\n", stride, stride / 2);
        for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int e = 0; e < 4; ++e) printf(" %4d", h[l * 4 + e]); printf("\n"); }
    }
    return 0;
}
