// Does the 256 MB memory-side cache (MALL / Infinity Cache) keep what a kernel just WROTE, so that the next kernel gains by reading the tensor
// in the opposite order (most recently written part first)?  Kernel W streams a buffer of `mb` megabytes out front to back; kernel R then
// streams it in (sum) front to back or back to front.  Sizes of the trunk's tensors at 192 frames: 113 MB (layer4), 226 / 453 MB, 906 MB (layer1 / stem).
//   hipcc --offload-arch=gfx950 -O3 -o mall_probe profiles/probe/mall_probe.hip && ./mall_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void wr(u32x4* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = u32x4{(unsigned)i, 1u, 2u, 3u};
}
template <bool REV>
__global__ __launch_bounds__(256) void rd(const u32x4* p, size_t n, unsigned* out) {
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc ^= p[REV ? n - 1 - i : i];
    if (acc[0] == 0x12345u) out[threadIdx.x] = acc[1];
}
// read A (old), then write B: bn_apply's pattern (read z, write y) followed by a consumer reading y
template <bool REV>
__global__ __launch_bounds__(256) void cp(const u32x4* a, u32x4* b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t j = REV ? n - 1 - i : i;
        b[j] = a[j] ^ u32x4{1u, 1u, 1u, 1u};
    }
}

int main() {
    const size_t cap = 1024ull << 20;
    u32x4 *a, *b;
    unsigned* out;
    CK(hipMalloc(&a, cap)); CK(hipMalloc(&b, cap)); CK(hipMalloc(&out, 4096));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * 8;
    const int sizes[5] = {113, 226, 453, 906, 1024};
    printf("MB      read fwd us (GB/s)   read rev us (GB/s) | copy fwd then copy fwd   copy fwd then copy rev (us of the 2nd copy, GB/s r+w)\n");
    for (int si = 0; si < 5; ++si) {
        const size_t bytes = (size_t)sizes[si] << 20, n = bytes / 16;
        float t[4] = {0, 0, 0, 0};
        for (int rep = 0; rep < 3; ++rep) {
            for (int mode = 0; mode < 2; ++mode) {
                hipLaunchKernelGGL(wr, dim3(grid), dim3(256), 0, 0, a, n);
                CK(hipEventRecord(e0, 0));
                if (mode == 0) hipLaunchKernelGGL(rd<false>, dim3(grid), dim3(256), 0, 0, a, n, out);
                else hipLaunchKernelGGL(rd<true>, dim3(grid), dim3(256), 0, 0, a, n, out);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&t[mode], e0, e1));
            }
            for (int mode = 0; mode < 2; ++mode) {
                hipLaunchKernelGGL(wr, dim3(grid), dim3(256), 0, 0, b, n);   // (b is the older tensor)
                hipLaunchKernelGGL(cp<false>, dim3(grid), dim3(256), 0, 0, b, a, n);   // produce a front to back
                CK(hipEventRecord(e0, 0));
                if (mode == 0) hipLaunchKernelGGL(cp<false>, dim3(grid), dim3(256), 0, 0, a, b, n);
                else hipLaunchKernelGGL(cp<true>, dim3(grid), dim3(256), 0, 0, a, b, n);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&t[2 + mode], e0, e1));
            }
        }
        printf("%4d   %8.1f (%5.0f)     %8.1f (%5.0f)   |   %8.1f (%5.0f)          %8.1f (%5.0f)\n", sizes[si], t[0] * 1e3, bytes / t[0] * 1e-6, t[1] * 1e3,
               bytes / t[1] * 1e-6, t[2] * 1e3, 2 * bytes / t[2] * 1e-6, t[3] * 1e3, 2 * bytes / t[3] * 1e-6);
    }
    return 0;
}
