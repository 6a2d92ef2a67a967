// Shared device helpers for the lp_hip kernels (gfx950 / CDNA4, wave64).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lp_hip.h"

namespace lp {

constexpr int kWave = 64;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// ---- bf16 <-> f32 (round-to-nearest-even, NaN preserved) -------------------------------------------
__device__ __forceinline__ float bf16_to_f32(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }

// gfx950 converts in hardware (v_cvt_pk_bf16_f32, two values per instruction); the arithmetic form is the same rounding and
// is what the host-side logic build (tests/hipemu) compiles
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ unsigned short f32_to_bf16(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
#else
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);  // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) { return (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16); }
#endif

// 8 floats -> 8 bf16 (16 B)
// GELU (exact form, x Phi(x): HF ViT's "gelu") and its derivative - vit.hip's stand-alone passes and the fused store passes of conv_pipe.h
// (kEkGeluFwd / kEkGeluBwd), eight values at a time in PAIRS: the arithmetic is v_pk_fma_f32 / v_pk_mul_f32 (two fp32 lanes per VALU slot), the
// reciprocal and the exponentials are the only scalar-rate instructions.  Instead of the device library's branching erff - in a store pass
// GELU's instructions are not hidden behind memory as they are in a streaming kernel (round 6, profiles/r06j/k/l_vit_step_ab.txt):
//   forward   Phi(-|x|) = erfc(|x| / sqrt 2) / 2 = t exp(-z^2 + P9(t)) / 2, t = 1 / (1 + z / 2): the Chebyshev fit of Numerical Recipes' erfcc,
//             FRACTIONAL error < 1.2e-7 everywhere, so the negative tail keeps its relative accuracy (0.5 x (1 + erf) cancels there): against
//             fp64 on bf16 inputs the rounded results differ in no element of 4 M (torch's fp32 F.gelu: 0.4 %)
//   backward  Phi(-|x|) = P5(t) E / 2, t = 1 / (1 + 0.3275911 z), E = exp(-x^2 / 2) (Abramowitz & Stegun 7.1.26, |error| < 1.5e-7 ABSOLUTE -
//             enough for Phi + x phi, which is O(1) wherever it matters): ONE exponential serves both terms; 0.014 % of the bf16 results
//             differ from fp64's by one place.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float fast_rcp(float d) {   // v_rcp_f32 (1 ulp)
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(d);
#else
    return 1.0f / d;
#endif
}
__device__ __forceinline__ f32x2_t fma2(f32x2_t a, f32x2_t b, f32x2_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_elementwise_fma(a, b, c);
#else
    return f32x2_t{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)};
#endif
}
__device__ __forceinline__ f32x2_t fma2(f32x2_t a, f32x2_t b, float c) { return fma2(a, b, f32x2_t{c, c}); }
__device__ __forceinline__ f32x2_t gelu_f2(f32x2_t x) {
    const f32x2_t z = f32x2_t{fabsf(x.x), fabsf(x.y)} * 0.70710678118654752f;
    const f32x2_t d = fma2(z, f32x2_t{0.5f, 0.5f}, 1.f);
    const f32x2_t t = {fast_rcp(d.x), fast_rcp(d.y)};
    f32x2_t p = fma2(t, f32x2_t{0.17087277f, 0.17087277f}, -0.82215223f);
    p = fma2(p, t, 1.48851587f);
    p = fma2(p, t, -1.13520398f);
    p = fma2(p, t, 0.27886807f);
    p = fma2(p, t, -0.18628806f);
    p = fma2(p, t, 0.09678418f);
    p = fma2(p, t, 0.37409196f);
    p = fma2(p, t, 1.00002368f);
    p = fma2(p, t, -1.26551223f);
    const f32x2_t a = fma2(-z, z, p);
    const f32x2_t tail = t * 0.5f * f32x2_t{__expf(a.x), __expf(a.y)};   // Phi(-|x|)
    const f32x2_t up = 1.f - tail;
    return x * f32x2_t{x.x < 0.f ? tail.x : up.x, x.y < 0.f ? tail.y : up.y};
}
__device__ __forceinline__ f32x2_t gelu_df2(f32x2_t x) {
    const f32x2_t z = f32x2_t{fabsf(x.x), fabsf(x.y)} * (0.3275911f * 0.70710678118654752f);
    const f32x2_t d = z + 1.f;
    const f32x2_t t = {fast_rcp(d.x), fast_rcp(d.y)};
    const f32x2_t h = x * x * -0.5f;
    const f32x2_t e = {__expf(h.x), __expf(h.y)};                            // exp(-x^2 / 2)
    f32x2_t p = fma2(t, f32x2_t{1.061405429f, 1.061405429f}, -1.453152027f);
    p = fma2(p, t, 1.421413741f);
    p = fma2(p, t, -0.284496736f);
    p = fma2(p, t, 0.254829592f);
    const f32x2_t tail = p * t * 0.5f * e;                                   // Phi(-|x|)
    const f32x2_t up = 1.f - tail;
    const f32x2_t phi = {x.x < 0.f ? tail.x : up.x, x.y < 0.f ? tail.y : up.y};
    return fma2(x * 0.3989422804014327f, e, phi);
}
// eight values at a time (every caller's unit: one 16-B piece of bf16)
__device__ __forceinline__ void gelu8(float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const f32x2_t g = gelu_f2(f32x2_t{v[i], v[i + 1]});
        v[i] = g.x, v[i + 1] = g.y;
    }
}
__device__ __forceinline__ void gelu8_bwd(float (&d)[8], const float (&u)[8]) {   // d *= GELU'(u)
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const f32x2_t g = f32x2_t{d[i], d[i + 1]} * gelu_df2(f32x2_t{u[i], u[i + 1]});
        d[i] = g.x, d[i + 1] = g.y;
    }
}

__device__ __forceinline__ u16x8 pack_bf16x8(const float (&f)[8]) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
    const u32x4_t p = {pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7])};
    return __builtin_bit_cast(u16x8, p);
}

// ReLU gate from a stored bf16 activation: positive (and not -0 / +0)  <=>  its bits read as int16 are > 0
__device__ __forceinline__ bool bf16_positive(unsigned short y) { return (short)y > 0; }

// ---- raw buffer addressing: 32-bit byte offsets against a (base, size) descriptor; loads past the end return zeros.
// The im2col gathers use it for zero padding: an invalid tap is simply given the offset 0xffffffff, so the loop needs no
// select on the loaded data and no 64-bit address arithmetic.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t buf_rsrc;
__device__ __forceinline__ buf_rsrc make_buf_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
// 16 B at byte offset voff + soff (soff wave-uniform, not range-checked); zeros if voff is out of range
__device__ __forceinline__ u16x8 buf_load16(buf_rsrc r, unsigned voff, unsigned soff) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return __builtin_bit_cast(u16x8, v);
}
// same, marked non-temporal (streamed once: do not keep the lines in L2 / MALL)
__device__ __forceinline__ u16x8 buf_load16_nt(buf_rsrc r, unsigned voff, unsigned soff) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 2);
    return __builtin_bit_cast(u16x8, v);
}
// 16 B per lane straight into LDS (`buffer_load_dwordx4 ... lds`): lane i of the wave writes lds_wave_base + 16 i (the LDS side is
// lane-linear from a wave-uniform base; only the SOURCE offset is per lane); an out-of-range voff stores zeros (measured:
// profiles/probe/glds_probe.hip).  Asynchronous: counted on vmcnt like any load, ordered for a ds_read only by that wait + a barrier.
__device__ __forceinline__ void buf_load16_lds(buf_rsrc r, void* lds_wave_base, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}
#else
struct buf_rsrc {
    const char* p;
    unsigned bytes;
};
__device__ __forceinline__ buf_rsrc make_buf_rsrc(const void* p, unsigned bytes) { return buf_rsrc{(const char*)p, bytes}; }
__device__ __forceinline__ u16x8 buf_load16(buf_rsrc r, unsigned voff, unsigned soff) {
    u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (voff < r.bytes && (unsigned long long)voff + 16 <= r.bytes) v = *reinterpret_cast<const u16x8*>(r.p + voff + soff);
    return v;
}
__device__ __forceinline__ u16x8 buf_load16_nt(buf_rsrc r, unsigned voff, unsigned soff) { return buf_load16(r, voff, soff); }
#if !defined(__HIP__)  // CPU logic build: the same data movement, synchronously
__device__ __forceinline__ void buf_load16_lds(buf_rsrc r, void* lds_wave_base, unsigned voff, unsigned soff) {
    *reinterpret_cast<u16x8*>(static_cast<char*>(lds_wave_base) + (threadIdx.x & 63) * 16) = buf_load16(r, voff, soff);
}
#else  // host pass of hipcc: declaration only
__device__ __forceinline__ void buf_load16_lds(buf_rsrc, void*, unsigned, unsigned) {}
#endif
#endif

// ---- LDS transpose read (ds_read_b64_tr_b16): within every group of 16 lanes the 16 addresses name a [4 rows][16 columns]
// block of 16-bit elements - lanes 4e .. 4e+3 supply row e as four runs of 4 contiguous elements - and lane c receives column
// c, i.e. {row0[c], row1[c], row2[c], row3[c]}.  It turns a [k][channel] LDS image into MFMA operand fragments (8 consecutive
// k per lane = two reads) without any register transposes.  (Lane mapping measured on gfx950: profiles/probe/tr_probe.hip.)
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ s16x4_t lds_read_tr16(const unsigned short* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
}
// The same read issued WITHOUT the compiler's knowledge of its memory access (inline asm): hipcc orders the builtin form behind every
// pending direct-to-LDS load with `s_waitcnt vmcnt(0)`, which drains a multi-stage operand ring once per k-slice (conv_wgrad_pipe_kernel
// measured 67 % of its wave cycles parked on it).  The caller owns both orderings: the LDS-DMA data is ready by its counted vmcnt wait +
// barrier, and the result registers may only be used after LP_WAIT_LGKM_TOUCH (the asm form is invisible to the compiler's lgkmcnt count).
__device__ __forceinline__ s16x4_t lds_read_tr16_async(const unsigned short* p) {
    s16x4_t v;
    const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned short*)p;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(a) : "memory");
    return v;
}
// ... with a compile-time byte offset in the instruction's offset field (no address arithmetic per read)
template <int OFF>
__device__ __forceinline__ s16x4_t lds_read_tr16_async_off(const unsigned char* p) {
    static_assert(OFF >= 0 && OFF < 65536, "16-bit offset field");
    s16x4_t v;
    const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)p;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF) : "memory");
    return v;
}
#elif defined(__HIP__)  // host pass of hipcc: declaration only, never executed
__device__ __forceinline__ s16x4_t lds_read_tr16(const unsigned short*) { return s16x4_t{0, 0, 0, 0}; }
__device__ __forceinline__ s16x4_t lds_read_tr16_async(const unsigned short*) { return s16x4_t{0, 0, 0, 0}; }
template <int OFF>
__device__ __forceinline__ s16x4_t lds_read_tr16_async_off(const unsigned char*) { return s16x4_t{0, 0, 0, 0}; }
#else  // CPU logic build (tests/hipemu): the same lane exchange spelled with shuffles
__device__ __forceinline__ s16x4_t lds_read_tr16(const unsigned short* p) {
    const int lane = threadIdx.x & 63, base = lane & ~15, c = lane & 15;
    s16x4_t r;
    for (int e = 0; e < 4; ++e) {
        const unsigned short* q = __shfl(p, base + 4 * e + (c >> 2), 64);
        r[e] = (short)q[c & 3];
    }
    return r;
}
__device__ __forceinline__ s16x4_t lds_read_tr16_async(const unsigned short* p) { return lds_read_tr16(p); }
template <int OFF>
__device__ __forceinline__ s16x4_t lds_read_tr16_async_off(const unsigned char* p) { return lds_read_tr16(reinterpret_cast<const unsigned short*>(p + OFF)); }
#endif

// Hide a value's provenance from the optimiser (keeps it from hoisting per-element address arithmetic out of a loop into
// dozens of long-lived registers)
#if defined(__HIP_DEVICE_COMPILE__)
#define LP_OPAQUE(x) asm volatile("" : "+v"(x))
#else
#define LP_OPAQUE(x) asm volatile("" : "+r"(x))
#endif

// ---- read-only tables indexed by WAVE-UNIFORM values (tap tables walked by a row counter, per-launch parameter rows) ----------------
// hipcc only emits scalar loads (s_load: one fetch per wave into SGPRs, far ahead of use, no vector-memory latency in the loop) for memory
// it can prove nobody writes during the kernel.  Pointers that arrive inside a by-value struct carry no such promise, so a uniform
// `taps[j * 9 + t]` becomes a per-lane global_load_dwordx4 of the same address in all 64 lanes followed by s_waitcnt vmcnt(0) - round 6 found
// decode_bwd_kernel waiting three L1 round trips per row group that way.  A pointer in the CONSTANT address space (4
// on amdgcn: the same memory as global, declared immutable for the kernel's lifetime) gets the scalar loads.
#ifndef LP_UNIFORM_LOADS
#define LP_UNIFORM_LOADS 1   // (A/B builds: 0 = plain pointers, the per-lane loads of rounds 1 - 5)
#endif
#if defined(__HIP_DEVICE_COMPILE__) && LP_UNIFORM_LOADS
template <typename T>
using uniform_ptr = const __attribute__((address_space(4))) T*;
template <typename T>
__device__ __forceinline__ uniform_ptr<T> as_uniform(const T* p) {
    return (uniform_ptr<T>)(unsigned long long)p;
}
#else
template <typename T>
using uniform_ptr = const T*;
template <typename T>
__host__ __device__ inline uniform_ptr<T> as_uniform(const T* p) {   // (hipcc's host pass and the CPU emulator build: a plain pointer)
    return p;
}
#endif

// ---- wave / block reductions ------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}

// Sum over a workgroup of NW waves (blockDim.x == 64*NW); every thread gets the total.  `scratch` holds
// >= NW floats of LDS; the trailing barrier makes it immediately reusable.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) t += scratch[i];
    __syncthreads();
    return t;
}

template <int NW>
__device__ __forceinline__ float block_max(float v, float* scratch) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    float t = scratch[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) t = fmaxf(t, scratch[i]);
    __syncthreads();
    return t;
}

// ---- bit-reproducible cross-workgroup sums (round 4): fixed point + INTEGER atomics ------------------------------------------------
// A BatchNorm statistic is a sum over rows spread across hundreds of workgroups.  Rounds 2 - 3 added the workgroups' fp32 partial sums
// into the total with fp32 atomics: the order of arrival decided the last bits, and a training step did not repeat.  Integer addition
// commutes, so each fp32 partial t is split - a pure function of t - into two 64-bit integers, hi = rint(t 2^12) and lo = rint((t - hi
// 2^-12) 2^60), and both are added with 64-bit integer atomics into an lp_fxsum {hi, lo} per value: whatever the order, the same bits.
// The consumer reads value = hi 2^-12 + lo 2^-60.  Range |sum| < 2^50 (1e15; a partial beyond it saturates), resolution 2^-61 (4e-19) per
// partial: a partial of 1e-10 or more is represented EXACTLY (its 24 bits lie above 2^-60), smaller ones to 4e-19 absolute - far inside
// fp32 atomics' own rounding.  Costs what the atomics cost (two 8-byte ones per value instead of a 4-byte one + the d beta / d gamma one),
// no workspace, no extra launch.  (The first form of round 4 - one row of partial sums per workgroup + an ordered reduction launch - was
// bit-reproducible too and 1.4 % slower per step: ~110 more small launches on the critical path, profiles/r04i_bench_*.)
// A NON-FINITE partial (a diverged run) must not turn into finite garbage - fmin / fmax drop a NaN, and the old fp32 atomics propagated it:
// it POISONS the sum instead: `lo` is raised to kFxPoison = 5 * 2^60 with an atomic max (idempotent, so any number of poisoned partials of
// this process leaves it there; the finite partials that still arrive move it by < 2^59), and fx_value reads |lo| >= 2^59 as NaN -
// BatchNorm's moments, running statistics and gradients then carry the NaN on, as the reference's do.  Across ranks SyncBatchNorm SUMS the
// words (all-reduce, or all-gather + add), modulo 2^64: k poisoned ranks leave k * 5 * 2^60 = (5 k mod 16) * 2^60, and 5 k is not a multiple
// of 16 for k = 1 .. 15, so the summed word still lies outside (-2^59, 2^59) for every world size up to 15 ranks - one node has 8 - whatever
// the number of ranks that diverged (round 5's 2^62 wrapped to 0 at exactly 4 and 8 poisoned ranks: tests/test_segmented_bn.py).
constexpr long long kFxPoison = 5LL << 60;
constexpr long long kFxPoisonSeen = 1LL << 59;
__device__ __forceinline__ void fx_add(lp_fxsum* p, float t) {
    double td = (double)t;
    if (!(fabs(td) <= 0x1p127 * 2.0)) {   // NaN or +-inf
        atomicMax(&p->lo, kFxPoison);
        return;
    }
    td = fmin(fmax(td, -0x1p49), 0x1p49);                         // (saturate: hi stays inside 2^61)
    const double h = rint(td * 0x1p12);
    const long long hi = (long long)h;
    const long long lo = (long long)rint((td - h * 0x1p-12) * 0x1p60);   // |remainder| <= 2^-13: |lo| <= 2^47
    atomicAdd(reinterpret_cast<unsigned long long*>(&p->hi), (unsigned long long)hi);
    atomicAdd(reinterpret_cast<unsigned long long*>(&p->lo), (unsigned long long)lo);
}
__device__ __forceinline__ float fx_value(const lp_fxsum* p) {
    const long long lo = p->lo;
    if (lo >= kFxPoisonSeen || lo <= -kFxPoisonSeen) return __int_as_float(0x7fc00000);   // poisoned by a non-finite partial
    return (float)((double)p->hi * 0x1p-12 + (double)lo * 0x1p-60);
}

// ---- host-side launch epilogue ------------------------------------------------------------------
// ---- the library's A/B switches (LP_CONV_PIPE, LP_CONV_HALO, ... - what each one selects is documented where it is used).  Read from the
// environment ONCE, when the library is loaded (api.hip), into this table; the launch paths only read the table, so no entry point calls
// getenv and the behaviour of a call does not depend on what the caller's environment holds at that moment.  lp_config_reload_env()
// re-reads it (tests and A/B scripts that flip a switch inside one process).
struct LpSwitches {
    int conv_pipe, conv_halo, conv_res2d, infer_pipe, gemm_pipe, wgrad_pipe, stem_2d, stem_wgrad_nb, pool_v2, conv_max_wgs, bn_bwd_wgs_per_cu;
};
const LpSwitches& lp_switches();

inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? LP_OK : (int)e;
}

}  // namespace lp

#define LP_REQUIRE(cond)                    \
    do {                                    \
        if (!(cond)) return LP_ERR_ARGUMENT; \
    } while (0)
