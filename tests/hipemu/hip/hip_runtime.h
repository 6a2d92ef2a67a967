// hipemu - a minimal HOST-CPU stand-in for <hip/hip_runtime.h>, TEST INFRASTRUCTURE ONLY.
//
// Purpose: compile the product's unmodified HIP kernel sources (lightning-pose_amd/csrc/*.hip) with the
// host clang++ (`-x c++ -I tests/hipemu`) and execute them on CPU threads, so kernel *logic* (indexing,
// LDS staging, wave reductions, MFMA fragment maps, barriers) can be checked against the oracle in the
// GPU-less build container (`pytest -m "not gpu"`).  It is never linked into the product library, never
// loaded by the product package, and says nothing about performance.
//
// Execution model: one OS thread runs one workgroup at a time; each work-item is a fiber (hand-rolled x86-64 context switch).
// __syncthreads() and the wave-collective operations (shuffles, ballot, MFMA) yield to a cooperative
// scheduler that releases a barrier when every live fiber of the workgroup / 64-lane wave has arrived.
// A collective executed by only part of a wave deadlocks the scheduler and aborts with a message -
// which doubles as a lint for divergent collectives.
//
// MFMA emulation follows the gfx950 fragment maps in /opt/skills/guides/cdna_hip_programming.md section 3:
//   32x32x16 bf16 : A lane l holds A[l&31][8*(l>>5)+j], B lane l holds B[8*(l>>5)+j][l&31], j<8;
//                   D reg r -> row (r&3)+8*(r>>2)+4*(l>>5), col l&31
//   16x16x32 bf16 : A[l&15][8*(l>>4)+j], B[8*(l>>4)+j][l&15]; D reg r -> row 4*(l>>4)+r, col l&15
//   32x32x2  f32  : A[l&31][l>>5], B[l>>5][l&31];  16x16x4 f32: A[l&15][l>>4], B[l>>4][l&15]
#pragma once


#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(::hipemu::tls().dyn_smem);

typedef int hipError_t;
static const hipError_t hipSuccess = 0;
static const hipError_t hipErrorInvalidValue = 1;
typedef void* hipStream_t;
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipPeekAtLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return hipSuccess; }
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 0 };
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

struct alignas(8) float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(8) uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
inline float2 make_float2(float x, float y) { return {x, y}; }
inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return {x, y, z, w}; }
inline uint2 make_uint2(unsigned x, unsigned y) { return {x, y}; }
inline int2 make_int2(int x, int y) { return {x, y}; }

namespace hipemu {

enum FiberState { RUNNABLE = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };

struct Fiber {
    void* sp;  // saved stack pointer while the fiber is switched out
    int state;
    void* stack;
};

// Minimal x86-64 context switch (callee-saved registers + stack pointer).  glibc's swapcontext issues a
// rt_sigprocmask system call per switch, which dominated the emulator's run time.
__attribute__((naked, noinline)) static void switch_ctx(void** /*save_sp*/, void* /*load_sp*/) {
    asm volatile(
        "pushq %rbp\n\tpushq %rbx\n\tpushq %r12\n\tpushq %r13\n\tpushq %r14\n\tpushq %r15\n\t"
        "movq %rsp, (%rdi)\n\t"
        "movq %rsi, %rsp\n\t"
        "popq %r15\n\tpopq %r14\n\tpopq %r13\n\tpopq %r12\n\tpopq %rbx\n\tpopq %rbp\n\t"
        "ret\n\t");
}

struct Tls {
    dim3 threadIdx, blockIdx, blockDim, gridDim;
    unsigned char* dyn_smem = nullptr;
    // scheduler
    void* sched_sp = nullptr;
    std::vector<Fiber> fibers;
    int cur = -1;
    int nthreads = 0;
    const std::function<void()>* body = nullptr;
    // wave exchange: per wave, double-buffered 64 lanes x 128 bytes
    std::vector<unsigned char> xbuf, smem_buf;
    std::vector<unsigned> xcnt;  // per-fiber collective-op counter
};

inline Tls& tls() {
    static thread_local Tls t;
    return t;
}

static const size_t kStackBytes = 256 * 1024;
static const size_t kSlotBytes = 128;

inline void yield_with(int state) {
    Tls& t = tls();
    Fiber& f = t.fibers[t.cur];
    f.state = state;
    switch_ctx(&f.sp, t.sched_sp);
}

inline void set_ids(Tls& t, int tid) {
    t.threadIdx.x = tid % t.blockDim.x;
    t.threadIdx.y = (tid / t.blockDim.x) % t.blockDim.y;
    t.threadIdx.z = tid / (t.blockDim.x * t.blockDim.y);
}

inline void fiber_entry() {
    Tls& t = tls();
    (*t.body)();
    Tls& t2 = tls();
    t2.fibers[t2.cur].state = DONE;
    switch_ctx(&t2.fibers[t2.cur].sp, t2.sched_sp);
    std::abort();  // a finished fiber is never resumed
}

inline void run_block(const std::function<void()>& body, dim3 grid, dim3 block, dim3 bidx, size_t shmem) {
    Tls& t = tls();
    t.gridDim = grid;
    t.blockDim = block;
    t.blockIdx = bidx;
    t.body = &body;
    const int n = (int)(block.x * block.y * block.z);
    t.nthreads = n;
    if (t.smem_buf.size() < shmem + 64) t.smem_buf.resize(shmem + 64);
    t.dyn_smem = reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(t.smem_buf.data()) + 63) & ~uintptr_t(63));
    if ((int)t.fibers.size() < n) {
        size_t old = t.fibers.size();
        t.fibers.resize(n);
        for (size_t i = old; i < (size_t)n; ++i) t.fibers[i].stack = std::malloc(kStackBytes);
    }
    const int nwaves = (n + 63) / 64;
    if (t.xbuf.size() < (size_t)nwaves * 2 * 64 * kSlotBytes) t.xbuf.resize((size_t)nwaves * 2 * 64 * kSlotBytes);
    t.xcnt.assign(n, 0);
    for (int i = 0; i < n; ++i) {
        Fiber& f = t.fibers[i];
        // initial frame: six zeroed callee-saved registers, then fiber_entry as the "return address" of switch_ctx
        uintptr_t top = (reinterpret_cast<uintptr_t>(f.stack) + kStackBytes) & ~uintptr_t(15);
        void** sp = reinterpret_cast<void**>(top);
        *--sp = nullptr;                                   // fake return address of fiber_entry (keeps 16-B call alignment)
        *--sp = reinterpret_cast<void*>(&fiber_entry);
        for (int r = 0; r < 6; ++r) *--sp = nullptr;
        f.sp = sp;
        f.state = RUNNABLE;
    }
    for (;;) {
        bool progress = false;
        int alive = 0;
        for (int i = 0; i < n; ++i) {
            if (t.fibers[i].state == RUNNABLE) {
                t.cur = i;
                set_ids(t, i);
                switch_ctx(&t.sched_sp, t.fibers[i].sp);
                progress = true;
            }
            if (t.fibers[i].state != DONE) ++alive;
        }
        if (alive == 0) break;
        bool released = false;
        // block barrier: every live fiber waits at the block barrier
        int wb = 0;
        for (int i = 0; i < n; ++i) wb += (t.fibers[i].state == WAIT_BLOCK);
        if (wb == alive) {
            for (int i = 0; i < n; ++i)
                if (t.fibers[i].state == WAIT_BLOCK) t.fibers[i].state = RUNNABLE;
            released = true;
        }
        // wave barriers
        for (int w = 0; w < nwaves; ++w) {
            int lo = w * 64, hi = std::min(n, lo + 64), live = 0, ww = 0;
            for (int i = lo; i < hi; ++i) {
                live += (t.fibers[i].state != DONE);
                ww += (t.fibers[i].state == WAIT_WAVE);
            }
            if (live > 0 && ww == live) {
                for (int i = lo; i < hi; ++i)
                    if (t.fibers[i].state == WAIT_WAVE) t.fibers[i].state = RUNNABLE;
                released = true;
            }
        }
        if (!released && !progress) {
            std::fprintf(stderr, "hipemu: deadlock in block (%u,%u,%u): divergent barrier or wave collective\n",
                         bidx.x, bidx.y, bidx.z);
            std::abort();
        }
    }
}

inline int& max_host_threads() {
    static int n = [] {
        const char* e = std::getenv("HIPEMU_THREADS");
        int v = e ? std::atoi(e) : (int)std::thread::hardware_concurrency();
        return std::max(1, std::min(v, 16));
    }();
    return n;
}

// Persistent worker pool: worker threads (and their thread-local fiber stacks) live across launches; spawning
// threads per launch made every launch re-map and re-fault 64 MB of fiber stacks per thread.
struct Pool {
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::vector<std::thread> workers;
    const std::function<void()>* job = nullptr;
    unsigned long generation = 0;
    int active = 0;
    bool stop = false;

    explicit Pool(int n) {
        for (int i = 0; i < n; ++i) workers.emplace_back([this] { loop(); });
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv_work.notify_all();
        for (auto& w : workers) w.join();
    }
    void loop() {
        unsigned long seen = 0;
        for (;;) {
            const std::function<void()>* j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || generation != seen; });
                if (stop) return;
                seen = generation;
                j = job;
            }
            (*j)();
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--active == 0) cv_done.notify_all();
            }
        }
    }
    void run(const std::function<void()>& j) {
        std::unique_lock<std::mutex> lk(mu);
        job = &j;
        active = (int)workers.size();
        ++generation;
        cv_work.notify_all();
        cv_done.wait(lk, [&] { return active == 0; });
    }
};

inline Pool& pool() {
    static Pool p(max_host_threads());
    return p;
}

inline void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
    const size_t nblocks = (size_t)grid.x * grid.y * grid.z;
    if (nblocks == 0) return;
    std::atomic<size_t> next{0};
    std::function<void()> worker = [&]() {
        for (;;) {
            size_t b = next.fetch_add(1);
            if (b >= nblocks) break;
            dim3 bidx((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((size_t)grid.x * grid.y)));
            run_block(body, grid, block, bidx, shmem);
        }
    };
    if (nblocks == 1 || max_host_threads() <= 1) {
        worker();
    } else {
        static std::mutex launch_mu;  // one launch at a time through the pool
        std::lock_guard<std::mutex> lk(launch_mu);
        pool().run(worker);
    }
}

// ---- wave collectives -----------------------------------------------------------------------------
inline int flat_tid() {
    Tls& t = tls();
    return t.cur;
}

// deposit `bytes` of `src` in this lane's slot, wave-sync, return pointer to the wave's slot array of this op
inline unsigned char* exchange(const void* src, size_t bytes) {
    Tls& t = tls();
    const int tid = t.cur, wave = tid >> 6, lane = tid & 63;
    const unsigned op = t.xcnt[tid]++;
    unsigned char* base = t.xbuf.data() + ((size_t)(wave * 2 + (op & 1)) * 64) * kSlotBytes;
    std::memcpy(base + lane * kSlotBytes, src, bytes);
    yield_with(WAIT_WAVE);
    return tls().xbuf.data() + ((size_t)(wave * 2 + (op & 1)) * 64) * kSlotBytes;
}

template <typename T>
inline T shfl_from(T v, int src_lane) {
    static_assert(sizeof(T) <= kSlotBytes, "slot too small");
    unsigned char* base = exchange(&v, sizeof(T));
    T r;
    std::memcpy(&r, base + (size_t)(src_lane & 63) * kSlotBytes, sizeof(T));
    return r;
}

}  // namespace hipemu

#define threadIdx (::hipemu::tls().threadIdx)
#define blockIdx (::hipemu::tls().blockIdx)
#define blockDim (::hipemu::tls().blockDim)
#define gridDim (::hipemu::tls().gridDim)
static const int warpSize = 64;

inline void __syncthreads() { ::hipemu::yield_with(::hipemu::WAIT_BLOCK); }
inline void __builtin_amdgcn_s_barrier_emu() { ::hipemu::yield_with(::hipemu::WAIT_BLOCK); }
inline void __threadfence() {}
inline void __threadfence_block() {}

template <typename T> inline T __shfl(T v, int src, int width = 64) {
    int lane = ::hipemu::flat_tid() & 63;
    int base = lane & ~(width - 1);
    return ::hipemu::shfl_from(v, base + (src & (width - 1)));
}
template <typename T> inline T __shfl_xor(T v, int mask, int width = 64) {
    int lane = ::hipemu::flat_tid() & 63;
    (void)width;
    return ::hipemu::shfl_from(v, lane ^ mask);
}
template <typename T> inline T __shfl_down(T v, unsigned d, int width = 64) {
    int lane = ::hipemu::flat_tid() & 63;
    int src = lane + (int)d;
    if ((src & ~(width - 1)) != (lane & ~(width - 1)) || src > 63) src = lane;
    return ::hipemu::shfl_from(v, src);
}
template <typename T> inline T __shfl_up(T v, unsigned d, int width = 64) {
    int lane = ::hipemu::flat_tid() & 63;
    int src = lane - (int)d;
    if (src < 0 || (src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
    return ::hipemu::shfl_from(v, src);
}
inline unsigned long long __ballot(int pred) {
    int p = pred ? 1 : 0;
    unsigned char* base = ::hipemu::exchange(&p, sizeof(int));
    unsigned long long m = 0;
    auto& t = ::hipemu::tls();
    int wave = t.cur >> 6;
    int lanes = std::min(64, t.nthreads - wave * 64);
    for (int l = 0; l < lanes; ++l) {
        int q;
        std::memcpy(&q, base + (size_t)l * ::hipemu::kSlotBytes, sizeof(int));
        if (q) m |= (1ull << l);
    }
    return m;
}
inline int __all(int pred) {
    auto& t = ::hipemu::tls();
    int wave = t.cur >> 6;
    int lanes = std::min(64, t.nthreads - wave * 64);
    unsigned long long full = lanes == 64 ? ~0ull : ((1ull << lanes) - 1);
    return __ballot(pred) == full;
}
inline int __any(int pred) { return __ballot(pred) != 0; }

// ---- atomics ------------------------------------------------------------------------------------
inline float atomicAdd(float* p, float v) {
    auto* a = reinterpret_cast<std::atomic<float>*>(p);
    float old = a->load(std::memory_order_relaxed);
    while (!a->compare_exchange_weak(old, old + v)) {}
    return old;
}
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline int atomicMax(int* p, int v) {
    int old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
inline long long atomicMax(long long* p, long long v) {
    long long old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
    }
    return old;
}
inline unsigned atomicOr(unsigned* p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }

using std::max;
using std::min;
// ---- math ---------------------------------------------------------------------------------------
inline float __expf(float x) { return std::exp(x); }
inline float __fdividef(float a, float b) { return a / b; }
inline float __frcp_rn(float a) { return 1.0f / a; }
inline float rsqrtf(float a) { return 1.0f / std::sqrt(a); }
inline float __int_as_float(int i) { float f; std::memcpy(&f, &i, 4); return f; }
inline int __float_as_int(float f) { int i; std::memcpy(&i, &f, 4); return i; }
inline unsigned __float_as_uint(float f) { unsigned i; std::memcpy(&i, &f, 4); return i; }
inline float __uint_as_float(unsigned i) { float f; std::memcpy(&f, &i, 4); return f; }

// ---- MFMA ---------------------------------------------------------------------------------------
namespace hipemu {
typedef __attribute__((ext_vector_type(8))) __bf16 v8bf16;
typedef __attribute__((ext_vector_type(4))) float v4f;
typedef __attribute__((ext_vector_type(16))) float v16f;

inline float bf16_bits_to_f32(unsigned short b) {
    unsigned u = (unsigned)b << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

struct AB16 { unsigned short a[8]; unsigned short b[8]; };

inline const unsigned short* slot_u16(const unsigned char* base, int lane) {
    return reinterpret_cast<const unsigned short*>(base + (size_t)lane * kSlotBytes);
}

inline v16f mfma_32x32x16_bf16(v8bf16 a, v8bf16 b, v16f c, int, int, int) {
    AB16 ab;
    std::memcpy(ab.a, &a, 16);
    std::memcpy(ab.b, &b, 16);
    unsigned char* base = exchange(&ab, sizeof(ab));
    const int lane = flat_tid() & 63;
    const int col = lane & 31;
    float bcol[16];
    for (int k = 0; k < 16; ++k) bcol[k] = bf16_bits_to_f32(slot_u16(base, col + 32 * (k >> 3))[8 + (k & 7)]);
    v16f d = c;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const unsigned short* a0 = slot_u16(base, row);
        const unsigned short* a1 = slot_u16(base, row + 32);
        float acc = 0.f;
        for (int k = 0; k < 8; ++k) acc += bf16_bits_to_f32(a0[k]) * bcol[k];
        for (int k = 0; k < 8; ++k) acc += bf16_bits_to_f32(a1[k]) * bcol[8 + k];
        d[r] = c[r] + acc;
    }
    return d;
}

inline v4f mfma_16x16x32_bf16(v8bf16 a, v8bf16 b, v4f c, int, int, int) {
    AB16 ab;
    std::memcpy(ab.a, &a, 16);
    std::memcpy(ab.b, &b, 16);
    unsigned char* base = exchange(&ab, sizeof(ab));
    const int lane = flat_tid() & 63;
    const int col = lane & 15;
    v4f d = c;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (lane >> 4) + r;
        float acc = 0.f;
        for (int k = 0; k < 32; ++k)
            acc += bf16_bits_to_f32(slot_u16(base, row + 16 * (k >> 3))[k & 7]) * bf16_bits_to_f32(slot_u16(base, col + 16 * (k >> 3))[8 + (k & 7)]);
        d[r] = c[r] + acc;
    }
    return d;
}

struct ABf { float a, b; };

inline v16f mfma_32x32x2_f32(float a, float b, v16f c, int, int, int) {
    ABf ab{a, b};
    unsigned char* base = exchange(&ab, sizeof(ab));
    const int lane = flat_tid() & 63;
    const int col = lane & 31;
    v16f d = c;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            ABf ra, rb;
            std::memcpy(&ra, base + (size_t)(row + 32 * k) * kSlotBytes, sizeof(ABf));
            std::memcpy(&rb, base + (size_t)(col + 32 * k) * kSlotBytes, sizeof(ABf));
            acc = std::fma(ra.a, rb.b, acc);
        }
        d[r] = acc;
    }
    return d;
}

inline v4f mfma_16x16x4_f32(float a, float b, v4f c, int, int, int) {
    ABf ab{a, b};
    unsigned char* base = exchange(&ab, sizeof(ab));
    const int lane = flat_tid() & 63;
    const int col = lane & 15;
    v4f d = c;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (lane >> 4) + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            ABf ra, rb;
            std::memcpy(&ra, base + (size_t)(row + 16 * k) * kSlotBytes, sizeof(ABf));
            std::memcpy(&rb, base + (size_t)(col + 16 * k) * kSlotBytes, sizeof(ABf));
            acc = std::fma(ra.a, rb.b, acc);
        }
        d[r] = acc;
    }
    return d;
}
}  // namespace hipemu

#define __builtin_amdgcn_mfma_f32_32x32x16_bf16 ::hipemu::mfma_32x32x16_bf16
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16 ::hipemu::mfma_16x16x32_bf16
#define __builtin_amdgcn_mfma_f32_32x32x2f32 ::hipemu::mfma_32x32x2_f32
#define __builtin_amdgcn_mfma_f32_16x16x4f32 ::hipemu::mfma_16x16x4_f32
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
// lanes of a wave run in lock step on the device; here they are fibers, so code that exchanges data between the lanes of ONE wave
// through LDS marks the exchange points with a wave barrier (a pure scheduling fence on the device)
#define __builtin_amdgcn_wave_barrier() ::hipemu::yield_with(::hipemu::WAIT_WAVE)

// ---- launch -------------------------------------------------------------------------------------
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)                                   \
    do {                                                                                                \
        (void)(stream);                                                                                 \
        ::hipemu::launch(dim3(grid), dim3(block), (size_t)(shmem), [=]() { kernel(__VA_ARGS__); });     \
    } while (0)

// ---- function attributes (no-ops on the host) ---------------------------------------------------
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
