// tests/emu/hip_emu.hpp -- TEST INFRASTRUCTURE ONLY.
//
// A tiny single-threaded emulation of the HIP execution model, used to run the *same kernel
// sources* that ship in bzip3_amd/csrc on this GPU-less build container, so kernel logic can be
// checked against the oracle before GPU minutes are spent.  It is compiled only into
// tests/emu/libbz3_emu_TESTONLY.so by tests/emu/build_emu.py (-DBZ3_EMU).  The product library
// (bzip3_amd/lib/libbzip3.so, hipcc, gfx950) never sees this header and has no CPU path at all.
//
// Model: a kernel launch runs its blocks one after another; the threads of a block are fibers
// (hand-rolled x86-64 context switch) scheduled round-robin; __syncthreads() and the wave
// collectives (__shfl, __ballot, ...) are rendezvous points.  Wave size is 64.  Collectives must be
// called from wave-uniform control flow by all live lanes of the wave (the kernels are written
// that way).  Blocks never run concurrently, so inter-block spin protocols are not supported.
#pragma once
#include <sys/mman.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __constant__ static const

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }

typedef int hipError_t;
typedef void * hipStream_t;
struct emu_event { std::chrono::steady_clock::time_point t; };
typedef emu_event * hipEvent_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorUnknown = 999 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault, hipMemcpyHostToHost };
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0, hipEventDisableTiming = 2 };

namespace emu {

constexpr int kWave = 64;
constexpr size_t kStack = 256 * 1024;

struct Fiber {
    void * sp = nullptr;
    char * stack = nullptr;
    dim3 tid;
    bool done = true;
};

struct Block {
    dim3 bid, bdim, gdim;
    int nthreads = 0, alive = 0;
    unsigned bar_count = 0, bar_gen = 0;
    int nwaves = 0;
    int wave_alive[32];
    unsigned wbar_count[32], wbar_gen[32];
    uint64_t slots[32][kWave];
    uint64_t live_mask[32];
    char * dyn = nullptr;
    std::function<void()> body;
};

extern Block g_blk;
extern Fiber * g_cur;
extern void * g_sched_sp;
extern std::vector<Fiber> g_pool;

extern "C" void emu_swap(void ** save_sp, void * load_sp);
// Set under BZ3_EMU_SCHED (stress scheduling): the thread that completes a rendezvous yields once, so that all lanes of a
// wave leave it in the same scheduler sweep -- whole waves are then stalled or run, never a wave minus one lane (which
// no GPU can produce, and which would let one lane publish a flag on behalf of lanes that have not stored their data).
extern bool g_lockstep_release;

inline void yield() { emu_swap(&g_cur->sp, g_sched_sp); }

inline int cur_linear() { return (int)(g_cur->tid.x); }
inline int cur_wave() { return cur_linear() / kWave; }
inline int cur_lane() { return cur_linear() % kWave; }

inline void syncthreads() {
    Block & b = g_blk;
    unsigned gen = b.bar_gen;
    if (++b.bar_count == (unsigned)b.alive) {
        b.bar_count = 0;
        b.bar_gen++;
        if (g_lockstep_release) yield();  // stress scheduling: the releasing thread resumes together with its wave
    } else {
#ifdef BZ3_EMU_WATCH
        unsigned long long w_ = 0;
        while (b.bar_gen == gen) { yield(); if (++w_ == 2000000ull) fprintf(stderr, "[emu] thread %d stuck in __syncthreads (count %u alive %d)\n", cur_linear(), b.bar_count, b.alive); }
#else
        while (b.bar_gen == gen) yield();
#endif
    }
}

inline void wave_barrier() {
    Block & b = g_blk;
    int w = cur_wave();
    unsigned gen = b.wbar_gen[w];
    if (++b.wbar_count[w] == (unsigned)b.wave_alive[w]) {
        b.wbar_count[w] = 0;
        b.wbar_gen[w]++;
        if (g_lockstep_release) yield();
    } else {
#ifdef BZ3_EMU_WATCH
        unsigned long long w_ = 0;
        while (b.wbar_gen[w] == gen) { yield(); if (++w_ == 2000000ull) fprintf(stderr, "[emu] thread %d stuck in a wave rendezvous (wave %d count %u alive %d)\n", cur_linear(), w, b.wbar_count[w], b.wave_alive[w]); }
#else
        while (b.wbar_gen[w] == gen) yield();
#endif
    }
}

void fiber_exit_hook();
void run_grid(dim3 grid, dim3 block, size_t shmem, const std::function<void()> & body);

template <typename T>
inline T exchange(T v, int src_lane) {
    static_assert(sizeof(T) <= 8, "exchange up to 8 bytes");
    Block & b = g_blk;
    int w = cur_wave();
    uint64_t bits = 0;
    memcpy(&bits, &v, sizeof(T));
    b.slots[w][cur_lane()] = bits;
    wave_barrier();
    uint64_t r = b.slots[w][src_lane & (kWave - 1)];
    wave_barrier();
    T out;
    memcpy(&out, &r, sizeof(T));
    return out;
}

inline uint64_t ballot(int pred) {
    Block & b = g_blk;
    int w = cur_wave();
    b.slots[w][cur_lane()] = pred ? 1 : 0;
    wave_barrier();
    uint64_t m = 0;
    for (int l = 0; l < kWave; l++)
        if (((b.live_mask[w] >> l) & 1) && b.slots[w][l]) m |= 1ull << l;
    wave_barrier();
    return m;
}

}  // namespace emu

#define threadIdx (emu::g_cur->tid)
#define blockIdx (emu::g_blk.bid)
#define blockDim (emu::g_blk.bdim)
#define gridDim (emu::g_blk.gdim)

inline void __syncthreads() { emu::syncthreads(); }
inline void __threadfence() {}
inline void __threadfence_block() {}
inline unsigned long long __ballot(int p) { return emu::ballot(p); }
template <typename T> inline T __shfl(T v, int src, int = 64) { return emu::exchange(v, src); }
template <typename T> inline T __shfl_up(T v, unsigned d, int = 64) {
    int l = emu::cur_lane();
    T r = emu::exchange(v, l >= (int)d ? l - (int)d : l);
    return l >= (int)d ? r : v;
}
template <typename T> inline T __shfl_down(T v, unsigned d, int = 64) {
    int l = emu::cur_lane();
    T r = emu::exchange(v, l + (int)d < 64 ? l + (int)d : l);
    return l + (int)d < 64 ? r : v;
}
template <typename T> inline T __shfl_xor(T v, int m, int = 64) { return emu::exchange(v, emu::cur_lane() ^ m); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
inline int __ffs(unsigned v) { return __builtin_ffs((int)v); }
inline int __clz(unsigned v) { return v ? __builtin_clz(v) : 32; }
inline int __clzll(unsigned long long v) { return v ? __builtin_clzll(v) : 64; }
inline unsigned __brev(unsigned v) {
    unsigned r = 0;
    for (int i = 0; i < 32; i++) r |= ((v >> i) & 1u) << (31 - i);
    return r;
}

template <typename T> inline T atomicAdd(T * p, T v) { T o = *p; *p = (T)(o + v); return o; }
template <typename T> inline T atomicSub(T * p, T v) { T o = *p; *p = (T)(o - v); return o; }
template <typename T> inline T atomicMax(T * p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <typename T> inline T atomicMin(T * p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <typename T> inline T atomicOr(T * p, T v) { T o = *p; *p = (T)(o | v); return o; }
template <typename T> inline T atomicAnd(T * p, T v) { T o = *p; *p = (T)(o & v); return o; }
template <typename T> inline T atomicXor(T * p, T v) { T o = *p; *p = (T)(o ^ v); return o; }
template <typename T> inline T atomicExch(T * p, T v) { T o = *p; *p = v; return o; }
template <typename T> inline T atomicCAS(T * p, T c, T v) { T o = *p; if (o == c) *p = v; return o; }

// ---- runtime API subset -------------------------------------------------------------------
// BZ3_EMU_MALLOC_LIMIT=<bytes>: a single allocation larger than this fails with hipErrorOutOfMemory (tests of the fallback paths)
inline hipError_t hipMalloc(void ** p, size_t n) {
    if (const char * e = getenv("BZ3_EMU_MALLOC_LIMIT"))
        if (n > (size_t)strtoull(e, nullptr, 10)) { *p = nullptr; return hipErrorOutOfMemory; }
    *p = aligned_alloc(256, (n + 255) / 256 * 256 + 256);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
template <typename T> inline hipError_t hipMalloc(T ** p, size_t n) { return hipMalloc((void **)p, n); }
inline hipError_t hipFree(void * p) { free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void ** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
template <typename T> inline hipError_t hipHostMalloc(T ** p, size_t n, unsigned = 0) { return hipMalloc((void **)p, n); }
inline hipError_t hipHostFree(void * p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void * d, const void * s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void * d, const void * s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void * d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void * d, int v, size_t n, hipStream_t = nullptr) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t * s) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t * s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t * s, unsigned, int) { *s = nullptr; return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int * least, int * greatest) { *least = 0; *greatest = 0; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
// BZ3_EMU_DEVICES=<n>: pretend there are n devices (tests of the per-device host threads).  All "devices" share the one
// emulated GPU: kernel launches of different host threads are serialised by a lock in run_grid.
inline hipError_t hipGetDeviceCount(int * n) {
    const char * e = getenv("BZ3_EMU_DEVICES");
    *n = (e && atoi(e) > 0) ? atoi(e) : 1;
    return hipSuccess;
}
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int * d) { *d = 0; return hipSuccess; }
inline const char * hipGetErrorString(hipError_t) { return "emu error"; }
inline hipError_t hipEventCreate(hipEvent_t * e) { *e = new emu_event; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t * e, unsigned) { *e = new emu_event; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float * ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
inline hipError_t hipMemGetInfo(size_t * f, size_t * t) { *f = *t = (size_t)16 << 30; return hipSuccess; }

namespace emu {
template <typename... KArgs, typename... Args>
inline void launch(void (*k)(KArgs...), dim3 grid, dim3 block, size_t shmem, Args... args) {
    std::function<void()> body = [=]() { k(args...); };
    run_grid(grid, block, shmem, body);
}
inline char * dyn_smem() { return g_blk.dyn; }
}  // namespace emu
