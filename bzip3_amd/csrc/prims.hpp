// prims.hpp -- wave64 / workgroup building blocks shared by every kernel file.
// gfx950 wavefronts are 64 lanes wide; every constant below is written for that width.
#pragma once
#include "hipx.hpp"

namespace bz3 {

constexpr int WAVE = 64;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ int wave_id() { return (int)(threadIdx.x >> 6); }
__device__ __forceinline__ u64 lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// Ordering point between the lanes of ONE wave (e.g. "all lanes read, then leaders write" on LDS).
// On the GPU a wave runs in lockstep and LDS operations of a wave complete in order, so this is only
// a compiler scheduling fence; under the test emulation (independent fibers) it is a real rendezvous.
__device__ __forceinline__ void wave_sync() {
#ifdef BZ3_EMU
    emu::wave_barrier();
#else
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#endif
}

// Workgroup-scope release / acquire around hand-offs through LDS between waves of one workgroup
// (all waves of a workgroup share the CU's LDS; only ordering is needed, no cache maintenance).
__device__ __forceinline__ void lds_release() {
#ifndef BZ3_EMU
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
#endif
}
__device__ __forceinline__ void lds_acquire() {
    // Consumer side of an LDS hand-off inside one workgroup.  A wave's LDS operations are serviced in issue order
    // and the producer drained its LDS writes (lds_release) before it published the flag, so once the flag read
    // has returned the new value, later LDS reads of this wave see the data: only the COMPILER must not hoist
    // them.  (A workgroup-scope acquire fence would also wait for vmcnt(0), i.e. for this wave's outstanding
    // global stores -- measured at ~550 cycles per coded byte in the CM coder.)
#ifndef BZ3_EMU
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "singlethread");
    asm volatile("" ::: "memory");
#endif
}

// Drop this CU's vector-L1 lines so that later plain loads see what global atomics (executed at L2) wrote.
__device__ __forceinline__ void l1_invalidate() {
#ifndef BZ3_EMU
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
}

// Load that bypasses the CU's vector L1 (served by L2): for words that other lanes update with global
// atomics (which execute at L2) inside the same kernel.
__device__ __forceinline__ u32 ld_l2(const u32 * p) {
#ifdef BZ3_EMU
    return *p;
#else
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

template <typename T>
__device__ __forceinline__ T wave_incl_add(T v) {
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        T t = __shfl_up(v, (unsigned)d);
        if (l >= d) v += t;
    }
    return v;
}

template <typename T>
__device__ __forceinline__ T wave_incl_max(T v) {
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        T t = __shfl_up(v, (unsigned)d);
        if (l >= d && t > v) v = t;
    }
    return v;
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        T t = __shfl_xor(v, d);
        if (t > v) v = t;
    }
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_min(T v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        T t = __shfl_xor(v, d);
        if (t < v) v = t;
    }
    return v;
}
__device__ __forceinline__ u32 wave_xor(u32 v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v ^= __shfl_xor(v, d);
    return v;
}

// Workgroup exclusive prefix sum of one value per thread.  `lds` needs BLOCK/64 + 1 words.
// Returns the exclusive prefix; `total` receives the workgroup sum.  Ends with a barrier so
// `lds` may be reused immediately.
template <int BLOCK, typename T>
__device__ __forceinline__ T block_excl_add(T v, T * lds, T & total) {
    constexpr int NW = BLOCK / WAVE;
    const T incl = wave_incl_add(v);
    if (lane_id() == WAVE - 1) lds[wave_id()] = incl;
    __syncthreads();
    if (wave_id() == 0) {
        T x = lane_id() < NW ? lds[lane_id()] : (T)0;
        T xs = wave_incl_add(x);
        if (lane_id() < NW) lds[lane_id()] = xs - x;
        if (lane_id() == NW - 1) lds[NW] = xs;
    }
    __syncthreads();
    const T r = incl - v + lds[wave_id()];
    total = lds[NW];
    __syncthreads();
    return r;
}

// Workgroup inclusive running maximum (one value per thread).  `lds` needs BLOCK/64 words.
template <int BLOCK, typename T>
__device__ __forceinline__ T block_incl_max(T v, T * lds) {
    constexpr int NW = BLOCK / WAVE;
    T incl = wave_incl_max(v);
    if (lane_id() == WAVE - 1) lds[wave_id()] = incl;
    __syncthreads();
    T carry = 0;
    for (int w = 0; w < wave_id(); w++) {
        T t = lds[w];
        if (t > carry) carry = t;
    }
    __syncthreads();
    (void)NW;
    return incl > carry ? incl : carry;
}

template <int BLOCK, typename T>
__device__ __forceinline__ T block_sum(T v, T * lds) {
    constexpr int NW = BLOCK / WAVE;
    v = wave_sum(v);
    if (lane_id() == 0) lds[wave_id()] = v;
    __syncthreads();
    T r = 0;
    for (int w = 0; w < NW; w++) r += lds[w];
    __syncthreads();
    return r;
}
template <int BLOCK, typename T>
__device__ __forceinline__ T block_min(T v, T * lds) {
    constexpr int NW = BLOCK / WAVE;
    v = wave_min(v);
    if (lane_id() == 0) lds[wave_id()] = v;
    __syncthreads();
    T r = lds[0];
    for (int w = 1; w < NW; w++) r = lds[w] < r ? lds[w] : r;
    __syncthreads();
    return r;
}
template <int BLOCK, typename T>
__device__ __forceinline__ T block_max(T v, T * lds) {
    constexpr int NW = BLOCK / WAVE;
    v = wave_max(v);
    if (lane_id() == 0) lds[wave_id()] = v;
    __syncthreads();
    T r = lds[0];
    for (int w = 1; w < NW; w++) r = lds[w] > r ? lds[w] : r;
    __syncthreads();
    return r;
}

// Device addresses travel inside job structs as plain 64-bit integers (u64).  A pointer field would be "generic" to
// the compiler, which then emits FLAT loads/stores; flat operations also tick lgkmcnt, so every wait for an LDS
// read would stall on outstanding global stores as well.  Re-materialising the address as an address_space(1)
// pointer makes every access through it a global_load / global_store (vmcnt only).
template <typename T>
__device__ __forceinline__ T * global_ptr(u64 raw) {
#ifdef BZ3_EMU
    return reinterpret_cast<T *>(raw);
#else
    return (T *)(__attribute__((address_space(1))) T *)raw;
#endif
}
inline u64 dev_addr(const void * p) { return (u64)reinterpret_cast<uintptr_t>(p); }

// One 32-bit load at any byte address (gfx950 global/LDS accesses need no natural alignment).
struct __attribute__((packed)) PackedU32 { u32 v; };
struct __attribute__((packed)) PackedU128 { u32 v[4]; };  // one 128-bit access at any byte address
__device__ __forceinline__ u32 load_u32_any(const u8 * p) { return reinterpret_cast<const PackedU32 *>(p)->v; }

__device__ __forceinline__ u32 load_le32(const u8 * p) {
    return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24);
}
__device__ __forceinline__ u32 load_be32(const u8 * p) {
    return ((u32)p[0] << 24) | ((u32)p[1] << 16) | ((u32)p[2] << 8) | (u32)p[3];
}

inline u32 ceil_div(u64 a, u64 b) { return (u32)((a + b - 1) / b); }

}  // namespace bz3
