// tools/ubench_coder.hip -- micro-benchmarks of the arithmetic-coder recurrence step variants (one wave).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32; typedef uint64_t u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// A: VALU unified multiplier, 2 x v_mad_u64_u32, branch on vcc
__global__ void k_a(u32 n, u32 m, u64 * out) {
    u32 z; asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    u32 range = 0xFFFFFFFFu ^ z; u64 low = z; u32 M = m ^ z, neg = z, s = z;
    u64 t0 = __builtin_amdgcn_s_memtime();
    for (u32 i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const u64 prod = (u64)range * M + (((u64)neg << 32) | neg);
            const u32 r2 = (u32)(prod >> 18);
            low += (u64)(range - r2) * s;
            range = r2;
            if (__ballot(range < (1u << 24)) != 0ull) { range = (range << 8) | 0xFF; low <<= 8; }
        }
    }
    u64 t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = range + low; }
}
// B: VALU, low via and/add
__global__ void k_b(u32 n, u32 m, u64 * out) {
    u32 z; asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    u32 range = 0xFFFFFFFFu ^ z, low = z; u32 M = m ^ z, neg = z;
    u64 t0 = __builtin_amdgcn_s_memtime();
    for (u32 i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const u64 prod = (u64)range * M + (((u64)neg << 32) | neg);
            const u32 r2 = (u32)(prod >> 18);
            low += (range - r2) & neg;
            range = r2;
            if (__ballot(range < (1u << 24)) != 0ull) { range = (range << 8) | 0xFF; low <<= 8; }
        }
    }
    u64 t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = range + low; }
}
// C: SALU unified multiplier
__global__ void k_c(u32 n, u32 m, u64 * out) {
    u32 range = 0xFFFFFFFFu, low = 0; u32 M = __builtin_amdgcn_readfirstlane(m), neg = __builtin_amdgcn_readfirstlane(n >> 31);
    u64 t0 = __builtin_amdgcn_s_memtime();
    for (u32 i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const u64 prod = (u64)range * M + (((u64)neg << 32) | neg);
            const u32 r2 = (u32)(prod >> 18);
            low += (range - r2) & neg;
            range = r2;
            if (range < (1u << 24)) { range = (range << 8) | 0xFF; low <<= 8; }
            asm volatile("" : "+s"(range), "+s"(low));
        }
    }
    u64 t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = range + low; }
}
// D: VALU with 32x32->hi/lo multiplies instead of mad_u64 (v_mul_hi_u32 + v_mul_lo_u32 + alignbit), no addend
__global__ void k_d(u32 n, u32 m, u64 * out) {
    u32 z; asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    u32 range = 0xFFFFFFFFu ^ z, low = z; u32 M = m ^ z, neg = z;
    u64 t0 = __builtin_amdgcn_s_memtime();
    for (u32 i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const u32 hi = __umulhi(range, M), lo = range * M;
            const u32 r2 = (hi << 14) | (lo >> 18);
            low += (range - r2) & neg;
            range = r2;
            if (__ballot(range < (1u << 24)) != 0ull) { range = (range << 8) | 0xFF; low <<= 8; }
        }
    }
    u64 t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = range + low; }
}
// E: A without the per-event branch (renorm folded branch-free) -- cost of the branch
__global__ void k_e(u32 n, u32 m, u64 * out) {
    u32 z; asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    u32 range = 0xFFFFFFFFu ^ z, low = z; u32 M = m ^ z, neg = z;
    u64 t0 = __builtin_amdgcn_s_memtime();
    for (u32 i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const u64 prod = (u64)range * M + (((u64)neg << 32) | neg);
            const u32 r2 = (u32)(prod >> 18);
            low += (range - r2) & neg;
            const u32 sh = r2 < (1u << 24) ? 8u : 0u;
            range = (r2 << sh) | ((1u << sh) - 1u);
        }
    }
    u64 t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = range + low; }
}
// F: 24-bit friendly: v_mul_u32_u24 / v_mul_hi_u32_u24 when range < 2^24?  (not exact for the codec; rate probe only)
__global__ void k_f(u32 n, u32 m, u64 * out) {
    u32 z; asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    u32 range = 0x00FFFFFFu ^ z, low = z; u32 M = m ^ z;
    u64 t0 = __builtin_amdgcn_s_memtime();
    for (u32 i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const u32 hi = __umul24(range >> 8, M) >> 16, lo = __umul24(range, M);
            const u32 r2 = ((hi << 14) | (lo >> 18)) | 0x800000u;
            low += range - r2;
            range = r2 & 0xFFFFFFu;
        }
    }
    u64 t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = range + low; }
}
template <typename F> int run(const char * name, F launch, double ops) {
    u64 * d; CK(hipMalloc(&d, 16)); launch(d); CK(hipDeviceSynchronize()); launch(d); CK(hipDeviceSynchronize());
    u64 h[2]; CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
    printf("%-44s cycles/event %.1f\n", name, h[0] / ops); (void)hipFree(d); return 0;
}
int main() {
    const u32 n = 100000, m = 0x3F000;  // P close to 1: range shrinks slowly, renorm rarely taken
    run("A valu 2x mad_u64, vcc branch", [&](u64 * d) { k_a<<<1, 64>>>(n, m, d); }, n * 8.0);
    run("B valu mad_u64 + and/add, vcc branch", [&](u64 * d) { k_b<<<1, 64>>>(n, m, d); }, n * 8.0);
    run("C salu mul_hi/mul/addc/lshr, scc branch", [&](u64 * d) { k_c<<<1, 64>>>(n, m, d); }, n * 8.0);
    run("D valu mul_hi + mul_lo + shifts, vcc branch", [&](u64 * d) { k_d<<<1, 64>>>(n, m, d); }, n * 8.0);
    run("E valu mad_u64, branch-free renorm", [&](u64 * d) { k_e<<<1, 64>>>(n, m, d); }, n * 8.0);
    run("F valu 24-bit multiplies (rate probe)", [&](u64 * d) { k_f<<<1, 64>>>(n, m, d); }, n * 8.0);
    return 0;
}
