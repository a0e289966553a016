// tools/ubench.hip -- single-wave micro-benchmarks that calibrate the CM coder design:
// clock rate, dependent SALU/VALU issue cadence, LDS round-trip latency, s_barrier cost.
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_salu_add(uint32_t n, uint32_t seed, uint64_t * out) {
    uint32_t a = __builtin_amdgcn_readfirstlane(seed);
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 64; k++) asm volatile("s_add_u32 %0, %0, %0" : "+s"(a) : : "scc");
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = a; }
}
__global__ void k_salu_mul(uint32_t n, uint32_t seed, uint64_t * out) {
    uint32_t a = __builtin_amdgcn_readfirstlane(seed);
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 64; k++) asm volatile("s_mul_hi_u32 %0, %0, %0" : "+s"(a));
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = a; }
}
__global__ void k_salu_indep(uint32_t n, uint32_t seed, uint64_t * out) {
    uint32_t a = __builtin_amdgcn_readfirstlane(seed), b = a + 1, c = a + 2, d = a + 3;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) asm volatile("s_add_u32 %0, %0, %0\n s_add_u32 %1, %1, %1\n s_add_u32 %2, %2, %2\n s_add_u32 %3, %3, %3" : "+s"(a), "+s"(b), "+s"(c), "+s"(d) : : "scc");
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = a + b + c + d; }
}
__global__ void k_valu_add(uint32_t n, uint32_t seed, uint64_t * out) {
    uint32_t a = seed + threadIdx.x;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 64; k++) asm volatile("v_add_u32 %0, %0, %0" : "+v"(a));
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = a; }
}
__global__ void k_valu_mad64(uint32_t n, uint32_t seed, uint64_t * out) {
    uint32_t a = seed + threadIdx.x;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 64; k++) { uint64_t p = (uint64_t)a * 0x3FFF1u; a = (uint32_t)(p >> 18) | 1u; asm volatile("" : "+v"(a)); }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = a; }
}
__global__ void k_readlane_chain(uint32_t n, uint32_t seed, uint64_t * out) {
    uint32_t v = seed * (threadIdx.x + 1);
    uint32_t s = 1;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 32; k++) { s = __builtin_amdgcn_readlane(v, s & 63) ; asm volatile("s_add_u32 %0, %0, 1" : "+s"(s) : : "scc"); }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = s; }
}
__global__ void k_lds_chase(uint32_t n, uint64_t * out) {
    __shared__ uint16_t tab[32768];
    for (int i = threadIdx.x; i < 32768; i += blockDim.x) tab[i] = (uint16_t)((i * 7919 + 13) & 32767);
    __syncthreads();
    uint32_t p = threadIdx.x;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 32; k++) p = tab[p];
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = p; }
}
__global__ void k_lds_chase_uniform(uint32_t n, uint64_t * out) {  // all lanes same address (broadcast)
    __shared__ uint16_t tab[32768];
    for (int i = threadIdx.x; i < 32768; i += blockDim.x) tab[i] = (uint16_t)((i * 7919 + 13) & 32767);
    __syncthreads();
    uint32_t p = 5;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 32; k++) p = tab[p];
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = p; }
}
__global__ void k_barrier(uint32_t n, uint64_t * out) {
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) __syncthreads();
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = 0; }
}
__global__ void k_lds_pingpong(uint32_t n, uint64_t * out) {  // two waves hand a token back and forth through LDS
    __shared__ uint32_t flag;
    if (threadIdx.x == 0) flag = 0;
    __syncthreads();
    const uint32_t w = threadIdx.x >> 6;
    volatile uint32_t * f = &flag;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t want = 2 * i + w;
        while (*f != want) {}
        if ((threadIdx.x & 63) == 0) *f = want + 1;
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = 0; }
}

template <typename F>
int run(const char * name, F launch, double ops) {
    uint64_t * d; CK(hipMalloc(&d, 16));
    launch(d); CK(hipDeviceSynchronize());
    auto t0 = std::chrono::steady_clock::now();
    launch(d); CK(hipDeviceSynchronize());
    auto t1 = std::chrono::steady_clock::now();
    uint64_t h[2]; CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
    double ns = std::chrono::duration<double, std::nano>(t1 - t0).count();
    printf("%-22s ticks/op %.2f   ns/op %.2f   (ticks %llu, %.3f ms => %.1f MHz tick)\n", name, h[0] / ops, ns / ops, (unsigned long long)h[0], ns / 1e6, h[0] / (ns / 1e3));
    hipFree(d);
    return 0;
}

int main() {
    const uint32_t n = 200000;
    run("salu add dep", [&](uint64_t * d) { k_salu_add<<<1, 64>>>(n, 3, d); }, n * 64.0);
    run("salu mul_hi dep", [&](uint64_t * d) { k_salu_mul<<<1, 64>>>(n, 3, d); }, n * 64.0);
    run("salu add indep x4", [&](uint64_t * d) { k_salu_indep<<<1, 64>>>(n, 3, d); }, n * 64.0);
    run("valu add dep", [&](uint64_t * d) { k_valu_add<<<1, 64>>>(n, 3, d); }, n * 64.0);
    run("valu mad64+align dep", [&](uint64_t * d) { k_valu_mad64<<<1, 64>>>(n, 3, d); }, n * 64.0);
    run("readlane+sadd dep", [&](uint64_t * d) { k_readlane_chain<<<1, 64>>>(n, 3, d); }, n * 32.0);
    run("lds u16 chase", [&](uint64_t * d) { k_lds_chase<<<1, 64>>>(n / 4, d); }, n / 4 * 32.0);
    run("lds u16 chase uniform", [&](uint64_t * d) { k_lds_chase_uniform<<<1, 64>>>(n / 4, d); }, n / 4 * 32.0);
    run("barrier 2 waves", [&](uint64_t * d) { k_barrier<<<1, 128>>>(n / 4, d); }, n / 4 * 16.0);
    run("barrier 5 waves", [&](uint64_t * d) { k_barrier<<<1, 320>>>(n / 4, d); }, n / 4 * 16.0);
    run("lds pingpong 2 waves", [&](uint64_t * d) { k_lds_pingpong<<<1, 128>>>(n / 4, d); }, n / 4 * 2.0);
    return 0;
}
