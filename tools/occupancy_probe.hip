// occupancy_probe.hip -- how many workgroups of a given shape share one MI355X CU?  (round 2: the 320-thread CM decoders did not
// reach the co-residency their LDS size allows; this probe separates the effect of threads per workgroup from LDS per workgroup.)
//   hipcc --offload-arch=gfx950 -O2 -o tools/occupancy_probe.bin tools/occupancy_probe.hip && tools/occupancy_probe.bin
// Every workgroup spins for a fixed number of cycles (one lane reads the cycle counter, the others wait in barriers), so the
// launch time of G workgroups is ceil(G / (CUs x resident workgroups per CU)) x spin time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

extern "C" __global__ void spin(unsigned long long cycles, unsigned * sink) {
    extern __shared__ unsigned lds[];
    if (threadIdx.x == 0) lds[0] = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned it = 0;
    while (__builtin_readcyclecounter() - t0 < cycles) {
        __syncthreads();
        if (threadIdx.x == 0) lds[0] = ++it;
        __syncthreads();
    }
    if (threadIdx.x == 0 && lds[0] == 0xFFFFFFFFu) *sink = it;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("# %s: %d CUs, %zu B LDS per workgroup max, %d max threads per CU\n", p.gcnArchName, cus, p.sharedMemPerBlock, p.maxThreadsPerMultiProcessor);
    unsigned * sink;
    hipMalloc(&sink, 4);
    hipFuncSetAttribute((const void *)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const unsigned long long cycles = 20ull * 1000 * 1000;  // ~8 ms at 2.4 GHz (the counter may tick at 100 MHz: then it is just longer)
    struct Cfg { int threads, lds; };
    std::vector<Cfg> cfgs;
    for (int threads : {64, 128, 192, 256, 320, 384, 512})
        for (int lds : {160 * 1024 / 4 - 256, 49764, 50796, 52140, 53248, 71044, 72088, 79536}) cfgs.push_back({threads, lds});
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (const Cfg & c : cfgs) {
        float base = 0.f;
        printf("threads %4d lds %6d :", c.threads, c.lds);
        for (int per_cu = 1; per_cu <= 4; per_cu++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(spin, dim3(cus * per_cu), dim3(c.threads), c.lds, 0, cycles, sink);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            if (per_cu == 1) base = ms;
            printf("  %dx: %.2f", per_cu, ms / base);
        }
        printf("   (1x = %.1f ms)\n", base);
    }
    return 0;
}
