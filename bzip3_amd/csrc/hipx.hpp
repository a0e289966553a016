// hipx.hpp -- the one include every source in bzip3_amd/csrc uses for the HIP runtime.
//
// Product build (hipcc --offload-arch=gfx950): <hip/hip_runtime.h>, kernels run on the MI355X.
// There is NO CPU code path in the product: without a HIP device bz3_new() fails.
//
// -DBZ3_EMU is set only by tests/emu/build_emu.py, which compiles these same sources against a
// test-only fiber emulation of the HIP execution model so that kernel logic can be diffed against
// the oracle on the GPU-less build container.  That library lives under tests/ and is never
// loaded by the bzip3_amd package.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#ifdef BZ3_EMU
#include "hip_emu.hpp"
#define BZ3_DYN_SMEM(name) char * name = emu::dyn_smem()
#define BZ3_SPIN_PAUSE() emu::yield()
#define BZ3_SPIN_TIGHT() emu::yield()
#else
#include <hip/hip_runtime.h>
#define BZ3_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
#define BZ3_SPIN_PAUSE() __builtin_amdgcn_s_sleep(1)
#define BZ3_SPIN_TIGHT() ((void)0)  // latency-critical hand-offs poll without sleeping
#endif

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int8_t s8;
typedef int32_t s32;
typedef int64_t s64;

namespace bz3 {

// Launch helper: `k<<<grid, block, shmem, stream>>>(args...)` on the GPU, fiber grid under emulation.
template <typename... KArgs, typename... Args>
inline void launch(void (*k)(KArgs...), dim3 grid, dim3 block, size_t shmem, hipStream_t stream, Args... args) {
#ifdef BZ3_EMU
    (void)stream;
    emu::launch(k, grid, block, shmem, static_cast<KArgs>(args)...);
#else
    hipLaunchKernelGGL(k, grid, block, shmem, stream, static_cast<KArgs>(args)...);
#endif
}

struct HipError {
    hipError_t code;
    const char * what;
    const char * file;
    int line;
};

}  // namespace bz3

// Any HIP runtime failure is turned into a C++ exception that the C-ABI layer maps to a bz3 error
// code (BZ3_ERR_INIT at allocation time, BZ3_ERR_BWT at run time: SURVEY.md section 8b).
#define HIP_CHECK(expr)                                                                   \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) throw bz3::HipError{_e, hipGetErrorString(_e), __FILE__, __LINE__}; \
    } while (0)
