// Internal helpers shared by the HIP translation units of libcpe_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/cpe.h"

namespace cpe {

void set_error(const char *fmt, ...);

#define CPE_CHECK_ARG(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            cpe::set_error(__VA_ARGS__);         \
            return CPE_ERR_ARG;                  \
        }                                        \
    } while (0)

// clear any stale sticky error left by other users of the runtime (e.g. torch device probing)
#define CPE_LAUNCH_BEGIN() (void)hipGetLastError()

#define CPE_CHECK_LAUNCH(name)                                                      \
    do {                                                                            \
        hipError_t e_ = hipGetLastError();                                          \
        if (e_ != hipSuccess) {                                                     \
            cpe::set_error("%s: launch failed: %s", name, hipGetErrorString(e_));   \
            return CPE_ERR_LAUNCH;                                                  \
        }                                                                           \
    } while (0)

#define CPE_CHECK_HIP(expr)                                                         \
    do {                                                                            \
        hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess) {                                                     \
            cpe::set_error("%s: %s", #expr, hipGetErrorString(e_));                 \
            return CPE_ERR_LAUNCH;                                                  \
        }                                                                           \
    } while (0)

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// BORDER_REFLECT_101
__device__ __forceinline__ int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        else i = 2 * (n - 1) - i;
    }
    return i;
}

// optional per-kernel hipEvent timers (cpe_profile_enable / cpe_profile_report); off by default, and a
// no-op while off so the launch functions stay graph-capturable
void prof_begin(const char *name, hipStream_t s);
void prof_end(hipStream_t s);

#define CPE_KLAUNCH(kernel, grid, block, shmem, stream, ...)                         \
    do {                                                                             \
        cpe::prof_begin(#kernel, stream);                                            \
        hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);         \
        cpe::prof_end(stream);                                                       \
    } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace cpe
