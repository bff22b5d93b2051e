// Launch helper: hipGetLastError() is per-thread sticky state that other runtime users in the process (e.g.
// torch's event queries returning hipErrorNotReady) may have set; clear it before our launch so that the status we
// return describes OUR launch only.
#pragma once
#include <hip/hip_runtime.h>
#define PM_LAUNCH(kernel, grid, block, smem, stream, ...)                       \
    do {                                                                        \
        (void)hipGetLastError();                                                \
        hipLaunchKernelGGL(kernel, grid, block, smem, stream, __VA_ARGS__);     \
    } while (0)

#include <stdio.h>
// status of the launch just issued; on failure the HIP error string goes to stderr (the C ABI returns only a code)
static inline int pm_launch_status() {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return 0;
    fprintf(stderr, "libpmctf_hip: kernel launch failed: %s (%s)\n", hipGetErrorName(e), hipGetErrorString(e));
    return -2;
}

// Flat indices of the elementwise / few-channel kernels are non-negative and, for every plane of the path, below 2^31:
// divide on 32 bits whenever both operands fit.  A 64-bit integer division is an emulated sequence of about a hundred
// vector-ALU instructions, which made kernels with a handful of instructions per element ALU-bound instead of
// HBM-bound (tools/bench_hbm.py: plane add 12.3 us against 3.9 us for a copy of the same bytes).  The branch is
// uniform over a launch in practice.
__device__ __forceinline__ long pm_div(long x, long d) {
    return ((x | d) >> 31) == 0 ? (long)((unsigned)x / (unsigned)d) : x / d;
}
__device__ __forceinline__ long pm_mod(long x, long d) {
    return ((x | d) >> 31) == 0 ? (long)((unsigned)x % (unsigned)d) : x % d;
}
