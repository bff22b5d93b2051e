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
