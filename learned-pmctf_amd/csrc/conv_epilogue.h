// Shared epilogue of the matrix-core convolution kernels.  Include after pm_device_math.h.
#pragma once
typedef float pm_f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------
// Epilogue of the MFMA convolution kernels (conv_mfma.hip, conv_split.hip): activation, up to two residual adds, store of one fragment (4 couts of one pixel).
// Activation and residual pointers are wave-uniform, so they are resolved ONCE per workgroup (PM_EPILOGUE expands its
// body under compile-time ACT / RES) instead of once per value: a switch inside the unrolled fragment loops compiles
// to scalar branches around 224 copies of tanh / sigmoid per kernel, and the common path (none / relu / leaky) hops
// through ~100 KB of code it never executes.  ACT = -1 / RES = -1: resolved at run time (tanh, sigmoid, odd Cout).
// ACT = -2: run-time activation of a kernel that never sees tanh / sigmoid (the register-tight wide-cout kernels: the
// host routes those activations to the generic kernel, conv_mfma.hip launch<>): their table-driven schedules would cost the
// hot kernels registers (conv3x3s1_wave_kernel<7,2> went to 14 spilled VGPRs and 2.9 instead of 0.93 ms with them inlined).
template <int ACT>
__device__ __forceinline__ float act_c(float v, int act, float slope) {
    if constexpr (ACT == pm::ACT_NONE) return v;
    else if constexpr (ACT == pm::ACT_RELU) return v > 0.0f ? v : 0.0f;
    else if constexpr (ACT == pm::ACT_LEAKY) return v > 0.0f ? v : v * slope;
    else if constexpr (ACT == -2) {
        switch (act) {
        case pm::ACT_RELU: return v > 0.0f ? v : 0.0f;
        case pm::ACT_LEAKY: return v > 0.0f ? v : v * slope;
        default: return v;
        }
    }
    else return pm::apply_act(v, act, slope);
}

template <int ACT, int RES, typename Args>
__device__ __forceinline__ void store_frag(const Args &a, pm_f32x4 v, size_t pbase, int co) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = act_c<ACT>(v[i], a.act, a.slope);
    if constexpr (RES >= 0) {                       // Cout % 4 == 0, residual pointers known
        if constexpr (RES >= 1) { const pm_f32x4 r1 = *(const pm_f32x4 *)(a.res1 + pbase + co); v = v + r1; }
        if constexpr (RES >= 2) { const pm_f32x4 r2 = *(const pm_f32x4 *)(a.res2 + pbase + co); v = v + r2; }
        *(pm_f32x4 *)(a.y + pbase + co) = v;
    } else if ((a.Cout & 3) == 0) {
        if (a.res1) { const pm_f32x4 r1 = *(const pm_f32x4 *)(a.res1 + pbase + co); v = v + r1; }
        if (a.res2) { const pm_f32x4 r2 = *(const pm_f32x4 *)(a.res2 + pbase + co); v = v + r2; }
        *(pm_f32x4 *)(a.y + pbase + co) = v;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (co + i < a.Cout) {
                float s = v[i];
                if (a.res1) s = s + a.res1[pbase + co + i];
                if (a.res2) s = s + a.res2[pbase + co + i];
                a.y[pbase + co + i] = s;
            }
        }
    }
}

#define PM_EPI_RES(A_, ACT_, ...)                                                                                     \
    if ((A_).res1 && (A_).res2) { constexpr int ACT = ACT_, RES = 2; __VA_ARGS__ }                                    \
    else if ((A_).res1) { constexpr int ACT = ACT_, RES = 1; __VA_ARGS__ }                                            \
    else { constexpr int ACT = ACT_, RES = 0; __VA_ARGS__ }
#define PM_EPILOGUE_(RT_, A_, ...)                                                                                    \
    if (((A_).Cout & 3) != 0 || (A_).act > pm::ACT_LEAKY || (A_).act < 0 || ((A_).res2 && !(A_).res1)) {             \
        constexpr int ACT = RT_, RES = -1; __VA_ARGS__                                                                \
    } else if ((A_).act == pm::ACT_NONE) { PM_EPI_RES(A_, pm::ACT_NONE, __VA_ARGS__) }                                \
    else if ((A_).act == pm::ACT_RELU) { PM_EPI_RES(A_, pm::ACT_RELU, __VA_ARGS__) }                                  \
    else { PM_EPI_RES(A_, pm::ACT_LEAKY, __VA_ARGS__) }
#define PM_EPILOGUE(A_, ...) PM_EPILOGUE_(-1, A_, __VA_ARGS__)
// for kernels the host never launches with tanh / sigmoid (see act_c<-2>)
#define PM_EPILOGUE_NOTRANS(A_, ...) PM_EPILOGUE_(-2, A_, __VA_ARGS__)

