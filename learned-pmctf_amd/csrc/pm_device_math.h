// PM-F32 scalar arithmetic on the device: the same operation sequences as the
// arithmetic spec in DESIGN.md §"PM-F32" (one IEEE binary32 rounding per written
// operation, explicit fmaf only; build with -ffp-contract=off).
// Replaces torch.tanh / torch.sigmoid / torch.log of the reference
// (pMCTF/layers/lifting_1d.py:39,42; pMCTF/layers/long_context.py:24-32;
//  pMCTF/entropy_models/entropy_models.py:271).
#pragma once
#include <hip/hip_runtime.h>
#include "pm_tanh_tables.h"
#include "pm_log_tables.h"
#define PM_SLEEF_FN __device__ __forceinline__
#include "pm_sleef_f32.h"
#define PM_GLIBC_FN __device__ __forceinline__
#include "pm_glibc_expf.h"

namespace pm {

__device__ __forceinline__ float u2f(unsigned u) { return __uint_as_float(u); }
__device__ __forceinline__ unsigned f2u(float f) { return __float_as_uint(f); }

// exp(x) = (1+q) * 2^n
__device__ __forceinline__ float exp_core(float x, int &n) {
    x = x < -87.0f ? -87.0f : x;
    x = x > 88.0f ? 88.0f : x;
    const float nf = __builtin_rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(nf, -0.693145751953125f, x);
    r = __builtin_fmaf(nf, -1.42860682030941723212e-6f, r);
    float p = 1.9841270e-4f;
    p = __builtin_fmaf(p, r, 1.3888889e-3f);
    p = __builtin_fmaf(p, r, 8.3333333e-3f);
    p = __builtin_fmaf(p, r, 4.1666667e-2f);
    p = __builtin_fmaf(p, r, 1.6666667e-1f);
    p = __builtin_fmaf(p, r, 0.5f);
    const float r2 = r * r;
    n = (int)nf;
    return __builtin_fmaf(p, r2, r);
}

__device__ __forceinline__ float expf_(float x) {
    int n;
    const float q = exp_core(x, n);
    const float s = u2f((unsigned)(n + 127) << 23);
    return (q + 1.0f) * s;
}

// tanh: the schedule of Intel MKL's vmsTanh (high accuracy, AVX-512 kernel) = torch.tanh on a float32 CPU tensor, bit for
// bit (oracle/c/pm_math.h pm_tanhf has the derivation; tools/mkl_tanh_tables.py --verify checks it on all 2^32 inputs).
// 32 intervals of |x|, 12 words per interval = three 16-byte loads from a 1.5 KB table that lives in L1 / the scalar cache.
struct __attribute__((aligned(16))) TanhRow { unsigned w[12]; };
static __device__ const TanhRow tanh_rows[32] = PM_TANH_TABLE_BY_INTERVAL;

// rows: the 32 x 48-byte interval table, in global memory (tanh_rows) or copied to LDS by the workgroup (tanh_rows_to_lds:
// the kernels that take tanh of every value keep it there — three 16-byte LDS reads per value instead of three divergent
// global loads: pu_fused 567 -> see DESIGN, conv3x3_cin1_pix16 162 -> ..)
__device__ __forceinline__ float tanhf_rows(float x, const uint4 *rows) {
    const unsigned ux = f2u(x);
    const unsigned ix = ux & PM_TANH_EXPMASK;
    int t = (int)(ix - PM_TANH_BIAS);
    t = t < 0 ? 0 : t;
    t = t > (int)PM_TANH_IDXMAX ? (int)PM_TANH_IDXMAX : t;
    const uint4 *row = rows + 3 * (t >> 21);
    const uint4 r0 = row[0], r1 = row[1], r2 = row[2];        // B T_hi T_lo C1 | C3 C4 C5 C6 | C7 - - -
    const float y = u2f(ux & PM_TANH_ABS) - u2f(r0.x);
    float p = u2f(r2.x);
    p = __builtin_fmaf(p, y, u2f(r1.w));
    p = __builtin_fmaf(p, y, u2f(r1.z));
    p = __builtin_fmaf(p, y, u2f(r1.y));
    p = __builtin_fmaf(p, y, u2f(r1.x));
    p = p * y;
    p = __builtin_fmaf(p, y, u2f(r0.z));
    p = __builtin_fmaf(u2f(r0.w), y, p);
    float r = u2f(f2u(p + u2f(r0.y)) | (ux & PM_TANH_SIGN));
    if ((int)ix > (int)PM_TANH_BIG) {                         // |x| >= 2^127 * 1.25, infinities, NaN
        const bool nan = (ux & 0x7f800000u) == 0x7f800000u && (ux & 0x007fffffu);
        r = nan ? x + x : u2f((ux & PM_TANH_SIGN) | 0x3f800000u);
    }
    return r;
}

__device__ __forceinline__ float tanhf_(float x) { return tanhf_rows(x, (const uint4 *)tanh_rows[0].w); }

constexpr int TANH_LDS_UINT4 = 96;                            // 32 rows x 3
// every thread of the workgroup calls this, then __syncthreads(): dst = the table in LDS
__device__ __forceinline__ void tanh_rows_to_lds(uint4 *dst, int tid, int nthreads) {
    const uint4 *src = (const uint4 *)tanh_rows[0].w;
    for (int i = tid; i < TANH_LDS_UINT4; i += nthreads) dst[i] = src[i];
}

// torch.sigmoid of a float CPU tensor: 1 / (1 + Sleef_expf16_u10(0 - x)), transcribed in pm_sleef_f32.h
__device__ __forceinline__ float sigmoidf_(float x) { return pm_aten_sigmoidf(x); }

// natural log: the schedule of Intel MKL's vmsLn (high accuracy, AVX-512 kernel) = torch.log on a float32 CPU tensor, bit
// for bit on [2^-100, 2^100) (oracle/c/pm_math.h pm_logf has the derivation; tools/mkl_log_tables.py --verify the check);
// outside that range, which the path never reaches (scales are clamped to >= 1e-5), the former polynomial schedule.
static __device__ const unsigned log_step_m[36] = PM_LOG_STEP_M_PADDED;     // 33 thresholds + sentinels
static __device__ const unsigned log_step_r[33] = PM_LOG_STEP_R;
static __device__ const unsigned log_bucket[32] = PM_LOG_BUCKET;            // step at the start of mantissa bucket m >> 18
static __device__ const unsigned log_thi[32] = PM_LOG_THI;
static __device__ const unsigned log_tlo[32] = PM_LOG_TLO;

__device__ __forceinline__ float logf_poly_(float x) {
    const unsigned u = f2u(x);
    int e = (int)(u >> 23) - 127;
    float m = u2f((u & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421356237f) { m = m * 0.5f; e += 1; }
    const float f = m - 1.0f;
    const float s = f / (2.0f + f);
    const float z = s * s;
    float p = 0.2222222222f;
    p = __builtin_fmaf(p, z, 0.2857142857f);
    p = __builtin_fmaf(p, z, 0.4f);
    p = __builtin_fmaf(p, z, 0.6666666667f);
    const float lm = __builtin_fmaf(p * z, s, 2.0f * s);
    const float ef = (float)e;
    const float lo = __builtin_fmaf(ef, 9.0580006145e-6f, lm);
    return __builtin_fmaf(ef, 0.693138123f, lo);
}

__device__ __forceinline__ float logf_(float x) {
    const unsigned ux = f2u(x);
    const int E = (int)(ux >> 23) - 127;
    if ((ux >> 31) || E < -100 || E >= 100) return logf_poly_(x);
    const unsigned m = ux & 0x007fffffu;
    unsigned k = log_bucket[m >> 18];                   // at most three thresholds fall into one bucket
    k += log_step_m[k + 1] <= m;
    k += log_step_m[k + 1] <= m;
    k += log_step_m[k + 1] <= m;
    const unsigned rb = log_step_r[k] - ((unsigned)E << 23);
    const float R = u2f(rb);
    const int i = (int)(rb >> 18) & 31;
    const float e = (float)((int)(rb >> 23) - 127);
    const float u = __builtin_fmaf(R, x, -u2f(PM_LOG_ONE));
    const float lo = __builtin_fmaf(e, -u2f(PM_LOG_LN2LO), u2f(log_tlo[i]));
    const float hi = __builtin_fmaf(-u2f(PM_LOG_LN2HI), e, u2f(log_thi[i]));
    float p = __builtin_fmaf(u2f(PM_LOG_C4), u, u2f(PM_LOG_C3));
    const float u2 = u * u;
    p = __builtin_fmaf(p, u, u2f(PM_LOG_C2));
    p = __builtin_fmaf(p, u2, lo);
    const float s = u + hi;
    const float t = s - hi;
    float r = u - t;
    r = r + p;
    return s + r;
}

// activations used in conv epilogues
enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_LEAKY = 2, ACT_TANH = 3, ACT_SIGMOID = 4 };

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
    switch (act) {
    case ACT_RELU: return v > 0.0f ? v : 0.0f;            // torch.relu: max(v,0); -0 -> 0 numerically equal
    case ACT_LEAKY: return v > 0.0f ? v : v * slope;      // F.leaky_relu
    case ACT_TANH: return tanhf_(v);
    case ACT_SIGMOID: return sigmoidf_(v);
    default: return v;
    }
}

}  // namespace pm
