// estimate_ops.hip — estimate-mode forward: Laplace / factorized bit ESTIMATES and squared-error sums instead of
// range coding (pMCTF_L.py:244-295,332-379; pWave.py:243-312; gaussian_model.py:36-67).
//
// PM-F32 definition (identical line by line to oracle/pmctf_oracle/kernels.py:CdefK): per element, in f32 with one
// rounding per written operation,
//     sigma = min(max(s, 1e-5), 1e10)
//     cdf(v) = 0.5 - (0.5 * sign(v)) * (pm_exp(-|v| / sigma) - 1)
//     p     = cdf(y + 0.5) - cdf(y - 0.5)
//     bits  = max(-1 * pm_log(p + 1e-5) / (float)ln 2, 0)
// and every total is an f64 sum of those f32 values (block tree + one f64 atomic per wave), so totals are reproducible
// to f64 rounding.  All entry points ACCUMULATE into device doubles the caller has zeroed.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pm_device_math.h"
#include "launch.h"
#include "../../include/pmctf_hip.h"

namespace {

inline int launch_ok() { return pm_launch_status(); }
inline unsigned grid_for(long n, unsigned cap = 4096) {
    long g = (n + 255) / 256;
    if (g < 1) g = 1;
    return g > cap ? cap : (unsigned)g;
}

__device__ __forceinline__ float lap_cdf(float v, float sigma) {
    const float e = pm::expf_(-__builtin_fabsf(v) / sigma) - 1.0f;
    const float sgn = v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f);
    return 0.5f - (0.5f * sgn) * e;
}

__device__ __forceinline__ float neglog2(float p) {
    const float bits = (-1.0f * pm::logf_(p + 1e-5f)) / 0.693147182464599609375f;     // (float)math.log(2.0)
    return bits > 0.0f ? bits : 0.0f;
}

__device__ __forceinline__ float laplace_bits(float y, float s) {
    float sigma = s > 1e-5f ? s : 1e-5f;
    sigma = sigma < 1e10f ? sigma : 1e10f;
    return neglog2(lap_cdf(y + 0.5f, sigma) - lap_cdf(y - 0.5f, sigma));
}

// sum of `v` over the wave, one f64 atomic per wave
__device__ __forceinline__ void wave_accumulate(double v, double *dst) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(dst, v);
}

// four-step coder, step k (context_fusion_4step.py:127-137,156-186): x_hat at the class-k positions and their bits
__global__ void fourstep_estimate_kernel(const float *__restrict__ x, const float *__restrict__ params, float *so_far,
                                         int H, int W, int k, int psub, double *bits) {
    const int n = blockIdx.y;
    const long HW = (long)H * W;
    double acc = 0.0;
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < HW; j += (long)gridDim.x * blockDim.x) {
        const long i = n * HW + j;
        const int xw = (int)(j % W);
        const int yy = (int)(j / W);
        if ((yy & 1) * 2 + (xw & 1) == k) {
            const long pi = psub ? ((long)n * (H >> 1) + (yy >> 1)) * (W >> 1) + (xw >> 1) : i;
            const float scale = params[pi * 2], mean = params[pi * 2 + 1];
            const float q = __builtin_rintf(x[i] - mean);
            so_far[i] = q + mean;
            acc += (double)laplace_bits(q, scale);
        } else if (k == 0) {
            so_far[i] = 0.0f;
        }
    }
    wave_accumulate(acc, bits + n);
}

// LL subband (pWave.py:255-263): ll_hat = round(ll), bits of the UNROUNDED residual ll_hat - mean
__global__ void ll_estimate_kernel(const float *__restrict__ ll_hat, const float *__restrict__ params, long HW,
                                   double *bits) {
    const int n = blockIdx.y;
    double acc = 0.0;
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < HW; j += (long)gridDim.x * blockDim.x) {
        const long i = n * HW + j;
        acc += (double)laplace_bits(ll_hat[i] - params[i * 2 + 1], params[i * 2]);
    }
    wave_accumulate(acc, bits + n);
}

// factorized prior of the MV hyper latent (entropy_models.py:72-77,114-122): z NHWC [HW][C]; consts [11][C] =
// softplus(h1..4), b1..4, tanh(a1..3)
__device__ __forceinline__ float bitparm_cdf(float x, const float *__restrict__ k, int c, int C) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        x = x * k[i * C + c] + k[(4 + i) * C + c];
        x = x + pm::tanhf_(x) * k[(8 + i) * C + c];
    }
    x = x * k[3 * C + c] + k[7 * C + c];
    return pm::sigmoidf_(x);
}

__global__ void z_estimate_kernel(const float *__restrict__ z, float *z_hat, const float *__restrict__ consts, long total,
                                  int C, double *bits) {
    double acc = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const float q = __builtin_rintf(z[i]);
        z_hat[i] = q;
        acc += (double)neglog2(bitparm_cdf(q + 0.5f, consts, c, C) - bitparm_cdf(q - 0.5f, consts, c, C));
    }
    wave_accumulate(acc, bits);
}

__constant__ int MVE_PERM[4][4] = {{0, 1, 2, 3}, {3, 2, 1, 0}, {2, 3, 0, 1}, {1, 0, 3, 2}};

// MV four-part coder, step t (four_part_prior.py:89-195): y_hat contribution and bits of the elements coded in step t
__global__ void mv_fourpart_estimate_kernel(const float *__restrict__ y, const float *__restrict__ common,
                                            const float *__restrict__ sp, float *so_far, int H, int W, int t,
                                            double *bits) {
    const long total = (long)H * W * 64;
    double acc = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i & 63);
        const long p = i >> 6;
        const int xw = (int)(p % W), yy = (int)(p / W);
        const int cls = (yy & 1) * 2 + (xw & 1);
        const int g = c >> 4, cc = c & 15;
        if (MVE_PERM[t][g] == cls) {
            float qs = common[p * 192 + c];
            qs = qs > 0.5f ? qs : 0.5f;
            const float q_enc = 1.0f / qs;
            float scale, mean;
            if (t == 0) { scale = common[p * 192 + 64 + c]; mean = common[p * 192 + 128 + c]; }
            else { scale = sp[p * 128 + g * 16 + cc]; mean = sp[p * 128 + 64 + g * 16 + cc]; }
            const float q = __builtin_rintf(y[i] * q_enc - mean);
            so_far[i] = q + mean;
            acc += (double)laplace_bits(q, scale);
        } else if (t == 0) {
            so_far[i] = 0.0f;
        }
    }
    wave_accumulate(acc, bits);
}

__global__ void sqdiff_sum_kernel(const float *__restrict__ a, const float *__restrict__ b, long n, double *out) {
    double acc = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float d = a[i] - b[i];
        acc += (double)(d * d);
    }
    wave_accumulate(acc, out);
}

}  // namespace

extern "C" int pmctf_fourstep_estimate_f32(const float *x, const float *params, float *so_far, int N, int H, int W, int k,
                                           int params_sub, double *bits_per_plane, void *stream) {
    if (!x || !params || !so_far || !bits_per_plane || N <= 0 || N > 65535 || H <= 0 || W <= 0 || k < 0 || k > 3 ||
        (params_sub && ((H | W) & 1)))
        return PMCTF_EINVAL;
    PM_LAUNCH(fourstep_estimate_kernel, dim3(grid_for((long)H * W), N), dim3(256), 0, (hipStream_t)stream, x, params,
              so_far, H, W, k, params_sub, bits_per_plane);
    return launch_ok();
}

extern "C" int pmctf_ll_estimate_f32(const float *ll_hat, const float *params, int N, int64_t HW, double *bits_per_plane,
                                     void *stream) {
    if (!ll_hat || !params || !bits_per_plane || N <= 0 || N > 65535 || HW <= 0) return PMCTF_EINVAL;
    PM_LAUNCH(ll_estimate_kernel, dim3(grid_for(HW), N), dim3(256), 0, (hipStream_t)stream, ll_hat, params, (long)HW,
              bits_per_plane);
    return launch_ok();
}

extern "C" int pmctf_z_estimate_f32(const float *z, float *z_hat, const float *consts, int64_t HW, int C, double *bits,
                                    void *stream) {
    if (!z || !z_hat || !consts || !bits || HW <= 0 || C <= 0) return PMCTF_EINVAL;
    PM_LAUNCH(z_estimate_kernel, dim3(grid_for(HW * C)), dim3(256), 0, (hipStream_t)stream, z, z_hat, consts,
              (long)HW * C, C, bits);
    return launch_ok();
}

extern "C" int pmctf_mv_fourpart_estimate_f32(const float *y, const float *common, const float *sp, float *so_far, int H,
                                              int W, int t, double *bits, void *stream) {
    if (!y || !common || !so_far || !bits || H <= 0 || W <= 0 || t < 0 || t > 3 || (t > 0 && !sp)) return PMCTF_EINVAL;
    PM_LAUNCH(mv_fourpart_estimate_kernel, dim3(grid_for((long)H * W * 64)), dim3(256), 0, (hipStream_t)stream, y, common,
              sp, so_far, H, W, t, bits);
    return launch_ok();
}

extern "C" int pmctf_sqdiff_sum_f32(const float *a, const float *b, int64_t n, double *sum, void *stream) {
    if (!a || !b || !sum || n <= 0) return PMCTF_EINVAL;
    PM_LAUNCH(sqdiff_sum_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, (long)n, sum);
    return launch_ok();
}
