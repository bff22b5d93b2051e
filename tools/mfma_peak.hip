// Sustained f32 matrix throughput of the whole chip with nothing but MFMAs (8 / 4 independent accumulators per wave,
// 2048 workgroups): v_mfma_f32_16x16x4_f32 148-151 TFLOP/s, v_mfma_f32_32x32x2_f32 156 TFLOP/s on MI355X (nominal 157.3).
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k16(float *out, float a, float b, int iters) {
    f32x4 acc[8];
    for (int c = 0; c < 8; ++c) acc[c] = {a, b, a, b};
    const float va = a + (threadIdx.x & 7) * 0.001f, vb = b + (threadIdx.x & 3) * 0.002f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(va, vb, acc[c], 0, 0, 0);
    float s = 0; for (int c = 0; c < 8; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k32(float *out, float a, float b, int iters) {
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = a + i;
    const float va = a + (threadIdx.x & 7) * 0.001f, vb = b + (threadIdx.x & 3) * 0.002f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(va, vb, acc[c], 0, 0, 0);
    float s = 0; for (int c = 0; c < 4; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float *o; (void)hipMalloc(&o, 4096 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int wgs = 2048, iters = 40000;
    for (int rep = 0; rep < 3; ++rep) {
        float ms;
        (void)hipEventRecord(e0, 0); k16<<<wgs, 256>>>(o, 0.5f, 0.25f, iters); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("16x16x4 f32: %.1f ms  %.1f TFLOP/s\n", ms, (double)wgs * 4 * iters * 8 * 2048.0 / (ms * 1e-3) / 1e12);
        (void)hipEventRecord(e0, 0); k32<<<wgs, 256>>>(o, 0.5f, 0.25f, iters); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("32x32x2 f32: %.1f ms  %.1f TFLOP/s\n", ms, (double)wgs * 4 * iters * 4 * 4096.0 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
