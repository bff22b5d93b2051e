// Do f32 MFMA and vector-ALU instructions overlap on a gfx950 SIMD?  (They do not.)
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/mfma_valu_overlap.hip -o /tmp/mfma_valu_overlap && /tmp/mfma_valu_overlap
// One workgroup of W waves; every wave runs 256 x CH v_mfma_f32_16x16x4_f32 (CH independent accumulator chains) with V
// independent v_fma_f32 after each; reported: cycles (s_memtime) from the first wave's start to the last wave's end per
// MFMA of one wave.  Measured on MI355X:
//   chains 2, 0 VALU, 4 waves (1 per SIMD): 34.1      chains 2, 0 VALU, 8 waves (2 per SIMD): 65.6  (the pipe is full)
//   chains 2, 4 VALU: 62.4   6 VALU: 72.4   8 VALU: 83.7   16 VALU: 119.5      (4 waves: + 5.3 cycles per VALU instruction)
//   chains 2, 6 VALU, 8 waves: 110.7   16 VALU, 8 waves: 212.4                  (= 2 x (32 + 4.6 V): no overlap across waves)
// i.e. time = MFMA cycles + VALU cycles, within a wave and between the waves of a SIMD.  A single dependent chain issues
// a 16x16x4 f32 MFMA every 35 cycles (32 = 8 passes), so one chain nearly saturates the pipe.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CH, int VALU>
__global__ void k(float *out, long long *cyc, float a, float b) {
    f32x4 acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = {a, b, a, b};
    float v = a + threadIdx.x;
    float w[8]; for (int j = 0; j < 8; ++j) w[j] = v + j;
    const float ca = a * 3.0f + threadIdx.x, cb = b * 5.0f + threadIdx.x;
    long long t0 = clock64();
    for (int it = 0; it < 256; ++it) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < VALU; ++j) w[j & 7] = __builtin_fmaf(w[j & 7], ca, cb);
            for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(w[j]));
        }
    }
    long long t1 = clock64();
    float s = v; for (int j = 0; j < 8; ++j) s += w[j];
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[threadIdx.x + blockIdx.x * blockDim.x] = s;
    if ((threadIdx.x & 63) == 0) { atomicMin((unsigned long long *)cyc, (unsigned long long)t0); atomicMax((unsigned long long *)cyc + 1, (unsigned long long)t1); }
}
template <int CH, int VALU> void run(int waves) {
    float *o; long long *c; hipMalloc(&o, 4096 * 4); hipMalloc(&c, 16);
    long long init[2] = {(long long)0x7fffffffffffffffLL, 0};
    k<CH, VALU><<<1, 64 * waves>>>(o, c, 0.5f, 0.25f); hipDeviceSynchronize();
    hipMemcpy(c, init, 16, hipMemcpyHostToDevice);
    k<CH, VALU><<<1, 64 * waves>>>(o, c, 0.5f, 0.25f); hipDeviceSynchronize();
    long long hh[2]; hipMemcpy(hh, c, 16, hipMemcpyDeviceToHost); long long h = hh[1] - hh[0];
    printf("chains %d  valu/mfma %d  waves/WG %d: %.1f cycles per MFMA\n", CH, VALU, waves, (double)h / (256.0 * CH));
}
int main() {
    run<2, 0>(4); run<2, 4>(4); run<2, 6>(4); run<2, 8>(4); run<2, 16>(4); run<2, 6>(8); run<2, 8>(8); run<2, 16>(8); run<4, 6>(4);
    return 0;
}
