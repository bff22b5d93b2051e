// Is v_mfma_f32_32x32x2_f32 the same sequential fmaf chain over k as v_mfma_f32_16x16x4_f32?  Yes: 0 of 1024 outputs differ
// from fmaf(a[k][m], b[k][n], acc) applied for k = 0..143 in order (MI355X).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/mfma32_exact.hip -o /tmp/mfma32_exact && /tmp/mfma32_exact
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
// one wave: D(32x32) = C + A(32x2) B(2x32), repeated over KS k-steps (chain)
__global__ void k(const float *A, const float *B, const float *C, float *D, int KS) {
    const int lane = threadIdx.x;
    f32x16 acc;
    for (int i = 0; i < 16; ++i) {
        const int row = 8 * (i / 4) + (lane / 32) * 4 + (i % 4), col = lane % 32;
        acc[i] = C[row * 32 + col];
    }
    for (int ks = 0; ks < KS; ++ks) {
        const float a = A[(ks * 2 + lane / 32) * 32 + lane % 32];   // A[k][m]
        const float b = B[(ks * 2 + lane / 32) * 32 + lane % 32];   // B[k][n]
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) {
        const int row = 8 * (i / 4) + (lane / 32) * 4 + (i % 4), col = lane % 32;
        D[row * 32 + col] = acc[i];
    }
}
int main() {
    const int KS = 72;
    std::vector<float> A(KS * 2 * 32), B(KS * 2 * 32), C(1024), D(1024);
    srand(1);
    auto rnd = [] { return (float)(rand() % 20001 - 10000) / 3000.0f * ((rand() & 7) == 0 ? 1e-3f : 1.0f); };
    for (auto &v : A) v = rnd();
    for (auto &v : B) v = rnd();
    for (auto &v : C) v = rnd();
    float *dA, *dB, *dC, *dD;
    (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&dC, 4096); (void)hipMalloc(&dD, 4096);
    (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dC, C.data(), 4096, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dC, dD, KS); (void)hipDeviceSynchronize();
    (void)hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) {
        float acc = C[m * 32 + n];
        for (int kk = 0; kk < KS * 2; ++kk) acc = fmaf(A[kk * 32 + m], B[kk * 32 + n], acc);
        if (acc != D[m * 32 + n]) { if (bad < 5) printf("mismatch (%d,%d): %a vs %a\n", m, n, acc, D[m * 32 + n]); ++bad; }
    }
    printf("32x32x2 f32 vs sequential fmaf chain over %d k: %d of 1024 differ\n", KS * 2, bad);
    return 0;
}
