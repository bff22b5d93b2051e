// Per-phase cycle counts of the fused PredictUpdate kernel (debug tool, not part of the product library).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 tools/pu_prof.hip -o learned-pmctf_amd/lib/pu_prof
//   learned-pmctf_amd/lib/pu_prof <planes> [extra LDS bytes: 30000 forces one workgroup per CU]
// Measured on MI355X (1152x1920 plane): P0 5.5k, P1 (tanh(conv1), vector ALU) 28.8k, P2 (MFMA + tanh) 19.9k, P3 (MFMA + c1)
// 12.7k, P4 (16->1) 5.9k cycles per workgroup with two workgroups per CU; one per CU: P1 15.6k, the MFMA phases
// unchanged - vector and matrix work of the two workgroups add up on a SIMD instead of overlapping.
#define PMCTF_PU_PROFILE 1
#include "../learned-pmctf_amd/csrc/pu_fused.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 1, H = 1152, W = 1920;
    pu_extra_lds = argc > 2 ? (size_t)atoi(argv[2]) : 0;
    const size_t px = (size_t)N * H * W;
    std::vector<float> hx(px);
    for (size_t i = 0; i < px; ++i) hx[i] = (float)((i * 2654435761u) % 1000) / 500.0f - 1.0f;
    float *x, *o, *w;
    hipMalloc(&x, px * 4); hipMalloc(&o, px * 4); hipMalloc(&w, 65536 * 4);
    hipMemcpy(x, hx.data(), px * 4, hipMemcpyHostToDevice);
    std::vector<float> hw(65536);
    for (int i = 0; i < 65536; ++i) hw[i] = (float)((i * 40503u) % 200) / 1000.0f - 0.1f;
    hipMemcpy(w, hw.data(), 65536 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        unsigned long long z[8] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(pu_prof), z, sizeof(z));
        hipEventRecord(e0, 0);
        int rc = pmctf_predict_update_fused_f32(x, nullptr, o, w, w + 200, w + 1024, w + 4096, w + 8192, w + 12000, w + 16000,
                                                w + 17000, N, H, W, 0, 0.5f, 1.0f, 0, 0, 0, 0, nullptr);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpyFromSymbol(z, HIP_SYMBOL(pu_prof), sizeof(z));
        const double blocks = (double)N * ((H + 7) / 8) * ((W + 31) / 32);
        printf("rc %d  %.1f us   cycles per workgroup: P0 %.0f  P1 %.0f  P2 %.0f  P3 %.0f  P4 %.0f  total %.0f\n", rc, ms * 1e3,
               z[0] / blocks, z[1] / blocks, z[2] / blocks, z[3] / blocks, z[4] / blocks, (z[0] + z[1] + z[2] + z[3] + z[4]) / blocks);
    }
    return 0;
}
