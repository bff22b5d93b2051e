// Latency of a dependent v_fma_f32 chain on gfx950 (one wave per SIMD): what bounds the sequential LL decoder's
// 640-term chains.  Also two and four interleaved independent chains, and a chain fed by LDS broadcast reads.
// build: hipcc -O3 --offload-arch=gfx950 tools/fma_chain.hip -o /tmp/fma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH>
__global__ void chain(float *out, const float *in, long long *cyc, int iters) {
    __shared__ float lds[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = in[i];
    __syncthreads();
    float a[CH];
    for (int c = 0; c < CH; ++c) a[c] = in[threadIdx.x + c];
    float w = in[threadIdx.x + 7];
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 64; ++k)
#pragma unroll
            for (int c = 0; c < CH; ++c) a[c] = __builtin_fmaf(a[c], w, lds[(k * 4 + c) & 1023]);
    }
    long long t1 = clock64();
    float s = 0;
    for (int c = 0; c < CH; ++c) s += a[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int CH>
void run(const char *name, int threads) {
    float *out, *in; long long *cyc;
    hipMalloc(&out, 4096 * 4); hipMalloc(&in, 4096 * 4); hipMalloc(&cyc, 64);
    hipMemset(in, 0, 4096 * 4);
    const int iters = 2000;
    chain<CH><<<1, threads>>>(out, in, cyc, iters);
    chain<CH><<<1, threads>>>(out, in, cyc, iters);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-44s %6.2f clock64 ticks per fma (chain step %6.2f)\n", name, (double)c / (iters * 64.0 * CH), (double)c / (iters * 64.0));
}
int main() {
    run<1>("1 chain, 128 threads (2 waves, 2 SIMDs)", 128);
    run<2>("2 interleaved chains, 128 threads", 128);
    run<4>("4 interleaved chains, 128 threads", 128);
    run<1>("1 chain, 256 threads (1 wave per SIMD)", 256);
    run<1>("1 chain, 512 threads (2 waves per SIMD)", 512);
    return 0;
}
