// pu_fused.hip — the PredictUpdate CNN of the learned lifting steps as ONE kernel.
//
// Replaces, per call, the chain the reference runs as separate torch ops (pMCTF/layers/lifting_1d.py:25-49 `PredictUpdate`,
// :103-145 the iWave lifting branch, pMCTF/layers/wavelet_transform_temporal_mctf.py:27-45 the temporal predict / update
// filters):
//     c1 = conv3x3(x; 1->16)            t = conv3x3(tanh(c1); 16->16)
//     t  = conv3x3(tanh(t); 16->16)     pu = conv3x3(c1 + t; 16->1)          (zero padding everywhere)
//   mode 0 (temporal):  out = (x + pu * 0.1) * c
//   mode 1 (iWave):     skip = reflect-pad 3x1 conv along H of x (+bias);  pu = PU(skip / 256)
//                       out = other +/- (skip + (pu * 256) * 0.1)
// The unfused stack moves >= 128 B per pixel through HBM for 9.2 kFLOP; fused, a pixel costs 8-12 B and the two 16->16
// layers (4 608 MAC per pixel) bound the kernel on the f32 matrix pipe.
//
// Workgroup = 256 threads = one 8x32 output tile of one plane.  Everything between the input tile and the output tile
// lives in LDS (74.5 KB, two workgroups per CU so that the vector-ALU phases of one overlap the matrix phases of the
// other):   in[16x40] -> (VALU) tanh(c1) on 14x38 -> (MFMA) tanh(conv2) on 12x36 -> (MFMA) conv3 + c1 on 10x34 ->
// (VALU) conv4 on 8x32.  Values outside the image are stored as 0: every layer zero-pads its own input.
// PM-F32 order, identical to the separate kernels (and to oracle/c/pm_ops.c): ky, kx, ci ascending fmaf, the chain starting
// at the bias (rule 0) or at zero with the bias added last (rule 1 — every layer here has at most 16 input channels, one
// block; `rule` for the four CNN layers, `skip_rule` for the 3x1 lifting filter, which ATen evaluates through another path
// on small planes); v_mfma_f32_16x16x4_f32 is that chain bit for bit.  The residual c1 of layer 3 is re-evaluated from the input tile
// with the same nine fmaf, so no second copy of c1 is kept.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pm_device_math.h"
#include "launch.h"
#include "../../include/pmctf_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TH = 8, TW = 32;
constexpr int IH = TH + 8, IW = TW + 8;          // input region (halo 4)
constexpr int R1H = TH + 6, R1W = TW + 6;        // c1 / tanh(c1)
constexpr int R2H = TH + 4, R2W = TW + 4;        // tanh(conv2)
constexpr int R3H = TH + 2, R3W = TW + 2;        // conv3 + c1
constexpr int CP = 18;                           // LDS words per pixel of a 16-channel map (conflict-free, 8-B aligned)
constexpr int N1 = R1H * R1W, N2 = R2H * R2W, N3 = R3H * R3W;
constexpr int LDS_IN = 0, LDS_SK = IH * IW, LDS_A1 = 2 * IH * IW, LDS_A2 = LDS_A1 + N1 * CP;
constexpr int LDS_FLOATS = LDS_A2 + N2 * CP;

struct PuArgs {
    const float *x, *other;
    float *out;
    const float *w1, *b1;        // (16,1,3,3) OIHW + (16)
    const float *w2p, *b2p;      // packed MFMA fragments [tap][lane][ks] (pmctf_conv2d_pack_weights) + padded bias
    const float *w3p, *b3p;
    const float *w4, *b4;        // (1,16,3,3) OIHW + (1)
    int N, H, W, tiles_x, tiles_y, mode;
    float c, sign, lw0, lw1, lw2, lb;
    int rule, skip_rule;
};

// one 16->16 3x3 layer on the matrix cores: src (row stride SW pixels, CP words per pixel) -> D fragments per segment.
// A wave walks its 16-pixel segments two at a time: two independent accumulator chains keep the matrix pipe issuing
// back to back (a single chain waits 40 cycles per dependent v_mfma_f32_16x16x4_f32) and cover the LDS read latency.
template <typename Epilogue>
__device__ __forceinline__ void mfma_layer(const float *src, int SW, int RW, int npix, const float *wp, const float *bp,
                                           int wave, int lane, int rule, Epilogue epi) {
    f32x4 af[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) af[t] = *(const f32x4 *)(wp + t * 256 + lane * 4);
    const f32x4 bias = *(const f32x4 *)(bp + 4 * (lane >> 4));
    const int nseg = (npix + 15) >> 4;
    for (int s = wave; s < nseg; s += 8) {
        const bool two = s + 4 < nseg;                      // wave-uniform
        int idx0 = s * 16 + (lane & 15), idx1 = (s + 4) * 16 + (lane & 15);
        const bool live0 = idx0 < npix, live1 = two && idx1 < npix;
        if (!live0) idx0 = npix - 1;
        if (!live1) idx1 = npix - 1;
        const int r0 = idx0 / RW, c0 = idx0 - r0 * RW;
        const int r1 = idx1 / RW, c1 = idx1 - r1 * RW;
        const float *p0 = src + (r0 * SW + c0) * CP + (lane >> 4);
        const float *p1 = src + (r1 * SW + c1) * CP + (lane >> 4);
        const f32x4 start = rule ? f32x4{0.f, 0.f, 0.f, 0.f} : bias;
        f32x4 acc0 = start, acc1 = start;
        if (two) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int toff = ((t / 3) * SW + (t % 3)) * CP;
                float b0[4], b1[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) { b0[ks] = p0[toff + ks * 4]; b1[ks] = p1[toff + ks * 4]; }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[t][ks], b0[ks], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[t][ks], b1[ks], acc1, 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int toff = ((t / 3) * SW + (t % 3)) * CP;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[t][ks], p0[toff + ks * 4], acc0, 0, 0, 0);
            }
        }
        if (rule) { acc0 = acc0 + bias; acc1 = acc1 + bias; }
        if (live0) epi(idx0, r0, c0, acc0);
        if (live1) epi(idx1, r1, c1, acc1);
    }
}

// per-phase cycle counters for tools/pu_prof.hip (compiled in only there)
#ifdef PMCTF_PU_PROFILE
static size_t pu_extra_lds = 0;
__device__ unsigned long long pu_prof[8];
#define PU_STAMP(i) do { if (tid == 0) { const long long t_ = clock64(); atomicAdd(&pu_prof[i], (unsigned long long)(t_ - t_prev_)); t_prev_ = t_; } } while (0)
#else
#define PU_STAMP(i) do { } while (0)
#endif

__global__ __launch_bounds__(256, 2) void pu_fused_kernel(PuArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef PMCTF_PU_PROFILE
    long long t_prev_ = clock64();
#endif
    __shared__ uint4 tanh_tab[pm::TANH_LDS_UINT4];
    pm::tanh_rows_to_lds(tanh_tab, threadIdx.x, 256);          // visible after the barrier behind phase 0
    float *in = lds + LDS_IN, *sk = lds + LDS_SK, *A1 = lds + LDS_A1, *A2 = lds + LDS_A2;
    float *A3 = A1;                                   // layer-3 output reuses the (dead) tanh(c1) buffer
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b = blockIdx.x;
    const int tx = b % a.tiles_x; b /= a.tiles_x;
    const int ty = b % a.tiles_y;
    const int n = b / a.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    const long plane = (long)n * a.H * a.W;
    const float *xp = a.x + plane;

    // ---- P0: input tile (PU input; zero outside the image).  mode 1: skip = 3x1 reflect conv, PU input = skip / 256
    for (int e = tid; e < IH * IW; e += 256) {
        const int r = e / IW, c = e - r * IW;
        const int gy = y0 - 4 + r, gx = x0 - 4 + c;
        float v = 0.0f, s = 0.0f;
        if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
            if (a.mode == 0) {
                v = xp[(long)gy * a.W + gx];
            } else {
                const int ym = gy == 0 ? 1 : gy - 1;
                const int yp = gy == a.H - 1 ? a.H - 2 : gy + 1;
                float acc = a.skip_rule ? 0.0f : a.lb;
                acc = __builtin_fmaf(xp[(long)ym * a.W + gx], a.lw0, acc);
                acc = __builtin_fmaf(xp[(long)gy * a.W + gx], a.lw1, acc);
                acc = __builtin_fmaf(xp[(long)yp * a.W + gx], a.lw2, acc);
                if (a.skip_rule) acc = acc + a.lb;
                s = acc;
                v = acc / 256.0f;
            }
        }
        in[e] = v;
        sk[e] = s;
    }
    __syncthreads();
    PU_STAMP(0);

    // ---- P1 (vector ALU): tanh(conv1) on the 14x38 region.  A thread owns two pixels; the 16 couts go in four groups of
    // four so that a group's 36 weights + 4 biases are wave-uniform scalars while all pixels use them.  The region has
    // 532 = 2 * 256 + 20 pixels: a third pass over 20 live lanes would cost wave 0 a whole pass (and every other wave
    // the wait for it), so the 20 left-over pixels are shared out by cout group instead: wave w computes group w of
    // all of them — a quarter of a pass per wave.
    {
        constexpr int PPT = N1 / 256, LEFT = N1 - PPT * 256;
        static_assert(LEFT >= 0 && LEFT <= 64, "left-over pixels of the conv1 region must fit one wave");
        float iv[PPT + 1][9];
        bool ok[PPT + 1];
#pragma unroll
        for (int j = 0; j <= PPT; ++j) {
            const int idx = j < PPT ? tid + 256 * j : PPT * 256 + lane;
            const bool in_region = j < PPT || lane < LEFT;
            const int r = idx / R1W, c = idx - r * R1W;
            const int gy = y0 - 3 + r, gx = x0 - 3 + c;
            ok[j] = in_region && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
#pragma unroll
            for (int t = 0; t < 9; ++t) iv[j][t] = in_region ? in[(r + t / 3) * IW + c + t % 3] : 0.0f;
        }
        auto group = [&](int q, int j, int idx) {
            float wq[4][9], bq[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bq[i] = a.b1[q * 4 + i];
#pragma unroll
                for (int t = 0; t < 9; ++t) wq[i][t] = a.w1[(q * 4 + i) * 9 + t];
            }
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float acc = a.rule ? 0.0f : bq[i];
#pragma unroll
                for (int t = 0; t < 9; ++t) acc = __builtin_fmaf(iv[j][t], wq[i][t], acc);
                if (a.rule) acc = acc + bq[i];
                v[i] = ok[j] ? pm::tanhf_rows(acc, tanh_tab) : 0.0f;
            }
            float2 *dst = (float2 *)(A1 + idx * CP + 4 * q);
            dst[0] = make_float2(v[0], v[1]);
            dst[1] = make_float2(v[2], v[3]);
        };
#pragma unroll 1
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int j = 0; j < PPT; ++j) group(q, j, tid + 256 * j);
        }
        if (LEFT > 0 && lane < LEFT) group(wave, PPT, PPT * 256 + lane);
    }
    __syncthreads();
    PU_STAMP(1);

    // ---- P2 (matrix cores): tanh(conv2(tanh c1)) on the 12x36 region
    mfma_layer(A1, R1W, R2W, N2, a.w2p, a.b2p, wave, lane, a.rule, [&](int idx, int r, int c, f32x4 acc) {
        const int gy = y0 - 2 + r, gx = x0 - 2 + c;
        const bool inside = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = inside ? pm::tanhf_rows(acc[i], tanh_tab) : 0.0f;
        float2 *dst = (float2 *)(A2 + idx * CP + 4 * (lane >> 4));
        dst[0] = make_float2(v[0], v[1]);
        dst[1] = make_float2(v[2], v[3]);
    });
    __syncthreads();
    PU_STAMP(2);

    // ---- P3 (matrix cores): conv3(.) + c1 on the 10x34 region; c1 re-evaluated from the input tile (same nine fmaf)
    float w1g[4][9], b1g[4];                  // conv1 weights of this lane's four couts (4*(lane>>4) ..)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int co = 4 * (lane >> 4) + i;
        b1g[i] = a.b1[co];
#pragma unroll
        for (int t = 0; t < 9; ++t) w1g[i][t] = a.w1[co * 9 + t];
    }
    mfma_layer(A2, R2W, R3W, N3, a.w3p, a.b3p, wave, lane, a.rule, [&](int idx, int r, int c, f32x4 acc) {
        const int gy = y0 - 1 + r, gx = x0 - 1 + c;
        const bool inside = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (inside) {
            float iv[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) iv[t] = in[(r + 2 + t / 3) * IW + c + 2 + t % 3];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float c1 = a.rule ? 0.0f : b1g[i];
#pragma unroll
                for (int t = 0; t < 9; ++t) c1 = __builtin_fmaf(iv[t], w1g[i][t], c1);
                if (a.rule) c1 = c1 + b1g[i];
                v[i] = acc[i] + c1;
            }
        }
        float2 *dst = (float2 *)(A3 + idx * CP + 4 * (lane >> 4));
        dst[0] = make_float2(v[0], v[1]);
        dst[1] = make_float2(v[2], v[3]);
    });
    __syncthreads();
    PU_STAMP(3);

    // ---- P4 (vector ALU): conv4 (16 -> 1) on the 8x32 tile + the lifting arithmetic; thread = output pixel
    {
        const int r = tid >> 5, c = tid & 31;
        const int gy = y0 + r, gx = x0 + c;
        if (gy < a.H && gx < a.W) {
            float acc = a.rule ? 0.0f : a.b4[0];
#pragma unroll 1
            for (int t = 0; t < 9; ++t) {                       // one tap's 16 weights at a time stay scalar
                const float2 *p = (const float2 *)(A3 + ((r + t / 3) * R3W + c + t % 3) * CP);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float2 v = p[q];
                    acc = __builtin_fmaf(v.x, a.w4[(2 * q) * 9 + t], acc);
                    acc = __builtin_fmaf(v.y, a.w4[(2 * q + 1) * 9 + t], acc);
                }
            }
            if (a.rule) acc = acc + a.b4[0];
            const long o = plane + (long)gy * a.W + gx;
            const int e = (r + 4) * IW + c + 4;
            float res;
            if (a.mode == 0) {
                res = (in[e] + acc * 0.1f) * a.c;                          // EW_ADD_MULS_MULS(x, pu, 0.1, c)
            } else {
                const float br = sk[e] + (acc * 256.0f) * 0.1f;            // EW_ADD_MULS2(skip, pu, 256, 0.1)
                res = a.sign > 0.0f ? a.other[o] + br : a.other[o] - br;   // EW_ADD / EW_SUB
            }
            a.out[o] = res;
        }
    }
    PU_STAMP(4);
}

}  // namespace

extern "C" int pmctf_predict_update_fused_f32(const float *x, const float *other, float *out, const float *w1,
                                              const float *b1, const float *w2_packed, const float *b2_packed,
                                              const float *w3_packed, const float *b3_packed, const float *w4,
                                              const float *b4, int N, int H, int W, int mode, float c, float sign,
                                              float lw0, float lw1, float lw2, float lbias, int sum_rule,
                                              int skip_sum_rule, void *stream) {
    if (!x || !out || !w1 || !b1 || !w2_packed || !b2_packed || !w3_packed || !b3_packed || !w4 || !b4 || N <= 0 ||
        H <= 0 || W <= 0 || (mode != 0 && mode != 1) || (mode == 1 && (!other || H < 2)) ||
        (sum_rule != 0 && sum_rule != 1) || (skip_sum_rule != 0 && skip_sum_rule != 1))
        return PMCTF_EINVAL;
    PuArgs a;
    a.x = x; a.other = other; a.out = out; a.w1 = w1; a.b1 = b1; a.w2p = w2_packed; a.b2p = b2_packed;
    a.w3p = w3_packed; a.b3p = b3_packed; a.w4 = w4; a.b4 = b4;
    a.N = N; a.H = H; a.W = W; a.mode = mode; a.c = c; a.sign = sign; a.lw0 = lw0; a.lw1 = lw1; a.lw2 = lw2; a.lb = lbias;
    a.rule = sum_rule; a.skip_rule = skip_sum_rule;
    a.tiles_x = (W + TW - 1) / TW;
    a.tiles_y = (H + TH - 1) / TH;
    const long blocks = (long)a.tiles_x * a.tiles_y * N;
    if (blocks > 0x7fffffffL) return PMCTF_EINVAL;
    static bool attr = [] {
        (void)hipFuncSetAttribute((const void *)pu_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true;
    }();
    (void)attr;
#ifdef PMCTF_PU_PROFILE
    PM_LAUNCH(pu_fused_kernel, dim3((unsigned)blocks), dim3(256), LDS_FLOATS * sizeof(float) + pu_extra_lds, (hipStream_t)stream, a);
#else
    PM_LAUNCH(pu_fused_kernel, dim3((unsigned)blocks), dim3(256), LDS_FLOATS * sizeof(float), (hipStream_t)stream, a);
#endif
    return pm_launch_status();
}
