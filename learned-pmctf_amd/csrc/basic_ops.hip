// basic_ops.hip — bandwidth-bound primitives of the pMCTF encode path (vector ALU,
// coalesced NHWC / planar accesses).  Each kernel restates one torch functional op of
// the reference in PM-F32 arithmetic (same operation order as oracle/c/pm_ops.c).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include "pm_device_math.h"
#include "launch.h"
#include "../../include/pmctf_hip.h"

namespace {

inline int launch_ok() { return pm_launch_status(); }
inline unsigned nblocks(long n, int bs = 256) {
    long b = (n + bs - 1) / bs;
    return (unsigned)(b < 1 ? 1 : b);
}

// ---------------------------------------------------------------------------------
// direct conv for Cin <= 4: one thread per output element, cout fastest (NHWC store).
// acc = bias; for ky: for kx: for ci: fmaf   (single 16-channel chunk of the spec order)
__global__ void conv_smallcin_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                     const float *__restrict__ bias, const float *res1, const float *res2,
                                     float *y, int N, int H, int W, int Cin, int Cout, int KH, int KW, int S,
                                     int ph, int pw, int Ho, int Wo, int act, float slope, int rule) {
    extern __shared__ float wl[];  // [tap][ci][co]
    const int taps = KH * KW;
    for (int i = threadIdx.x; i < Cout * Cin * taps; i += blockDim.x) {
        const int co = i / (Cin * taps), r = i - co * Cin * taps;
        const int ci = r / taps, t = r - ci * taps;
        wl[(t * Cin + ci) * Cout + co] = w[i];
    }
    __syncthreads();
    const long total = (long)N * Ho * Wo * Cout;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int co = (int)(pm_mod(idx, Cout));
        long p = pm_div(idx, Cout);
        const int ox = (int)(pm_mod(p, Wo)); p = pm_div(p, Wo);
        const int oy = (int)(pm_mod(p, Ho));
        const int n = (int)(pm_div(p, Ho));
        const float bv = bias ? bias[co] : 0.0f;
        float acc = rule ? 0.0f : bv;            // rule 1 (one block, Cin <= 4): chain from zero, bias last
        if (rule == PMCTF_SUM_GEMV_3X3) {        // Cin = 1, 3x3 (checked by the launcher): the order of include/pmctf_hip.h
            float p[9];
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) {
                    const int iy = oy * S + ky - ph, ix = ox * S + kx - pw;
                    p[ky * 3 + kx] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? x[((long)n * H + iy) * W + ix] : 0.0f;
                }
            const float *wk = wl + co;           // [tap][ci = 0][co]
            const float E = __builtin_fmaf(p[4], wk[4 * Cout], __builtin_fmaf(p[6], wk[6 * Cout], bv));
            const float O = __builtin_fmaf(p[5], wk[5 * Cout], p[7] * wk[7 * Cout]);
            const float Ae = __builtin_fmaf(p[0], wk[0], p[2] * wk[2 * Cout]);
            const float Ao = __builtin_fmaf(p[1], wk[1 * Cout], p[3] * wk[3 * Cout]);
            acc = __builtin_fmaf(p[8], wk[8 * Cout], (E + O) + (Ae + Ao));
            float v = pm::apply_act(acc, act, slope);
            if (res1) v = v + res1[idx];
            if (res2) v = v + res2[idx];
            y[idx] = v;
            continue;
        }
        if (rule == PMCTF_SUM_GEMM) {            // im2col order: input channel outermost
            for (int ci = 0; ci < Cin; ++ci)
                for (int ky = 0; ky < KH; ++ky) {
                    const int iy = oy * S + ky - ph;
                    if (iy < 0 || iy >= H) continue;
                    for (int kx = 0; kx < KW; ++kx) {
                        const int ix = ox * S + kx - pw;
                        if (ix < 0 || ix >= W) continue;
                        acc = __builtin_fmaf(x[(((long)n * H + iy) * W + ix) * Cin + ci],
                                             wl[((ky * KW + kx) * Cin + ci) * Cout + co], acc);
                    }
                }
        } else {
            for (int ky = 0; ky < KH; ++ky) {
                const int iy = oy * S + ky - ph;
                if (iy < 0 || iy >= H) continue;
                for (int kx = 0; kx < KW; ++kx) {
                    const int ix = ox * S + kx - pw;
                    if (ix < 0 || ix >= W) continue;
                    const float *xp = x + (((long)n * H + iy) * W + ix) * Cin;
                    const float *wp = wl + ((ky * KW + kx) * Cin) * Cout + co;
                    for (int ci = 0; ci < Cin; ++ci) acc = __builtin_fmaf(xp[ci], wp[ci * Cout], acc);
                }
            }
        }
        if (rule) acc = acc + bv;
        float v = pm::apply_act(acc, act, slope);
        if (res1) v = v + res1[idx];
        if (res2) v = v + res2[idx];
        y[idx] = v;
    }
}


// Specialised direct conv for the small-Cin layers that dominate this family (3x3 with 1..3 input channels, 1x1 with 2):
// a thread owns ONE output channel (its K*K*CIN weights + bias live in registers) and walks over pixels; the
// 256-thread block covers PPP = 256/Cout pixels per pass, stores are contiguous over cout, the input taps of a pixel
// are wave-broadcast loads.  Same sum order as the generic kernel: acc = bias; for ky: for kx: for ci: fmaf.
template <int K, int CIN>
__global__ void conv_smallcin_reg_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                         const float *__restrict__ bias, const float *res1, const float *res2, float *y,
                                         int N, int H, int W, int Cout, int S, int ph, int pw, int Ho, int Wo, int act,
                                         float slope, int rule) {
    const int ppp = 256 / Cout;                 // pixels per pass (Cout <= 256)
    const int co = threadIdx.x % Cout;
    const int slot = threadIdx.x / Cout;
    if (slot >= ppp) return;
    float wr[K * K * CIN];
#pragma unroll
    for (int t = 0; t < K * K; ++t)
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) wr[t * CIN + ci] = w[(co * CIN + ci) * (K * K) + t];
    const float b = bias ? bias[co] : 0.0f;
    const int npix = N * Ho * Wo;               // < 2^31 on this path (checked by the launcher)
    for (int p = blockIdx.x * ppp + slot; p < npix; p += gridDim.x * ppp) {
        const int ox = p % Wo;
        const int r = p / Wo;
        const int oy = r % Ho;
        const int n = r / Ho;
        float acc = rule ? 0.0f : b;
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int iy = oy * S + ky - ph;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const int ix = ox * S + kx - pw;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
                    const float *xp = x + ((size_t)(n * H + iy) * W + ix) * CIN;
#pragma unroll
                    for (int ci = 0; ci < CIN; ++ci) acc = __builtin_fmaf(xp[ci], wr[(ky * K + kx) * CIN + ci], acc);
                }
            }
        }
        if (rule) acc = acc + b;
        float v = pm::apply_act(acc, act, slope);
        const size_t o = (size_t)p * Cout + co;
        if (res1) v = v + res1[o];
        if (res2) v = v + res2[o];
        y[o] = v;
    }
}


// 3x3, stride 1, pad 1, Cin <= 3, many couts (1->64/112/128, 2->112), Cout % 4 == 0: a lane owns FOUR consecutive
// couts (Q = Cout/4 lanes per pixel) of a strip of 8 consecutive output pixels; the G = 64/Q lane groups of a wave take
// adjacent strips, so the wave writes one contiguous run of G*8 pixels x Cout floats with 16-byte stores (whole cache
// lines; the layer is write-bound: 4*Cout bytes out per 4*Cin bytes in).  The 3x10xCIN input window is read once per
// strip (same address for the Q lanes of a group), no per-output index arithmetic.
// Same sum order: acc = bias; for ky: for kx: for ci: fmaf.
template <int CIN>
__global__ __launch_bounds__(256) void conv3x3_smallcin_strip_kernel(const float *__restrict__ x,
                                                                     const float *__restrict__ w,
                                                                     const float *__restrict__ bias, const float *res1,
                                                                     const float *res2, float *y, int N, int H, int W,
                                                                     int Cout, int act, float slope, int Q, int G,
                                                                     int wstrips_per_row, long total_items, int rule) {
    constexpr int P = 8;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int q = lane % Q, g = lane / Q;
    const bool active = g < G;
    const int co = 4 * q;
    float4 wr[9 * CIN];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
            wr[t * CIN + ci] = make_float4(w[((co + 0) * CIN + ci) * 9 + t], w[((co + 1) * CIN + ci) * 9 + t],
                                           w[((co + 2) * CIN + ci) * 9 + t], w[((co + 3) * CIN + ci) * 9 + t]);
    const float4 b = bias ? *(const float4 *)(bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (long item = (long)blockIdx.x * 4 + wave; item < total_items; item += (long)gridDim.x * 4) {
        const int ws = (int)(pm_mod(item, wstrips_per_row));
        const long r = pm_div(item, wstrips_per_row);
        const int oy = (int)(pm_mod(r, H));
        const int n = (int)(pm_div(r, H));
        const int sx = (ws * G + g) * P;
        if (!active || sx >= W) continue;
        float in[3][P + 2][CIN];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy + ky - 1;
            const bool rowok = iy >= 0 && iy < H;
            const float *xr = x + ((size_t)n * H + (rowok ? iy : 0)) * W * CIN;
#pragma unroll
            for (int j = 0; j < P + 2; ++j) {
                const int ix = sx + j - 1;
                const bool ok = rowok && ix >= 0 && ix < W;
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) in[ky][j][ci] = ok ? xr[(size_t)ix * CIN + ci] : 0.0f;
            }
        }
        const size_t obase = (((size_t)n * H + oy) * W + sx) * Cout + co;
#pragma unroll
        for (int j = 0; j < P; ++j) {
            if (sx + j < W) {
                float4 acc = rule ? make_float4(0.f, 0.f, 0.f, 0.f) : b;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int ci = 0; ci < CIN; ++ci) {
                            const float v = in[ky][j + kx][ci];
                            const float4 wt = wr[(ky * 3 + kx) * CIN + ci];
                            acc.x = __builtin_fmaf(v, wt.x, acc.x);
                            acc.y = __builtin_fmaf(v, wt.y, acc.y);
                            acc.z = __builtin_fmaf(v, wt.z, acc.z);
                            acc.w = __builtin_fmaf(v, wt.w, acc.w);
                        }
                if (rule) { acc.x = acc.x + b.x; acc.y = acc.y + b.y; acc.z = acc.z + b.z; acc.w = acc.w + b.w; }
                float4 o = make_float4(pm::apply_act(acc.x, act, slope), pm::apply_act(acc.y, act, slope),
                                       pm::apply_act(acc.z, act, slope), pm::apply_act(acc.w, act, slope));
                const size_t oi = obase + (size_t)j * Cout;
                if (res1) { const float4 r1 = *(const float4 *)(res1 + oi); o.x = o.x + r1.x; o.y = o.y + r1.y; o.z = o.z + r1.z; o.w = o.w + r1.w; }
                if (res2) { const float4 r2 = *(const float4 *)(res2 + oi); o.x = o.x + r2.x; o.y = o.y + r2.y; o.z = o.z + r2.z; o.w = o.w + r2.w; }
                *(float4 *)(y + oi) = o;
            }
        }
    }
}

// 3x3, stride 1, pad 1, ONE input channel, 16 couts (PredictUpdate conv1, lifting_1d.py:38): lane = pixel, the 144
// weights are wave-uniform, a lane writes its pixel's 16 couts as 64 contiguous bytes.  Optionally also writes
// act2(conv) to a second tensor (the PredictUpdate block needs both conv1 and tanh(conv1)).
__global__ __launch_bounds__(256) void conv3x3_cin1_pix16_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                                 const float *__restrict__ bias, float *y, float *y2,
                                                                 int N, int H, int W, int act2, float slope, int rule) {
    __shared__ uint4 tanh_tab[pm::TANH_LDS_UINT4];
    pm::tanh_rows_to_lds(tanh_tab, threadIdx.x, blockDim.x);
    __syncthreads();
    const long total = (long)N * H * W;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(pm_mod(idx, W));
        const long r = pm_div(idx, W);
        const int oy = (int)(pm_mod(r, H));
        const float *plane = x + (r - oy) * W;           // start of image n
        float in[9];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = oy + ky - 1, ix = ox + kx - 1;
                in[ky * 3 + kx] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? plane[(long)iy * W + ix] : 0.0f;
            }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int co = q * 4 + i;
                const float bv = bias ? bias[co] : 0.0f;
                float acc = rule ? 0.0f : bv;
#pragma unroll
                for (int t = 0; t < 9; ++t) acc = __builtin_fmaf(in[t], w[co * 9 + t], acc);
                v[i] = rule ? acc + bv : acc;
            }
            *(float4 *)(y + idx * 16 + q * 4) = make_float4(v[0], v[1], v[2], v[3]);
            if (y2)
                *(float4 *)(y2 + idx * 16 + q * 4) =
                    act2 == pm::ACT_TANH
                        ? make_float4(pm::tanhf_rows(v[0], tanh_tab), pm::tanhf_rows(v[1], tanh_tab),
                                      pm::tanhf_rows(v[2], tanh_tab), pm::tanhf_rows(v[3], tanh_tab))
                        : make_float4(pm::apply_act(v[0], act2, slope), pm::apply_act(v[1], act2, slope),
                                      pm::apply_act(v[2], act2, slope), pm::apply_act(v[3], act2, slope));
        }
    }
}

// KxK, stride 1, pad K/2, Cin a multiple of 16, ONE or TWO couts (PredictUpdate conv4 16->1, PostProcess 64->1, SpyNet
// 16->2): a matrix-core tile would be 15/16 empty, so lane = pixel on the vector ALU, weights wave-uniform, the 16
// channels of a tap are one 64-byte run per lane.  Sum order of the spec: chunk, ky, kx, ci.
template <int K, int CIN, int CO>
__global__ __launch_bounds__(256) void conv_fewcout_pix_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                               const float *__restrict__ bias, const float *res1,
                                                               const float *res2, float *y, int N, int H, int W, int act,
                                                               float slope, int rule) {
    constexpr int P = K / 2;
    const long total = (long)N * H * W;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(pm_mod(idx, W));
        const long r = pm_div(idx, W);
        const int oy = (int)(pm_mod(r, H));
        const float *img = x + (r - oy) * W * CIN;
        float acc[CO], tot[CO], bv[CO];
#pragma unroll
        for (int co = 0; co < CO; ++co) { bv[co] = bias ? bias[co] : 0.0f; acc[co] = rule ? 0.0f : bv[co]; tot[co] = 0.0f; }
#pragma unroll 1
        for (int cb = 0; cb < CIN / 16; ++cb) {
#pragma unroll 1
            for (int ky = 0; ky < K; ++ky) {
                const int iy = oy + ky - P;
                const bool rowok = iy >= 0 && iy < H;
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const int ix = ox + kx - P;
                    float v[16];
                    if (rowok && ix >= 0 && ix < W) {
                        const float4 *p = (const float4 *)(img + ((long)iy * W + ix) * CIN + cb * 16);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 t = p[q];
                            v[q * 4] = t.x; v[q * 4 + 1] = t.y; v[q * 4 + 2] = t.z; v[q * 4 + 3] = t.w;
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 16; ++i) v[i] = 0.0f;
                    }
#pragma unroll
                    for (int ci = 0; ci < 16; ++ci)
#pragma unroll
                        for (int co = 0; co < CO; ++co)
                            acc[co] = __builtin_fmaf(v[ci], w[((co * CIN + cb * 16 + ci) * K + ky) * K + kx], acc[co]);
                }
            }
            if (rule) {             // the chunk's sum is complete: (S_0 + bias), + S_1, ...
#pragma unroll
                for (int co = 0; co < CO; ++co) { tot[co] = cb == 0 ? acc[co] + bv[co] : tot[co] + acc[co]; acc[co] = 0.0f; }
            }
        }
#pragma unroll
        for (int co = 0; co < CO; ++co) {
            float v = pm::apply_act(rule ? tot[co] : acc[co], act, slope);
            const long o = idx * CO + co;
            if (res1) v = v + res1[o];
            if (res2) v = v + res2[o];
            y[o] = v;
        }
    }
}

// The same convolution with the input patch of an 8x32 pixel tile staged in LDS, 16 channels at a time (the spec's chunk):
// the kernel above pulls K*K 64-byte runs per pixel through L1 (576 B for 64 B of HBM traffic at 3x3) and is bound
// there; here every input value is read once from global memory and K*K times from LDS (pixel stride 20 words: eight
// lanes' 16-byte reads cover the 32 banks exactly).  The global loads of a chunk are all issued before the first LDS
// store, and the next chunk's while the current one is multiplied.  Thread = output pixel, weights wave-uniform scalars,
// identical fmaf chain: chunk, ky, kx, ci ascending, zeros outside the image multiplied in like any other value.
template <int K, int CIN, int CO>
__global__ __launch_bounds__(256) void conv_fewcout_lds_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                               const float *__restrict__ bias, const float *res1,
                                                               const float *res2, float *y, int H, int W, int tiles_x,
                                                               int tiles_y, int act, float slope, int rule) {
    constexpr int P = K / 2, TH = 8, TW = 32, PH = TH + K - 1, PW = TW + K - 1, PS = 20;
    constexpr int E = PH * PW * 4, SLOTS = (E + 255) / 256, NCH = CIN / 16;
    extern __shared__ __attribute__((aligned(16))) float patch[];
    const int tid = threadIdx.x;
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y;
    const int n = b / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int r = tid >> 5, c = tid & 31;
    const float *img = x + (long)n * H * W * CIN;
    float acc[CO], tot[CO], bv[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) { bv[co] = bias ? bias[co] : 0.0f; acc[co] = rule ? 0.0f : bv[co]; tot[co] = 0.0f; }
    // staging slots e = tid + 256*j: pixel e >> 2 of the patch, channel quad e & 3
    int goff[SLOTS];                                     // element offset inside the image, -1 = outside
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) {
        const int e = tid + 256 * j;
        const int px = e >> 2, q = e & 3;
        const int ly = px / PW, lx = px - ly * PW;
        const int gy = oy0 - P + ly, gx = ox0 - P + lx;
        goff[j] = (e < E && gy >= 0 && gy < H && gx >= 0 && gx < W) ? (gy * W + gx) * CIN + q * 4 : -1;
    }
    float4 pre[SLOTS];
    auto fetch = [&](int cb) {
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) {
            pre[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (goff[j] >= 0) pre[j] = *(const float4 *)(img + goff[j] + cb * 16);
        }
    };
    fetch(0);
#pragma unroll 1
    for (int cb = 0; cb < NCH; ++cb) {
        if (cb) __syncthreads();                         // previous chunk consumed
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) {
            const int e = tid + 256 * j;
            if (e < E) *(float4 *)(patch + (e >> 2) * PS + (e & 3) * 4) = pre[j];
        }
        __syncthreads();
        if (cb + 1 < NCH) fetch(cb + 1);
        const float *wc = w + cb * 16 * K * K;
#pragma unroll 1
        for (int ky = 0; ky < K; ++ky) {
            const float *row = patch + ((r + ky) * PW + c) * PS;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                float v[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 t = *(const float4 *)(row + kx * PS + q * 4);
                    v[q * 4] = t.x; v[q * 4 + 1] = t.y; v[q * 4 + 2] = t.z; v[q * 4 + 3] = t.w;
                }
#pragma unroll
                for (int ci = 0; ci < 16; ++ci)
#pragma unroll
                    for (int co = 0; co < CO; ++co)
                        acc[co] = __builtin_fmaf(v[ci], wc[((co * CIN + ci) * K + ky) * K + kx], acc[co]);
            }
        }
        if (rule) {                 // the chunk's sum is complete: (S_0 + bias), + S_1, ...
#pragma unroll
            for (int co = 0; co < CO; ++co) { tot[co] = cb == 0 ? acc[co] + bv[co] : tot[co] + acc[co]; acc[co] = 0.0f; }
        }
    }
    const int oy = oy0 + r, ox = ox0 + c;
    if (oy < H && ox < W) {
        const long idx = ((long)n * H + oy) * W + ox;
#pragma unroll
        for (int co = 0; co < CO; ++co) {
            float v = pm::apply_act(rule ? tot[co] : acc[co], act, slope);
            const long o = idx * CO + co;
            if (res1) v = v + res1[o];
            if (res2) v = v + res2[o];
            y[o] = v;
        }
    }
}

// depthwise KxK, stride 1, pad K/2; NHWC, channel fastest
__global__ void dwconv_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                              float *y, int N, int H, int W, int C, int K) {
    const int pad = K / 2;
    const long total = (long)N * H * W * C;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(pm_mod(idx, C));
        long p = pm_div(idx, C);
        const int ox = (int)(pm_mod(p, W)); p = pm_div(p, W);
        const int oy = (int)(pm_mod(p, H));
        const int n = (int)(pm_div(p, H));
        float acc = bias ? bias[c] : 0.0f;
        for (int ky = 0; ky < K; ++ky) {
            const int iy = oy + ky - pad;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < K; ++kx) {
                const int ix = ox + kx - pad;
                if (ix < 0 || ix >= W) continue;
                acc = __builtin_fmaf(x[(((long)n * H + iy) * W + ix) * C + c], w[(c * K + ky) * K + kx], acc);
            }
        }
        y[idx] = acc;
    }
}

// depthwise 3x3 for C % 4 == 0: a thread owns 4 consecutive channels of a strip of PX consecutive pixels, keeps its
// 36 weights in registers, reads the 3 x (PX+2) input window once as 16-byte loads and writes PX 16-byte stores.
// Same sum order as dwconv_kernel: acc = bias; for ky: for kx: fmaf.
template <int PX>
__global__ __launch_bounds__(256) void dwconv3_strip_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                            const float *__restrict__ bias, float *y, int N, int H, int W,
                                                            int C) {
    const int C4 = C >> 2;
    const int strips = (W + PX - 1) / PX;
    const long total = (long)N * H * strips * C4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(pm_mod(idx, C4)) * 4;
        long r = pm_div(idx, C4);
        const int sx = (int)(pm_mod(r, strips)) * PX;
        r /= strips;
        const int oy = (int)(pm_mod(r, H));
        const long n = pm_div(r, H);
        float4 wr[9];
#pragma unroll
        for (int t = 0; t < 9; ++t)
            wr[t] = make_float4(w[(c + 0) * 9 + t], w[(c + 1) * 9 + t], w[(c + 2) * 9 + t], w[(c + 3) * 9 + t]);
        const float4 b = bias ? *(const float4 *)(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 acc[PX];
#pragma unroll
        for (int j = 0; j < PX; ++j) acc[j] = b;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy + ky - 1;
            const bool rowok = iy >= 0 && iy < H;
            const float *row = x + ((n * H + (rowok ? iy : 0)) * W) * C + c;
            float4 in[PX + 2];
#pragma unroll
            for (int j = 0; j < PX + 2; ++j) {
                const int ix = sx + j - 1;
                in[j] = (rowok && ix >= 0 && ix < W) ? *(const float4 *)(row + (long)ix * C) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float4 wt = wr[ky * 3 + kx];
#pragma unroll
                for (int j = 0; j < PX; ++j) {
                    acc[j].x = __builtin_fmaf(in[j + kx].x, wt.x, acc[j].x);
                    acc[j].y = __builtin_fmaf(in[j + kx].y, wt.y, acc[j].y);
                    acc[j].z = __builtin_fmaf(in[j + kx].z, wt.z, acc[j].z);
                    acc[j].w = __builtin_fmaf(in[j + kx].w, wt.w, acc[j].w);
                }
            }
        }
        float *o = y + ((n * H + oy) * W + sx) * C + c;
#pragma unroll
        for (int j = 0; j < PX; ++j)
            if (sx + j < W) *(float4 *)(o + (long)j * C) = acc[j];
    }
}

// The same with a vertical walk: a thread owns 4 channels of a PX-wide column of RY output rows and keeps a rolling window
// of three input rows in registers, so every input row is loaded once per thread instead of three times ((RY+2)/RY x
// (PX+2)/PX = 1.9x read amplification through L1 instead of 4.5x).  Same fmaf chain per output: acc = bias; ky, kx.
template <int PX, int RY>
__global__ __launch_bounds__(256) void dwconv3_column_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                             const float *__restrict__ bias, float *y, int N, int H, int W,
                                                             int C) {
    const int C4 = C >> 2;
    const int strips = (W + PX - 1) / PX, bands = (H + RY - 1) / RY;
    const long total = (long)N * bands * strips * C4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(pm_mod(idx, C4)) * 4;
        long r = pm_div(idx, C4);
        const int sx = (int)(pm_mod(r, strips)) * PX;
        r /= strips;
        const int oy0 = (int)(pm_mod(r, bands)) * RY;
        const long n = pm_div(r, bands);
        float4 wr[9];
#pragma unroll
        for (int t = 0; t < 9; ++t)
            wr[t] = make_float4(w[(c + 0) * 9 + t], w[(c + 1) * 9 + t], w[(c + 2) * 9 + t], w[(c + 3) * 9 + t]);
        const float4 b = bias ? *(const float4 *)(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float *img = x + (n * H * W) * C + c;
        auto load_row = [&](int iy, float4 *in) {
            const bool rowok = iy >= 0 && iy < H;
            const float *row = img + (long)(rowok ? iy : 0) * W * C;
#pragma unroll
            for (int j = 0; j < PX + 2; ++j) {
                const int ix = sx + j - 1;
                in[j] = (rowok && ix >= 0 && ix < W) ? *(const float4 *)(row + (long)ix * C) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        float4 win[3][PX + 2];
        load_row(oy0 - 1, win[0]);
        load_row(oy0, win[1]);
#pragma unroll
        for (int ry = 0; ry < RY; ++ry) {
            const int oy = oy0 + ry;
            if (oy >= H) break;
            load_row(oy + 1, win[(ry + 2) % 3]);
            float4 acc[PX];
#pragma unroll
            for (int j = 0; j < PX; ++j) acc[j] = b;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float4 *in = win[(ry + ky) % 3];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float4 wt = wr[ky * 3 + kx];
#pragma unroll
                    for (int j = 0; j < PX; ++j) {
                        acc[j].x = __builtin_fmaf(in[j + kx].x, wt.x, acc[j].x);
                        acc[j].y = __builtin_fmaf(in[j + kx].y, wt.y, acc[j].y);
                        acc[j].z = __builtin_fmaf(in[j + kx].z, wt.z, acc[j].z);
                        acc[j].w = __builtin_fmaf(in[j + kx].w, wt.w, acc[j].w);
                    }
                }
            }
            float *o = y + ((n * H + oy) * W + sx) * C + c;
#pragma unroll
            for (int j = 0; j < PX; ++j)
                if (sx + j < W) *(float4 *)(o + (long)j * C) = acc[j];
        }
    }
}

// flow warp: one thread per output pixel, loop over planes
__global__ void flow_warp_kernel(const float *__restrict__ im, const float *__restrict__ flow,
                                 const float *__restrict__ lin_x, const float *__restrict__ lin_y, float *out,
                                 int N, int C, int H, int W, int flowN, float sign) {
    const long HW = (long)H * W;
    const long total = (long)N * HW;
    const float cx = (float)(W - 1) / 2.0f, cy = (float)(H - 1) / 2.0f;
    const float mx = (float)(W - 1), my = (float)(H - 1);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int n = (int)(pm_div(idx, HW));
        const long p = idx - (long)n * HW;
        const int y = (int)(pm_div(p, W)), x = (int)(p - (long)y * W);
        const float *f = flow + (flowN == 1 ? 0 : (long)n * 2 * HW);
        const float fx = sign * f[p], fy = sign * f[HW + p];
        const float gx = lin_x[x] + fx / cx;
        const float gy = lin_y[y] + fy / cy;
        float ix = (gx + 1.0f) * cx;
        float iy = (gy + 1.0f) * cy;
        ix = __builtin_fminf(mx, __builtin_fmaxf(ix, 0.0f));
        iy = __builtin_fminf(my, __builtin_fmaxf(iy, 0.0f));
        const float xw = __builtin_floorf(ix), yn = __builtin_floorf(iy);
        const float w = ix - xw, e = 1.0f - w;
        const float nn = iy - yn, s = 1.0f - nn;
        const float nw = s * e, ne = s * w, sw = nn * e, se = nn * w;
        const int x0 = (int)xw, y0 = (int)yn;
        const int x1 = x0 + 1, y1 = y0 + 1;
        const bool x1ok = x1 <= W - 1, y1ok = y1 <= H - 1;
        for (int c = 0; c < C; ++c) {
            const float *pl = im + ((long)n * C + c) * HW;
            const float v00 = pl[(long)y0 * W + x0];
            const float v01 = x1ok ? pl[(long)y0 * W + x1] : 0.0f;
            const float v10 = y1ok ? pl[(long)y1 * W + x0] : 0.0f;
            const float v11 = (x1ok && y1ok) ? pl[(long)y1 * W + x1] : 0.0f;
            float r = v00 * nw;
            r = __builtin_fmaf(v01, ne, r);
            r = __builtin_fmaf(v10, sw, r);
            r = __builtin_fmaf(v11, se, r);
            out[((long)n * C + c) * HW + p] = r;
        }
    }
}

__global__ void avgpool2_kernel(const float *__restrict__ x, float *y, int NC, int H, int W) {
    const int Ho = H / 2, Wo = W / 2;
    const long total = (long)NC * Ho * Wo;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(pm_mod(idx, Wo));
        long p = pm_div(idx, Wo);
        const int oy = (int)(pm_mod(p, Ho));
        const long nc = pm_div(p, Ho);
        const float *r0 = x + (nc * H + 2 * oy) * W + 2 * ox;
        const float *r1 = r0 + W;
        float s = r0[0] + r0[1];
        s = s + r1[0];
        s = s + r1[1];
        y[idx] = s / 4.0f;
    }
}

__device__ __forceinline__ void up_coef(int d, int size, float rf, int &i0, int &i1, float &l0, float &l1) {
    float src = rf * ((float)d + 0.5f) - 0.5f;
    if (src < 0.0f) src = 0.0f;
    int a = (int)__builtin_floorf(src);
    if (a > size - 1) a = size - 1;
    float lam = src - (float)a;
    if (lam < 0.0f) lam = 0.0f;
    if (lam > 1.0f) lam = 1.0f;
    i0 = a;
    i1 = a + (a < size - 1 ? 1 : 0);
    l1 = lam;
    l0 = 1.0f - lam;
}

// bilinear upsampling by a power-of-two factor f, align_corners=False: src = (dst + 0.5)/f - 0.5
__global__ void bilinear_up_kernel(const float *__restrict__ x, float *y, int NC, int H, int W, int f, float scale) {
    const int Ho = f * H, Wo = f * W;
    const float rf = 1.0f / (float)f;
    const long total = (long)NC * Ho * Wo;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(pm_mod(idx, Wo));
        long p = pm_div(idx, Wo);
        const int oy = (int)(pm_mod(p, Ho));
        const long nc = pm_div(p, Ho);
        int x0, x1, y0, y1;
        float lx0, lx1, ly0, ly1;
        up_coef(ox, W, rf, x0, x1, lx0, lx1);
        up_coef(oy, H, rf, y0, y1, ly0, ly1);
        const float *r0 = x + (nc * H + y0) * W, *r1 = x + (nc * H + y1) * W;
        const float t0 = __builtin_fmaf(r0[x0], lx0, r0[x1] * lx1);
        const float t1 = __builtin_fmaf(r1[x0], lx0, r1[x1] * lx1);
        y[idx] = __builtin_fmaf(t0, ly0, t1 * ly1) * scale;
    }
}

// bilinear downsampling by an even factor f, align_corners=False: src = f*dst + f/2 - 0.5 -> the two centre samples
__global__ void bilinear_down_kernel(const float *__restrict__ x, float *y, int NC, int H, int W, int f, float div) {
    const int Ho = H / f, Wo = W / f, c = f / 2 - 1;
    const long total = (long)NC * Ho * Wo;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(pm_mod(idx, Wo));
        long p = pm_div(idx, Wo);
        const int oy = (int)(pm_mod(p, Ho));
        const long nc = pm_div(p, Ho);
        const float *r0 = x + (nc * H + f * oy + c) * W + f * ox + c;
        const float *r1 = r0 + W;
        const float t0 = r0[0] * 0.5f + r0[1] * 0.5f;
        const float t1 = r1[0] * 0.5f + r1[1] * 0.5f;
        y[idx] = (t0 * 0.5f + t1 * 0.5f) / div;
    }
}

}  // namespace

extern "C" int pmctf_conv2d_smallcin_f32(const float *x, const float *w, const float *bias, const float *res1,
                                         const float *res2, float *y, int N, int H, int W, int Cin, int Cout,
                                         int KH, int KW, int stride, int pad_h, int pad_w, int act, float slope,
                                         int sum_rule, void *stream) {
    if (!x || !w || !y || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cin > 4 || Cout <= 0 || KH <= 0 || KW <= 0 ||
        stride <= 0 || (sum_rule != PMCTF_SUM_CHAIN && sum_rule != PMCTF_SUM_BLOCKS && sum_rule != PMCTF_SUM_GEMM &&
                        sum_rule != PMCTF_SUM_GEMV_3X3) ||
        (sum_rule == PMCTF_SUM_GEMV_3X3 && (Cin != 1 || KH != 3 || KW != 3)))
        return PMCTF_EINVAL;
    const int rule = sum_rule;
    const int Ho = (H + 2 * pad_h - KH) / stride + 1, Wo = (W + 2 * pad_w - KW) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return PMCTF_EINVAL;
    const bool gemm = rule == PMCTF_SUM_GEMM || rule == PMCTF_SUM_GEMV_3X3;   // small planes only (the rules' definition): the generic kernel
    if (!gemm && KH == 3 && KW == 3 && stride == 1 && pad_h == 1 && pad_w == 1 && Cin <= 3 && Cout >= 32 && Cout <= 256 &&
        (Cout & 3) == 0) {
        const int Q = Cout / 4, G = 64 / Q;                    // lanes per pixel, strips per wave
        const int wstrips_per_row = (W + G * 8 - 1) / (G * 8);
        const long total_items = (long)N * H * wstrips_per_row;
        long nb = (total_items + 3) / 4;
        if (nb > 16384) nb = 16384;
        dim3 grid((unsigned)nb), block(256);
        hipStream_t st = (hipStream_t)stream;
#define PM_STRIP(C_)                                                                                                 \
    PM_LAUNCH((conv3x3_smallcin_strip_kernel<C_>), grid, block, 0, st, x, w, bias, res1, res2, y, N, H, W, Cout, act,  \
              slope, Q, G, wstrips_per_row, total_items, rule);                                                      \
    return launch_ok();
        if (Cin == 1) { PM_STRIP(1) }
        if (Cin == 2) { PM_STRIP(2) }
        if (Cin == 3) { PM_STRIP(3) }
#undef PM_STRIP
    }
    if (!gemm && KH == KW && Cout <= 256 && (long)N * Ho * Wo < (1L << 31)) {
        const int ppp = 256 / Cout;
        long nb = ((long)N * Ho * Wo + ppp - 1) / ppp;
        if (nb > 16384) nb = 16384;
        dim3 grid((unsigned)nb), block(256);
        hipStream_t st = (hipStream_t)stream;
#define PM_SC(K_, C_)                                                                                               \
    PM_LAUNCH((conv_smallcin_reg_kernel<K_, C_>), grid, block, 0, st, x, w, bias, res1, res2, y, N, H, W, Cout, stride,  \
              pad_h, pad_w, Ho, Wo, act, slope, rule);                                                              \
    return launch_ok();
        if (KH == 3 && Cin == 1) { PM_SC(3, 1) }
        if (KH == 3 && Cin == 2) { PM_SC(3, 2) }
        if (KH == 3 && Cin == 3) { PM_SC(3, 3) }
        if (KH == 1 && Cin == 2) { PM_SC(1, 2) }
#undef PM_SC
    }
    const size_t smem = (size_t)Cout * Cin * KH * KW * sizeof(float);
    if (smem > 64 * 1024) return PMCTF_EINVAL;
    const long total = (long)N * Ho * Wo * Cout;
    unsigned g = nblocks(total);
    if (g > 8192) g = 8192;
    PM_LAUNCH(conv_smallcin_kernel, dim3(g), dim3(256), smem, (hipStream_t)stream, x, w, bias, res1, res2, y,
                       N, H, W, Cin, Cout, KH, KW, stride, pad_h, pad_w, Ho, Wo, act, slope, rule);
    return launch_ok();
}

extern "C" int pmctf_conv3x3_cin1_dual_f32(const float *x, const float *w, const float *bias, float *y, float *y2, int N,
                                           int H, int W, int Cout, int act2, float slope, int sum_rule, void *stream) {
    if (!x || !w || !y || N <= 0 || H <= 0 || W <= 0 || Cout != 16 || (sum_rule != 0 && sum_rule != 1)) return PMCTF_EINVAL;
    unsigned g = nblocks((long)N * H * W);
    if (g > 16384) g = 16384;
    PM_LAUNCH(conv3x3_cin1_pix16_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, y2, N, H, W, act2,
              slope, sum_rule);
    return launch_ok();
}

extern "C" int pmctf_conv2d_fewcout_supported(int Cin, int Cout, int K) {
    return (K == 3 && Cin == 16 && Cout == 1) || (K == 3 && Cin == 64 && Cout == 1) || (K == 7 && Cin == 16 && Cout == 2);
}

extern "C" int pmctf_conv2d_fewcout_f32(const float *x, const float *w, const float *bias, const float *res1,
                                        const float *res2, float *y, int N, int H, int W, int Cin, int Cout, int K,
                                        int act, float slope, int sum_rule, void *stream) {
    if (!x || !w || !y || N <= 0 || H <= 0 || W <= 0 || !pmctf_conv2d_fewcout_supported(Cin, Cout, K) ||
        (sum_rule != 0 && sum_rule != 1))
        return PMCTF_EINVAL;
    unsigned g = nblocks((long)N * H * W);
    if (g > 16384) g = 16384;
    hipStream_t st = (hipStream_t)stream;
    static const bool use_lds = [] { const char *e = getenv("PMCTF_FEWCOUT_LDS"); return !e || atoi(e) != 0; }();
    const int tiles_x = (W + 31) / 32, tiles_y = (H + 7) / 8;
    const long tiles = (long)N * tiles_x * tiles_y;
#define PM_FC(K_, CI_, CO_)                                                                                           \
    if (use_lds && tiles <= 0x7fffffffL && (long)H * W * CI_ < 0x7fffffffL) {                                         \
        const size_t smem = (size_t)(8 + K_ - 1) * (32 + K_ - 1) * 20 * sizeof(float);                                \
        PM_LAUNCH((conv_fewcout_lds_kernel<K_, CI_, CO_>), dim3((unsigned)tiles), dim3(256), smem, st, x, w, bias,     \
                  res1, res2, y, H, W, tiles_x, tiles_y, act, slope, sum_rule);                                       \
        return launch_ok();                                                                                           \
    }                                                                                                                 \
    PM_LAUNCH((conv_fewcout_pix_kernel<K_, CI_, CO_>), dim3(g), dim3(256), 0, st, x, w, bias, res1, res2, y, N, H, W,   \
              act, slope, sum_rule);                                                                                  \
    return launch_ok();
    if (K == 3 && Cin == 16) { PM_FC(3, 16, 1) }
    if (K == 3 && Cin == 64) { PM_FC(3, 64, 1) }
    PM_FC(7, 16, 2)
#undef PM_FC
}

extern "C" int pmctf_dwconv2d_nhwc_f32(const float *x, const float *w, const float *bias, float *y, int N, int H,
                                       int W, int C, int K, void *stream) {
    if (!x || !w || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || !(K & 1)) return PMCTF_EINVAL;
    if (K == 3 && (C & 3) == 0) {
        static const int column = [] { const char *e = getenv("PMCTF_DWCONV_COLUMN"); return e ? atoi(e) : 1; }();
        const long threads_col = (long)N * ((H + 7) / 8) * ((W + 3) / 4) * (C / 4);
        if (column && threads_col >= 256L * 1024) {          // enough columns to fill the chip four times over
            unsigned gc = nblocks(threads_col);
            if (gc > 16384) gc = 16384;
            PM_LAUNCH((dwconv3_column_kernel<4, 8>), dim3(gc), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, N, H, W, C);
            return launch_ok();
        }
        unsigned g4 = nblocks((long)N * H * ((W + 3) / 4) * (C / 4));
        if (g4 > 16384) g4 = 16384;
        PM_LAUNCH(dwconv3_strip_kernel<4>, dim3(g4), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, N, H, W, C);
        return launch_ok();
    }
    unsigned g = nblocks((long)N * H * W * C);
    if (g > 16384) g = 16384;
    PM_LAUNCH(dwconv_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, N, H, W, C, K);
    return launch_ok();
}

extern "C" int pmctf_flow_warp_f32(const float *im, const float *flow, const float *lin_x, const float *lin_y,
                                   float *out, int N, int C, int H, int W, int flowN, float flow_sign, void *stream) {
    if (!im || !flow || !lin_x || !lin_y || !out || N <= 0 || C <= 0 || H < 2 || W < 2 || (flowN != 1 && flowN != N))
        return PMCTF_EINVAL;
    unsigned g = nblocks((long)N * H * W);
    if (g > 16384) g = 16384;
    PM_LAUNCH(flow_warp_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, im, flow, lin_x, lin_y, out, N, C,
                       H, W, flowN, flow_sign);
    return launch_ok();
}

extern "C" int pmctf_avgpool2_f32(const float *x, float *y, int NC, int H, int W, void *stream) {
    if (!x || !y || NC <= 0 || H < 2 || W < 2) return PMCTF_EINVAL;
    unsigned g = nblocks((long)NC * (H / 2) * (W / 2));
    if (g > 16384) g = 16384;
    PM_LAUNCH(avgpool2_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, x, y, NC, H, W);
    return launch_ok();
}

extern "C" int pmctf_bilinear_up_f32(const float *x, float *y, int NC, int H, int W, int factor, float scale,
                                     void *stream) {
    if (!x || !y || NC <= 0 || H <= 0 || W <= 0 || (factor != 2 && factor != 4 && factor != 8)) return PMCTF_EINVAL;
    unsigned g = nblocks((long)NC * H * W * factor * factor);
    if (g > 16384) g = 16384;
    PM_LAUNCH(bilinear_up_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, x, y, NC, H, W, factor, scale);
    return launch_ok();
}

extern "C" int pmctf_bilinear_down_f32(const float *x, float *y, int NC, int H, int W, int factor, float div,
                                       void *stream) {
    if (!x || !y || NC <= 0 || (factor != 2 && factor != 4 && factor != 8) || H < factor || W < factor)
        return PMCTF_EINVAL;
    unsigned g = nblocks((long)NC * (H / factor) * (W / factor));
    if (g > 16384) g = 16384;
    PM_LAUNCH(bilinear_down_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, x, y, NC, H, W, factor, div);
    return launch_ok();
}

extern "C" int pmctf_bilinear_up2_f32(const float *x, float *y, int NC, int H, int W, float scale, void *stream) {
    return pmctf_bilinear_up_f32(x, y, NC, H, W, 2, scale, stream);
}

extern "C" int pmctf_bilinear_down2_f32(const float *x, float *y, int NC, int H, int W, float div, void *stream) {
    return pmctf_bilinear_down_f32(x, y, NC, H, W, 2, div, stream);
}
