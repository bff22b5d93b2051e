// ew_ops.hip — elementwise / layout / quantisation kernels of the pMCTF encode path.
// All are HBM-bound, one rounding per written operation (PM-F32), coalesced on the
// output's unit-stride dimension.  Reference call sites are listed in include/pmctf_hip.h.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pm_device_math.h"
#include "launch.h"
#include "../../include/pmctf_hip.h"

namespace {

inline int launch_ok() { return pm_launch_status(); }
inline unsigned grid_for(long n, int bs = 256, unsigned cap = 16384) {
    long b = (n + bs - 1) / bs;
    if (b < 1) b = 1;
    return (unsigned)(b > cap ? cap : b);
}

struct View { long s[4]; };  // element strides of the logical (n,c,h,w) index

__device__ __forceinline__ float ew_apply(int op, float a, float b, float alpha, float beta) {
    switch (op) {
    case PMCTF_EW_COPY: return a;
    case PMCTF_EW_ADD: return a + b;
    case PMCTF_EW_SUB: return a - b;
    case PMCTF_EW_MUL: return a * b;
    case PMCTF_EW_DIV: return a / b;
    case PMCTF_EW_MULS: return a * alpha;
    case PMCTF_EW_DIVS: return a / alpha;
    case PMCTF_EW_ADD_MULS: return a + b * alpha;
    case PMCTF_EW_SUB_MULS: return a - b * alpha;
    case PMCTF_EW_ADD_MULS_MULS: return (a + b * alpha) * beta;
    case PMCTF_EW_CLAMP_MULS: { float t = a * alpha; t = t < -beta ? -beta : t; return t > beta ? beta : t; }
    case PMCTF_EW_ROUND_CLAMP_MULS: { float t = a * alpha; t = t < -beta ? -beta : t; t = t > beta ? beta : t; return __builtin_rintf(t); }
    case PMCTF_EW_ROUND: return __builtin_rintf(a);
    case PMCTF_EW_LEAKY: return a > 0.0f ? a : a * alpha;
    case PMCTF_EW_ADD_MULS2: return a + (b * alpha) * beta;
    case PMCTF_EW_SUB_MULS2: return a - (b * alpha) * beta;
    case PMCTF_EW_TANH: return pm::tanhf_(a);
    case PMCTF_EW_ROUND_CLAMP: { float t = a < alpha ? alpha : a; t = t > beta ? beta : t; return __builtin_rintf(t); }
    default: return a;
    }
}

// IT = int when the element count fits 31 bits (every plane of the path): 64-bit integer division costs several times
// the 32-bit one and the index split is most of this kernel's instructions.
template <bool CFAST, typename IT>
__global__ void ew_kernel(int op, float *out, View vo, const float *a, View va, const float *b, View vb, int N, int C,
                          int H, int W, float alpha, float beta) {
    const long total = (long)N * C * H * W;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        IT r = (IT)idx;
        int n, c, h, w;
        if (CFAST) { c = (int)(r % C); r /= C; w = (int)(r % W); r /= W; h = (int)(r % H); n = (int)(r / H); }
        else { w = (int)(r % W); r /= W; h = (int)(r % H); r /= H; c = (int)(r % C); n = (int)(r / C); }
        const float av = a[n * va.s[0] + c * va.s[1] + h * va.s[2] + w * va.s[3]];
        const float bv = b ? b[n * vb.s[0] + c * vb.s[1] + h * vb.s[2] + w * vb.s[3]] : 0.0f;
        out[n * vo.s[0] + c * vo.s[1] + h * vo.s[2] + w * vo.s[3]] = ew_apply(op, av, bv, alpha, beta);
    }
}

// Channel-fastest views whose channel count and strides are multiples of four (channel slices / concatenations of NHWC
// tensors, the parity-class gathers of the four-step coder): four channels per thread, 16-byte accesses, one index split
// per four elements.  Same ew_apply per element.
template <typename IT>
__global__ void ew_c4_kernel(int op, float *out, View vo, const float *a, View va, const float *b, View vb, int N, int C4,
                             int H, int W, float alpha, float beta) {
    const long total = (long)N * C4 * H * W;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        IT r = (IT)idx;
        const int c = (int)(r % C4) * 4; r /= C4;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H);
        const int n = (int)(r / H);
        const float4 av = *(const float4 *)(a + n * va.s[0] + c + h * va.s[2] + w * va.s[3]);
        const float4 bv = b ? *(const float4 *)(b + n * vb.s[0] + c + h * vb.s[2] + w * vb.s[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 o;
        o.x = ew_apply(op, av.x, bv.x, alpha, beta);
        o.y = ew_apply(op, av.y, bv.y, alpha, beta);
        o.z = ew_apply(op, av.z, bv.z, alpha, beta);
        o.w = ew_apply(op, av.w, bv.w, alpha, beta);
        *(float4 *)(out + n * vo.s[0] + c + h * vo.s[2] + w * vo.s[3]) = o;
    }
}

// Unary op whose input is the TRANSPOSE of a dense plane (element (h, w) at w*H + h) written to dense rows: 32x32 tiles
// through LDS, so that both the reads (along h) and the writes (along w) are coalesced.  The strided-view form reads
// 64 different cache lines per wave (8.8 us for a 576x1920 plane against 3.8 us for a copy of the same bytes).
__global__ __launch_bounds__(256) void ew_transpose_kernel(int op, float *out, const float *a, int C, int H, int W,
                                                           long so_n, long so_c, long sa_n, long sa_c, float alpha,
                                                           float beta) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int w0 = blockIdx.x * 32, h0 = blockIdx.y * 32;
    const int n = blockIdx.z / C, c = blockIdx.z - n * C;
    const float *src = a + n * sa_n + c * sa_c;
    float *dst = out + n * so_n + c * so_c;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int w = w0 + ty + 8 * j, h = h0 + tx;
        if (w < W && h < H) tile[ty + 8 * j][tx] = src[(long)w * H + h];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int h = h0 + ty + 8 * j, w = w0 + tx;
        if (h < H && w < W) dst[(long)h * W + w] = ew_apply(op, tile[tx][ty + 8 * j], 0.0f, alpha, beta);
    }
}

// All operands dense with the same strides (whole planes, whole NHWC / NCHW tensors): the op is a flat map over `total`
// consecutive floats, whatever the logical order — no index arithmetic, 16-byte accesses.  Same ew_apply per element.
template <bool VEC>
__global__ void ew_flat_kernel(int op, float *out, const float *a, const float *b, long total, float alpha, float beta) {
    if (VEC) {
        const long total4 = total >> 2;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
            const float4 av = ((const float4 *)a)[i];
            const float4 bv = b ? ((const float4 *)b)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 o;
            o.x = ew_apply(op, av.x, bv.x, alpha, beta);
            o.y = ew_apply(op, av.y, bv.y, alpha, beta);
            o.z = ew_apply(op, av.z, bv.z, alpha, beta);
            o.w = ew_apply(op, av.w, bv.w, alpha, beta);
            ((float4 *)out)[i] = o;
        }
    } else {
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
            out[i] = ew_apply(op, a[i], b ? b[i] : 0.0f, alpha, beta);
    }
}

// strides of a dense tensor over dims d (any dimension order; size-1 dimensions may carry any stride)?
inline bool dense_strides(const long *s, const int *d) {
    int order[4] = {0, 1, 2, 3};
    for (int i = 0; i < 4; ++i)
        for (int j = i + 1; j < 4; ++j)
            if (s[order[j]] < s[order[i]]) { const int t = order[i]; order[i] = order[j]; order[j] = t; }
    long expect = 1;
    for (int i = 0; i < 4; ++i) {
        const int k = order[i];
        if (d[k] == 1) continue;
        if (s[k] != expect) return false;
        expect *= d[k];
    }
    return true;
}

// SpyNet level input: [im1 x3, warp x3, flow_up x2] -> NHWC 8 channels (video_net.py:116-119; the three image
// channels are identical copies of Y, pMCTF_L.py:453-454)
__global__ void spynet_pack8_kernel(const float *__restrict__ im1, const float *__restrict__ wrp,
                                    const float *__restrict__ fu, float *out, long HW) {
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += (long)gridDim.x * blockDim.x) {
        const float a = im1[p], b = wrp[p];
        float4 lo = make_float4(a, a, a, b), hi = make_float4(b, b, fu[p], fu[HW + p]);
        float4 *o = (float4 *)(out + p * 8);
        o[0] = lo; o[1] = hi;
    }
}

// reflectionPadSkip + 3x1 conv along H on single-channel planes (lifting_1d.py:98,105-106):
// rule 0: acc = bias; acc = fmaf(x[refl(y-1)], w0, acc); acc = fmaf(x[y], w1, acc); acc = fmaf(x[refl(y+1)], w2, acc)
// rule 1: the same three fmaf from zero, the bias added last (what ATen's oneDNN path computes; rule 0 is what its
//         im2col + gemv path computes for the planes that take it, see pmctf_hip.h)
__global__ void lift_skip3_kernel(const float *__restrict__ x, float *y, int NC, int H, int W, float w0, float w1,
                                  float w2, float bias, int rule) {
    const long total = (long)NC * H * W;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int xw = (int)(pm_mod(idx, W));
        long r = pm_div(idx, W);
        const int yy = (int)(pm_mod(r, H));
        const long nc = pm_div(r, H);
        const int ym = yy == 0 ? 1 : yy - 1;
        const int yp = yy == H - 1 ? H - 2 : yy + 1;
        const float *pl = x + nc * H * W;
        float acc = rule ? 0.0f : bias;
        acc = __builtin_fmaf(pl[(long)ym * W + xw], w0, acc);
        acc = __builtin_fmaf(pl[(long)yy * W + xw], w1, acc);
        acc = __builtin_fmaf(pl[(long)yp * W + xw], w2, acc);
        y[idx] = rule ? acc + bias : acc;
    }
}

__global__ void nearest_up2_kernel(const float *__restrict__ x, float *y, int N, int H, int W, int C) {
    const long total = (long)N * 2 * H * 2 * W * C;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(pm_mod(idx, C));
        long r = pm_div(idx, C);
        const int ox = (int)(pm_mod(r, (2 * W))); r = pm_div(r, 2 * W);
        const int oy = (int)(pm_mod(r, (2 * H)));
        const int n = (int)(pm_div(r, (2 * H)));
        y[idx] = x[(((long)n * H + (oy >> 1)) * W + (ox >> 1)) * C + c];
    }
}

// nn.PixelShuffle(2) on NHWC: out[n,2h+i,2w+j,c] = act(x[n,h,w,c*4+i*2+j])
__global__ void pixel_shuffle2_kernel(const float *__restrict__ x, float *y, int N, int H, int W, int C, int act,
                                      float slope) {
    const long total = (long)N * 2 * H * 2 * W * C;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(pm_mod(idx, C));
        long r = pm_div(idx, C);
        const int ox = (int)(pm_mod(r, (2 * W))); r = pm_div(r, 2 * W);
        const int oy = (int)(pm_mod(r, (2 * H)));
        const int n = (int)(pm_div(r, (2 * H)));
        const float v = x[(((long)n * H + (oy >> 1)) * W + (ox >> 1)) * (4 * C) + c * 4 + (oy & 1) * 2 + (ox & 1)];
        y[idx] = pm::apply_act(v, act, slope);
    }
}

// C % 4 == 0 forms: a thread owns four channels of one INPUT pixel and writes the four output pixels it feeds with
// 16-byte stores (the scalar kernels above spend their time on index arithmetic and 4-byte accesses).
__global__ __launch_bounds__(256) void nearest_up2_vec4_kernel(const float *__restrict__ x, float *y, int N, int H, int W, int C) {
    const int C4 = C >> 2;
    const long total = (long)N * H * W * C4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(pm_mod(idx, C4)) * 4;
        long r = pm_div(idx, C4);
        const int w = (int)(pm_mod(r, W)); r = pm_div(r, W);
        const int h = (int)(pm_mod(r, H));
        const long n = pm_div(r, H);
        const float4 v = *(const float4 *)(x + ((n * H + h) * W + w) * C + c);
        float *o = y + ((n * 2 * H + 2 * h) * (2L * W) + 2 * w) * C + c;
        *(float4 *)o = v;
        *(float4 *)(o + C) = v;
        *(float4 *)(o + 2L * W * C) = v;
        *(float4 *)(o + 2L * W * C + C) = v;
    }
}

__global__ __launch_bounds__(256) void pixel_shuffle2_vec4_kernel(const float *__restrict__ x, float *y, int N, int H, int W, int C,
                                                                  int act, float slope) {
    const int C4 = C >> 2;
    const long total = (long)N * H * W * C4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(pm_mod(idx, C4)) * 4;                       // output channels c .. c+3 = input channels 4c .. 4c+15
        long r = pm_div(idx, C4);
        const int w = (int)(pm_mod(r, W)); r = pm_div(r, W);
        const int h = (int)(pm_mod(r, H));
        const long n = pm_div(r, H);
        const float4 *p = (const float4 *)(x + ((n * H + h) * W + w) * (4L * C) + 4 * c);
        float4 q[4];                                             // q[k] = input channels of output channel c+k: (i,j) = x,y,z,w
#pragma unroll
        for (int k = 0; k < 4; ++k) q[k] = p[k];
        float *o = y + ((n * 2 * H + 2 * h) * (2L * W) + 2 * w) * C + c;
        const float4 o00 = make_float4(q[0].x, q[1].x, q[2].x, q[3].x), o01 = make_float4(q[0].y, q[1].y, q[2].y, q[3].y);
        const float4 o10 = make_float4(q[0].z, q[1].z, q[2].z, q[3].z), o11 = make_float4(q[0].w, q[1].w, q[2].w, q[3].w);
        auto a4 = [&](float4 v) {
            return make_float4(pm::apply_act(v.x, act, slope), pm::apply_act(v.y, act, slope), pm::apply_act(v.z, act, slope),
                               pm::apply_act(v.w, act, slope));
        };
        *(float4 *)o = a4(o00);
        *(float4 *)(o + C) = a4(o01);
        *(float4 *)(o + 2L * W * C) = a4(o10);
        *(float4 *)(o + 2L * W * C + C) = a4(o11);
    }
}

// ConvFFN3 gate (video/layers.py:163-167): out = leaky(x1, 0.1) + leaky(x2, 0.01), x = [x1 | x2] on channels
__global__ void ffn3_mix_kernel(const float *__restrict__ x, float *y, long P, int C) {
    const long total = P * C;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(pm_mod(idx, C));
        const long p = pm_div(idx, C);
        const float x1 = x[p * 2 * C + c], x2 = x[p * 2 * C + C + c];
        const float a = x1 > 0.0f ? x1 : x1 * 0.1f;
        const float b = x2 > 0.0f ? x2 : x2 * 0.01f;
        y[idx] = a + b;
    }
}

// LSTM2D gates (long_context.py:20-33): f = sigmoid(xh); cell' = f*cell + f*tanh(xh); hidden' = f*tanh(cell')
// aten_threads > 0: the gate tensor is the reference's contiguous (ref_planes, C, H, W) tensor evaluated by ATen with that
// many intra-op threads — the scalar tail of every thread's slice takes libm's expf (pm_glibc_expf.h)
__global__ void lstm_gates_kernel(const float *__restrict__ xh, const float *__restrict__ cell, float *cell_out,
                                  float *hid_out, long P, int C, int Cc, long HW, int ref_planes, int aten_threads) {
    __shared__ uint4 tanh_tab[pm::TANH_LDS_UINT4];
    pm::tanh_rows_to_lds(tanh_tab, threadIdx.x, blockDim.x);
    __syncthreads();
    const long total = P * C;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(pm_mod(idx, C));
        const long p = pm_div(idx, C);
        const float v = xh[idx];
        float g;
        if (aten_threads > 0) {
            const long n = pm_div(p, HW), hw = p - n * HW;
            const long ref_idx = ((n % ref_planes) * C + c) * HW + hw;        // NCHW index inside the reference's tensor
            g = pm_aten_sigmoid_tail(ref_idx, (long)ref_planes * C * HW, aten_threads) ? pm_aten_sigmoidf_scalar(v)
                                                                                       : pm::sigmoidf_(v);
        } else {
            g = pm::sigmoidf_(v);
        }
        const float ct = pm::tanhf_rows(v, tanh_tab);
        const float cprev = cell[p * Cc + (Cc == 1 ? 0 : c)];
        const float t1 = g * cprev;
        const float t2 = g * ct;
        const float cn = t1 + t2;
        cell_out[idx] = cn;
        hid_out[idx] = g * pm::tanhf_rows(cn, tanh_tab);
    }
}

// GaussianEncoder.build_indexes (entropy_models.py:269-273) in PM-F32
__device__ __forceinline__ int scale_index(float s, float lmin, float step) {
    s = s < 1e-5f ? 1e-5f : s;   // torch.maximum(scales, 1e-5)
    float v = (pm::logf_(s) - lmin) / step;
    v = v >= 0.0f ? v : 0.0f;      // also maps NaN to row 0 instead of an out-of-range row
    v = v > 255.0f ? 255.0f : v;
    return (int)v;
}
__device__ __forceinline__ short sym16(float q) {  // symbols.clamp(-30000, 30000).to(int16), entropy_models.py:38
    q = q < -30000.0f ? -30000.0f : q;
    q = q > 30000.0f ? 30000.0f : q;
    return (short)(int)q;
}

// One step of ContextFusionFourStep.forward(write=True) (context_fusion_4step.py:127-189) for parity class k.
// x, so_far: planes [N,H,W]; params NHWC [N,H,W,2] = (scale, mean).  Writes this step's full-size push
// (int16 symbol + int16 CDF row per element, NCHW order) and x_hat_so_far at the class-k positions.
__global__ void fourstep_quant_kernel(const float *__restrict__ x, const float *__restrict__ params, float *so_far,
                                      short *sym, short *idx, int N, int H, int W, int k, int psub, float lmin,
                                      float step) {
    const long total = (long)N * H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int xw = (int)(pm_mod(i, W));
        const int yy = (int)pm_mod(pm_div(i, W), H);
        const int cls = (yy & 1) * 2 + (xw & 1);
        if (cls == k) {
            const long pi = psub ? ((pm_div(i, ((long)W * H))) * (H >> 1) + (yy >> 1)) * (W >> 1) + (xw >> 1) : i;
            const float scale = params[pi * 2], mean = params[pi * 2 + 1];
            const float res = x[i] - mean;
            const float q = __builtin_rintf(res);
            so_far[i] = q + mean;
            sym[i] = sym16(q);
            idx[i] = (short)scale_index(scale, lmin, step);
        } else {
            if (k == 0) so_far[i] = 0.0f;
            sym[i] = 0;
            idx[i] = 0;
        }
    }
}

// LL subband (pWave.py:408-418 with gaussian_model.py:59-63): ll is already round(clamp(ll*QP_ll));
// res = ll - mean; sym = round(res); ll_hat = round(round(res) + mean)
__global__ void ll_quant_kernel(const float *__restrict__ ll, const float *__restrict__ params, float *ll_hat,
                                short *sym, short *idx, long total, int planes, float lmin, float step) {
    const long npos = planes > 0 ? total / planes : total;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long o = planes > 0 ? (pm_mod(i, npos)) * planes + pm_div(i, npos) : i;     // position-major for the sequential coder
        const float scale = params[i * 2], mean = params[i * 2 + 1];
        const float res = __builtin_rintf(ll[i]) - mean;
        const float q = __builtin_rintf(res);
        ll_hat[i] = __builtin_rintf(q + mean);
        sym[o] = sym16(q);
        idx[o] = (short)scale_index(scale, lmin, step);
    }
}

// MV hyper latent: z_hat = round(z) (NHWC), push in NCHW order with CDF row = channel (entropy_models.py:180-193)
__global__ void z_symbols_kernel(const float *__restrict__ z, float *z_hat, short *sym, short *idx, int HW, int C) {
    const long total = (long)HW * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(pm_mod(i, C));
        const long p = pm_div(i, C);
        const float q = __builtin_rintf(z[i]);
        z_hat[i] = q;
        sym[(long)c * HW + p] = sym16(q);
        idx[(long)c * HW + p] = (short)c;
    }
}

// One step t of MVCoderQuad.forward_four_part_prior(write=True) (four_part_prior.py:89-208, enc_dec_quant).
// y: NHWC [HW,64] latent; common: NHWC [HW,192] = (quant_step | scales | means); sp: NHWC [HW,128] spatial-prior
// output of this step (8 chunks of 16: scales_0..3, means_0..3), unused for t == 0.  Channel group g=c/16 is
// coded at the positions of parity class PERM[t][g].  so_far: NHWC [HW,64] (y_hat accumulated, un-dequantised).
__constant__ int MV_PERM[4][4] = {{0, 1, 2, 3}, {3, 2, 1, 0}, {2, 3, 0, 1}, {1, 0, 3, 2}};
__global__ void mv_fourpart_kernel(const float *__restrict__ y, const float *__restrict__ common,
                                   const float *__restrict__ sp, float *so_far, short *sym, short *idx, int H, int W,
                                   int t, float lmin, float step) {
    const long HW = (long)H * W;
    const long total = HW * 64;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i & 63);
        const long p = i >> 6;
        const int xw = (int)(pm_mod(p, W)), yy = (int)(pm_div(p, W));
        const int cls = (yy & 1) * 2 + (xw & 1);
        const int g = c >> 4, cc = c & 15;
        if (MV_PERM[t][g] == cls) {
            float qs = common[p * 192 + c];
            qs = qs > 0.5f ? qs : 0.5f;                    // LowerBound(quant_step, 0.5)
            const float q_enc = 1.0f / qs;
            float scale, mean;
            if (t == 0) { scale = common[p * 192 + 64 + c]; mean = common[p * 192 + 128 + c]; }
            else { scale = sp[p * 128 + g * 16 + cc]; mean = sp[p * 128 + 64 + g * 16 + cc]; }
            const float ys = y[i] * q_enc;
            const float q = __builtin_rintf(ys - mean);
            so_far[i] = q + mean;
            sym[(long)cc * HW + p] = sym16(q);
            idx[(long)cc * HW + p] = (short)scale_index(scale, lmin, step);
        } else if (t == 0) {
            so_far[i] = 0.0f;
        }
    }
}

// y_hat = y_hat_so_far * q_dec (four_part_prior.py:193-194)
__global__ void mv_dequant_kernel(const float *__restrict__ so_far, const float *__restrict__ common, float *y_hat,
                                  long HW) {
    const long total = HW * 64;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i & 63);
        const long p = i >> 6;
        float qs = common[p * 192 + c];
        qs = qs > 0.5f ? qs : 0.5f;
        y_hat[i] = so_far[i] * qs;
    }
}

}  // namespace

extern "C" int pmctf_ew_f32(int op, float *out, const int64_t *so, const float *a, const int64_t *sa, const float *b,
                            const int64_t *sb, int N, int C, int H, int W, float alpha, float beta, int cfast,
                            void *stream) {
    if (!out || !a || !so || !sa || N <= 0 || C <= 0 || H <= 0 || W <= 0 || op < 0 || op > PMCTF_EW_LAST)
        return PMCTF_EINVAL;
    View vo, va, vb;
    for (int i = 0; i < 4; ++i) { vo.s[i] = so[i]; va.s[i] = sa[i]; vb.s[i] = (b && sb) ? sb[i] : 0; }
    const long total = (long)N * C * H * W;
    hipStream_t st = (hipStream_t)stream;
    {   // flat form: every operand dense with the output's strides (size-1 dimensions do not matter)
        const int d[4] = {N, C, H, W};
        bool same = (b == nullptr) || (sb != nullptr);
        for (int i = 0; i < 4 && same; ++i)
            if (d[i] != 1 && (va.s[i] != vo.s[i] || (b && vb.s[i] != vo.s[i]))) same = false;
        if (same && dense_strides(vo.s, d)) {
            const bool vec = (total & 3) == 0 && (((uintptr_t)out | (uintptr_t)a | (uintptr_t)b) & 15) == 0;
            if (vec)
                PM_LAUNCH(ew_flat_kernel<true>, dim3(grid_for(total >> 2)), dim3(256), 0, st, op, out, a, b, total, alpha, beta);
            else
                PM_LAUNCH(ew_flat_kernel<false>, dim3(grid_for(total)), dim3(256), 0, st, op, out, a, b, total, alpha, beta);
            return launch_ok();
        }
    }
    if (!b && H > 1 && W > 1 && vo.s[3] == 1 && vo.s[2] == W && va.s[2] == 1 && va.s[3] == H && (long)N * C < 65536 &&
        (N == 1 || (vo.s[0] >= (long)H * W && va.s[0] >= (long)H * W)) && (C == 1 || (vo.s[1] >= (long)H * W && va.s[1] >= (long)H * W))) {
        const dim3 grid((unsigned)((W + 31) / 32), (unsigned)((H + 31) / 32), (unsigned)(N * C));
        PM_LAUNCH(ew_transpose_kernel, grid, dim3(256), 0, st, op, out, a, C, H, W, vo.s[0], vo.s[1], va.s[0], va.s[1], alpha, beta);
        return launch_ok();
    }
    const bool small = total < (1L << 31);
    if (cfast && (C & 3) == 0 && vo.s[1] == 1 && va.s[1] == 1 && (!b || (sb && vb.s[1] == 1)) &&
        (((uintptr_t)out | (uintptr_t)a | (uintptr_t)b) & 15) == 0) {
        bool ok4 = true;        // every other stride a multiple of four elements: all 16-byte accesses are aligned
        const int dims[4] = {N, C, H, W};
        for (int i = 0; i < 4; ++i)
            if (i != 1 && dims[i] != 1 && (((vo.s[i] | va.s[i]) & 3) != 0 || (b && (vb.s[i] & 3) != 0))) ok4 = false;
        if (ok4) {
            const unsigned g = grid_for(total >> 2);
            if (small) PM_LAUNCH((ew_c4_kernel<int>), dim3(g), dim3(256), 0, st, op, out, vo, a, va, b, vb, N, C / 4, H, W, alpha, beta);
            else PM_LAUNCH((ew_c4_kernel<long>), dim3(g), dim3(256), 0, st, op, out, vo, a, va, b, vb, N, C / 4, H, W, alpha, beta);
            return launch_ok();
        }
    }
    if (!cfast && (W & 3) == 0 && vo.s[3] == 1 && va.s[3] == 1 && (!b || (sb && vb.s[3] == 1)) &&
        (((uintptr_t)out | (uintptr_t)a | (uintptr_t)b) & 15) == 0) {
        // planar views whose rows are contiguous (even / odd row splits and interleaves of the lifting, row windows): the
        // four-per-thread kernel with the row in the role of the channel vector
        const int dims[4] = {N, C, H, W};
        bool ok4 = true;
        for (int i = 0; i < 3; ++i)
            if (dims[i] != 1 && (((vo.s[i] | va.s[i]) & 3) != 0 || (b && (vb.s[i] & 3) != 0))) ok4 = false;
        if (ok4) {
            auto rows = [](const View &v) { View r; r.s[0] = v.s[0]; r.s[1] = 1; r.s[2] = v.s[1]; r.s[3] = v.s[2]; return r; };
            const unsigned g = grid_for(total >> 2);
            if (small) PM_LAUNCH((ew_c4_kernel<int>), dim3(g), dim3(256), 0, st, op, out, rows(vo), a, rows(va), b, rows(vb), N, W / 4, C, H, alpha, beta);
            else PM_LAUNCH((ew_c4_kernel<long>), dim3(g), dim3(256), 0, st, op, out, rows(vo), a, rows(va), b, rows(vb), N, W / 4, C, H, alpha, beta);
            return launch_ok();
        }
    }
    if (cfast) {
        if (small) PM_LAUNCH((ew_kernel<true, int>), dim3(grid_for(total)), dim3(256), 0, st, op, out, vo, a, va, b, vb, N, C, H, W, alpha, beta);
        else PM_LAUNCH((ew_kernel<true, long>), dim3(grid_for(total)), dim3(256), 0, st, op, out, vo, a, va, b, vb, N, C, H, W, alpha, beta);
    } else {
        if (small) PM_LAUNCH((ew_kernel<false, int>), dim3(grid_for(total)), dim3(256), 0, st, op, out, vo, a, va, b, vb, N, C, H, W, alpha, beta);
        else PM_LAUNCH((ew_kernel<false, long>), dim3(grid_for(total)), dim3(256), 0, st, op, out, vo, a, va, b, vb, N, C, H, W, alpha, beta);
    }
    return launch_ok();
}

extern "C" int pmctf_spynet_pack8_f32(const float *im1, const float *warped, const float *flow_up, float *out, int H,
                                      int W, void *stream) {
    if (!im1 || !warped || !flow_up || !out || H <= 0 || W <= 0) return PMCTF_EINVAL;
    PM_LAUNCH(spynet_pack8_kernel, dim3(grid_for((long)H * W)), dim3(256), 0, (hipStream_t)stream, im1, warped,
                       flow_up, out, (long)H * W);
    return launch_ok();
}

extern "C" int pmctf_lift_skip3_f32(const float *x, float *y, int NC, int H, int W, float w0, float w1, float w2,
                                    float bias, int sum_rule, void *stream) {
    if (!x || !y || NC <= 0 || H < 2 || W <= 0 || (sum_rule != 0 && sum_rule != 1)) return PMCTF_EINVAL;
    PM_LAUNCH(lift_skip3_kernel, dim3(grid_for((long)NC * H * W)), dim3(256), 0, (hipStream_t)stream, x, y, NC,
                       H, W, w0, w1, w2, bias, sum_rule);
    return launch_ok();
}

extern "C" int pmctf_nearest_up2_nhwc_f32(const float *x, float *y, int N, int H, int W, int C, void *stream) {
    if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0) return PMCTF_EINVAL;
    if ((C & 3) == 0) {
        PM_LAUNCH(nearest_up2_vec4_kernel, dim3(grid_for((long)N * H * W * (C >> 2))), dim3(256), 0, (hipStream_t)stream, x,
                  y, N, H, W, C);
        return launch_ok();
    }
    PM_LAUNCH(nearest_up2_kernel, dim3(grid_for((long)N * H * W * C * 4)), dim3(256), 0, (hipStream_t)stream, x,
                       y, N, H, W, C);
    return launch_ok();
}

extern "C" int pmctf_pixel_shuffle2_nhwc_f32(const float *x, float *y, int N, int H, int W, int C, int act, float slope,
                                             void *stream) {
    if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0) return PMCTF_EINVAL;
    if ((C & 3) == 0) {
        PM_LAUNCH(pixel_shuffle2_vec4_kernel, dim3(grid_for((long)N * H * W * (C >> 2))), dim3(256), 0, (hipStream_t)stream,
                  x, y, N, H, W, C, act, slope);
        return launch_ok();
    }
    PM_LAUNCH(pixel_shuffle2_kernel, dim3(grid_for((long)N * H * W * C * 4)), dim3(256), 0, (hipStream_t)stream,
                       x, y, N, H, W, C, act, slope);
    return launch_ok();
}

extern "C" int pmctf_ffn3_mix_f32(const float *x, float *y, int64_t P, int C, void *stream) {
    if (!x || !y || P <= 0 || C <= 0) return PMCTF_EINVAL;
    PM_LAUNCH(ffn3_mix_kernel, dim3(grid_for(P * C)), dim3(256), 0, (hipStream_t)stream, x, y, (long)P, C);
    return launch_ok();
}

extern "C" int pmctf_lstm_gates_f32(const float *xh, const float *cell, float *cell_out, float *hid_out, int64_t P,
                                    int C, int Ccell, void *stream) {
    return pmctf_lstm_gates_aten_f32(xh, cell, cell_out, hid_out, P, C, Ccell, P, 1, 0, stream);
}

extern "C" int pmctf_lstm_gates_aten_f32(const float *xh, const float *cell, float *cell_out, float *hid_out, int64_t P,
                                         int C, int Ccell, int64_t HW, int ref_planes, int aten_threads, void *stream) {
    if (!xh || !cell || !cell_out || !hid_out || P <= 0 || C <= 0 || (Ccell != 1 && Ccell != C) || aten_threads < 0 ||
        (aten_threads > 0 && (HW <= 0 || ref_planes <= 0 || P % HW || (P / HW) % ref_planes)))
        return PMCTF_EINVAL;
    PM_LAUNCH(lstm_gates_kernel, dim3(grid_for(P * C)), dim3(256), 0, (hipStream_t)stream, xh, cell, cell_out,
                       hid_out, (long)P, C, Ccell, (long)HW, ref_planes, aten_threads);
    return launch_ok();
}

extern "C" int pmctf_fourstep_quant_f32(const float *x, const float *params, float *so_far, int16_t *sym, int16_t *idx,
                                        int N, int H, int W, int k, int params_sub, float log_scale_min,
                                        float log_scale_step, void *stream) {
    if (!x || !params || !so_far || !sym || !idx || N <= 0 || H <= 0 || W <= 0 || k < 0 || k > 3) return PMCTF_EINVAL;
    if (params_sub && ((H | W) & 1)) return PMCTF_EINVAL;
    PM_LAUNCH(fourstep_quant_kernel, dim3(grid_for((long)N * H * W)), dim3(256), 0, (hipStream_t)stream, x,
                       params, so_far, sym, idx, N, H, W, k, params_sub, log_scale_min, log_scale_step);
    return launch_ok();
}

extern "C" int pmctf_ll_quant_f32(const float *ll, const float *params, float *ll_hat, int16_t *sym, int16_t *idx,
                                  int64_t total, int planes, float log_scale_min, float log_scale_step, void *stream) {
    if (!ll || !params || !ll_hat || !sym || !idx || total <= 0 || planes < 0 || (planes > 0 && total % planes))
        return PMCTF_EINVAL;
    PM_LAUNCH(ll_quant_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, ll, params, ll_hat, sym,
                       idx, (long)total, planes, log_scale_min, log_scale_step);
    return launch_ok();
}

extern "C" int pmctf_z_symbols_f32(const float *z, float *z_hat, int16_t *sym, int16_t *idx, int HW, int C,
                                   void *stream) {
    if (!z || !z_hat || !sym || !idx || HW <= 0 || C <= 0) return PMCTF_EINVAL;
    PM_LAUNCH(z_symbols_kernel, dim3(grid_for((long)HW * C)), dim3(256), 0, (hipStream_t)stream, z, z_hat, sym,
                       idx, HW, C);
    return launch_ok();
}

extern "C" int pmctf_mv_fourpart_step_f32(const float *y, const float *common, const float *sp, float *so_far,
                                          int16_t *sym, int16_t *idx, int H, int W, int t, float log_scale_min,
                                          float log_scale_step, void *stream) {
    if (!y || !common || !so_far || !sym || !idx || H <= 0 || W <= 0 || t < 0 || t > 3 || (t > 0 && !sp))
        return PMCTF_EINVAL;
    PM_LAUNCH(mv_fourpart_kernel, dim3(grid_for((long)H * W * 64)), dim3(256), 0, (hipStream_t)stream, y, common,
                       sp, so_far, sym, idx, H, W, t, log_scale_min, log_scale_step);
    return launch_ok();
}

extern "C" int pmctf_mv_dequant_f32(const float *so_far, const float *common, float *y_hat, int64_t HW, void *stream) {
    if (!so_far || !common || !y_hat || HW <= 0) return PMCTF_EINVAL;
    PM_LAUNCH(mv_dequant_kernel, dim3(grid_for(HW * 64)), dim3(256), 0, (hipStream_t)stream, so_far, common,
                       y_hat, (long)HW);
    return launch_ok();
}
