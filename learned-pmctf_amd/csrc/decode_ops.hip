// decode_ops.hip — decoder-side kernels of the pMCTF path.
//
//  * pmctf_ll_ar_decode_f32: the sequential, per-position autoregressive decode of the LL subband
//    (pWave._decompress_subband_ar + ContextFusionSubband.forward_sequential, pWave.py:557-584,
//    context_fusion.py:140-204) as ONE persistent workgroup: per position it evaluates the masked-conv network
//    on the causal neighbourhood (activations of earlier positions are kept in HBM/L2-resident NHWC buffers),
//    decodes the position's symbols from the rANS stream IN the kernel (the stream order is raster order, so
//    nothing else can run ahead) and writes ll_hat.  The entropy-coder state is handed in and out so that the
//    host decoder continues with the following subbands.
//  * index / dequantisation kernels of the four-step and four-part decompress
//    (context_fusion_4step.py:196-249, four_part_prior.py:217-280).
//
// Arithmetic: PM-F32.  Each output channel is one sequential fmaf chain in the spec's order (16-channel chunks,
// taps in raster order, channel ascending); masked (zero-weight) and out-of-image taps are skipped, which leaves
// the chain's value unchanged, so the parameters equal those of the encoder's one-shot masked convolutions bit for bit.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <mutex>
#include <type_traits>
#include "pm_device_math.h"
#include "launch.h"
#include "../../include/pmctf_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

inline int launch_ok() { return pm_launch_status(); }
inline unsigned grid_for(long n, int bs = 256, unsigned cap = 16384) {
    long b = (n + bs - 1) / bs;
    if (b < 1) b = 1;
    return (unsigned)(b > cap ? cap : b);
}

constexpr int NF = 128;           // features of ContextFusionSubband
constexpr int TB = 5;             // causal taps of a type-B 3x3 mask: (0,0)(0,1)(0,2)(1,0)(1,1)

struct LLArgs {
    const float *w;               // packed weights, see pmctf_ll_ar_pack_weights
    const uint32_t *stream;       // rANS words (after the 1-byte stream header), device copy
    long n_words;
    unsigned long long x0;        // decoder state on entry
    long pos0;                    // next word index on entry
    const int32_t *cdf;           // [256][cols]
    const int32_t *sizes, *offsets;
    int cols;
    float lmin, lstep;
    float *ll_out;                // [N][H][W]
    float *bufs;                  // 5 activation buffers [N][(H+1)][(W+2)][NF], zero-initialised by the caller
    int N, H, W;
    unsigned long long *state_out;  // [0] = x, [1] = word position, [2] = error flag
    // summation rules (include/pmctf_hip.h PMCTF_SUM_*), the same the encoder's one-shot network runs under:
    int blocks;                   // masked 3x3 layers: 0 = one chain from the bias, 1 = per-16-channel-block sums ("blocks")
    unsigned head_mask[3];        // 1x1 head layers convs.0/1/2: bit cb set = a reduction block ("reduce-B") ends before
    int head_b[3];                // 16-channel chunk cb; head_b = B (0: one chain)
};

__host__ __device__ inline long ll_scratch_acts_dev(int N, int H, int W) { return (long)5 * N * (H + 1) * (W + 2) * NF; }

// reduce-B over 128 input channels: which 16-channel chunks start a new block
inline unsigned reduce_mask(int B) {
    unsigned m = 0;
    if (B >= 16 && B < NF)
        for (int cb = 1; cb < NF / 16; ++cb)
            if ((cb * 16) % B == 0) m |= 1u << cb;
    return m;
}

// offsets (in floats) inside the packed weight blob
constexpr long W_L0 = 0;                                  // [4 taps][NF]      maskedConv1 (type A, Cin = 1)
constexpr long B_L0 = W_L0 + 4 * NF;                      // [NF]
constexpr long W_MB = B_L0 + NF;                          // 5 type-B layers, each [8 chunks][5 taps][16][NF] + bias [NF]
constexpr long SZ_MB = (long)8 * TB * 16 * NF + NF;
constexpr long W_P0 = W_MB + 5 * SZ_MB;                   // convs.0: [NF k][NF] + bias
constexpr long W_P1 = W_P0 + (long)NF * NF + NF;          // convs.1
constexpr long W_P2 = W_P1 + (long)NF * NF + NF;          // convs.2: [NF k][2] + bias[2]
constexpr long W_V1_TOTAL = W_P2 + NF * 2 + 2;
// ---- second layout in the same blob, read by the streaming kernel (ll_ar_stream_kernel): the weights of the five
// type-B layers and of the two 128 -> 128 head layers as ONE linear stream of float4 groups [group][co][4]
// (k = 4 * group + j in the chain order of the layer), so that a thread fetches four consecutive chain weights of its
// output channel with one 16-byte load and the whole position is a single forward sweep over 1.7 MB
constexpr int GROUPS_B = 8 * TB * 16 / 4;                 // 160 groups per type-B layer
constexpr int GROUPS_D = NF / 4;                          // 32 groups per dense layer ...
constexpr int GROUPS_DP = GROUPS_D + 8;                   // ... stored as 40 (the last 8 are zero padding, never used)
constexpr int GROUPS_TOTAL = 5 * GROUPS_B + 2 * GROUPS_DP; // 880
constexpr long S_W = W_V1_TOTAL;                          // [GROUPS_TOTAL][NF][4]
constexpr long S_BIAS = S_W + (long)GROUPS_TOTAL * NF * 4;   // [7][NF]: five type-B layers, two dense layers
constexpr long S_P2 = S_BIAS + 7 * NF;                    // convs.2 as [2][NF] + bias[2]
constexpr long W_TOTAL_V2 = S_P2 + 2 * NF + 2;
// ---- third layout (the row-wise decode under rule "blocks", ll_ar_row_kernel / ll_ar_pre_kernel): a 16-channel chunk's
// chain runs over the taps (0,0) (0,1) (0,2) of the row above first, then (1,0) (1,1).  The first 48 terms depend on row
// h-1 only and are evaluated for a whole row at once (U_W: [layer][chunk][tap < 3][ci][co]); the sequential kernel streams
// the remaining 32 terms per chunk (R_W: [group][co][4] over k2 = 32 * chunk + 16 * (tap - 3) + ci, then the two dense
// layers as in the second layout).
constexpr int GROUPS_R = 8 * 2 * 16 / 4;                  // 64 groups per type-B layer
constexpr int GROUPS_RTOTAL = 5 * GROUPS_R + 2 * GROUPS_DP;  // 400
constexpr long R_W = W_TOTAL_V2;                          // [GROUPS_RTOTAL][NF][4]
constexpr long U_W = R_W + (long)GROUPS_RTOTAL * NF * 4;  // [5][8][3][16][NF]
constexpr long W_TOTAL = U_W + (long)5 * 8 * 3 * 16 * NF;

__device__ __forceinline__ float leaky02(float v) { return v > 0.0f ? v : v * 0.2f; }

// type-B masked 3x3, 128 -> 128, at one position: act[t][ch] in LDS (t = causal tap), one output channel per thread
// blocks: every 16-channel chunk's sum starts from zero; (S_0 + bias) + S_1 + ... (rule "blocks")
__device__ __forceinline__ float masked_b(const float *__restrict__ wl, const float *act, int co, int blocks) {
    const float bias = wl[(long)8 * TB * 16 * NF + co];
    float acc = blocks ? 0.0f : bias, tot = 0.0f;
    const float *wp = wl + co;
#pragma unroll 1
    for (int cb = 0; cb < 8; ++cb) {
#pragma unroll
        for (int t = 0; t < TB; ++t) {
            const float *a = act + t * NF + cb * 16;
#pragma unroll
            for (int ci = 0; ci < 16; ++ci) acc = __builtin_fmaf(a[ci], wp[(long)((cb * TB + t) * 16 + ci) * NF], acc);
        }
        if (blocks) {
            tot = cb == 0 ? acc + bias : tot + acc;
            acc = 0.0f;
        }
    }
    return blocks ? tot : acc;
}

// mask / B: "reduce-B" of a 1x1 layer (the first block's chain starts at the bias, later blocks at zero, block results
// added in turn); mask = 0: one chain
__device__ __forceinline__ float dense128(const float *__restrict__ wl, const float *act, int co, int nout, unsigned mask,
                                          int B) {
    float acc = wl[(long)NF * nout + co], tot = 0.0f;
    const float *wp = wl + co;
#pragma unroll 1
    for (int cb = 0; cb < 8; ++cb) {
        if ((mask >> cb) & 1u) {
            tot = cb * 16 == B ? acc : tot + acc;
            acc = 0.0f;
        }
#pragma unroll
        for (int ci = 0; ci < 16; ++ci) acc = __builtin_fmaf(act[cb * 16 + ci], wp[(long)(cb * 16 + ci) * nout], acc);
    }
    return mask ? tot + acc : acc;
}

constexpr int LL_MAX_PLANES = 4;          // one 128-thread group per plane of the stream (Y: 1, UV: 2, RGB: 3)

__global__ __launch_bounds__(128 * LL_MAX_PLANES) void ll_ar_decode_kernel(LLArgs a) {
    __shared__ float act[LL_MAX_PLANES][TB * NF];     // per plane: gathered causal activations of the current layer
    __shared__ float vec[LL_MAX_PLANES][NF];          // per plane: a 128-vector passed between 1x1 layers
    __shared__ float prm[LL_MAX_PLANES][2];           // per plane: (scale, mean)
    __shared__ float newval[LL_MAX_PLANES];
    const int tid = threadIdx.x;
    const int plane = tid >> 7;
    const int co = tid & (NF - 1);
    const bool live = plane < a.N;
    const int PW = a.W + 2, PH = a.H + 1;                 // buffers carry one zero row on top and one zero column each side
    const long bufsz = (long)a.N * PH * PW * NF;
    float *buf[5];
    for (int i = 0; i < 5; ++i) buf[i] = a.bufs + i * bufsz + (long)(live ? plane : 0) * PH * PW * NF;
    float *cur = a.ll_out + (long)(live ? plane : 0) * a.H * a.W;
    const float *w = a.w;
    // L0 weights in registers
    const float w00 = w[W_L0 + 0 * NF + co], w01 = w[W_L0 + 1 * NF + co], w02 = w[W_L0 + 2 * NF + co],
                w10 = w[W_L0 + 3 * NF + co], b0 = w[B_L0 + co];
    unsigned long long x = a.x0;
    long pos = a.pos0;
    int err = 0;

    for (int h = 0; h < a.H; ++h) {
        for (int wq = 0; wq < a.W; ++wq) {
            // ---- maskedConv1 (type A) on the decoded plane
            float v00 = 0.f, v01 = 0.f, v02 = 0.f, v10 = 0.f;
            if (live) {
                if (h > 0) {
                    if (wq > 0) v00 = cur[(long)(h - 1) * a.W + wq - 1];
                    v01 = cur[(long)(h - 1) * a.W + wq];
                    if (wq + 1 < a.W) v02 = cur[(long)(h - 1) * a.W + wq + 1];
                }
                if (wq > 0) v10 = cur[(long)h * a.W + wq - 1];
            }
            float t = a.blocks ? 0.0f : b0;           // one input channel = one block: bias last under "blocks"
            t = __builtin_fmaf(v00, w00, t);
            t = __builtin_fmaf(v01, w01, t);
            t = __builtin_fmaf(v02, w02, t);
            t = __builtin_fmaf(v10, w10, t);
            if (a.blocks) t = t + b0;
            const float conv1 = t;
            // position (h, wq) lives at buffer row h+1, column wq+1
            const long cpos = ((long)(h + 1) * PW + (wq + 1)) * NF + co;
            const long up = cpos - (long)PW * NF;
            float xin = conv1;
            // ---- two masked residual blocks, then maskedConv2: five type-B layers over buffers 0..4
#pragma unroll 1
            for (int layer = 0; layer < 5; ++layer) {
                float *bb = buf[layer];
                if (live) {
                    bb[cpos] = xin;
                    act[plane][0 * NF + co] = bb[up - NF];
                    act[plane][1 * NF + co] = bb[up];
                    act[plane][2 * NF + co] = bb[up + NF];
                    act[plane][3 * NF + co] = bb[cpos - NF];
                    act[plane][4 * NF + co] = xin;
                }
                __syncthreads();
                float o = 0.f;
                if (live) o = masked_b(w + W_MB + layer * SZ_MB, act[plane], co, a.blocks);
                __syncthreads();
                if (layer == 0 || layer == 2) {            // conv1 of a residual block: leaky, keep block input in xres
                    vec[plane][co] = xin;                   // (own slot: no race)
                    xin = leaky02(o);
                } else if (layer == 1 || layer == 3) {      // conv2 of a residual block: + block input
                    xin = o + vec[plane][co];
                    if (layer == 3) xin = xin + conv1;      // x = x + conv1 before maskedConv2 (context_fusion.py:119)
                } else {
                    xin = leaky02(o);                       // maskedConv2 + lrelu
                }
            }
            // ---- 1x1 head: 128 -> 128 -> 128 -> 2
            if (live) act[plane][co] = xin;
            __syncthreads();
            float p0 = 0.f;
            if (live) p0 = leaky02(dense128(w + W_P0, act[plane], co, NF, a.head_mask[0], a.head_b[0]));
            __syncthreads();
            if (live) act[plane][co] = p0;
            __syncthreads();
            float p1 = 0.f;
            if (live) p1 = leaky02(dense128(w + W_P1, act[plane], co, NF, a.head_mask[1], a.head_b[1]));
            __syncthreads();
            if (live) act[plane][co] = p1;
            __syncthreads();
            if (live && co < 2) prm[plane][co] = dense128(w + W_P2, act[plane], co, 2, a.head_mask[2], a.head_b[2]);
            __syncthreads();
            // ---- entropy decode (wave 0, uniform): plane 0 then plane 1 (pWave.py:566-575 with B planes)
            if (tid < 64) {
                for (int pl = 0; pl < a.N; ++pl) {
                    float s = prm[pl][0];
                    const float mean = prm[pl][1];
                    s = s < 1e-5f ? 1e-5f : s;
                    float iv = (pm::logf_(s) - a.lmin) / a.lstep;
                    iv = iv >= 0.0f ? iv : 0.0f;          // also maps NaN (corrupt stream) to row 0
                    iv = iv > 255.0f ? 255.0f : iv;
                    const int row = (int)iv;
                    const int32_t *cd = a.cdf + (long)row * a.cols;
                    const int size = a.sizes[row];
                    const int max_value = size - 2;
                    const unsigned cum = (unsigned)(x & 0xFFFFull);
                    // s = (number of entries <= cum) - 1, entries 0..size-1
                    int cnt = 0;
                    for (int base = 0; base < size; base += 64) {
                        const int i = base + tid;
                        const bool le = i < size && (unsigned)cd[i] <= cum;
                        cnt += __builtin_popcountll(__ballot(le));
                    }
                    const int sidx = cnt - 1;
                    const unsigned start = (unsigned)cd[sidx], freq = (unsigned)(cd[sidx + 1] - cd[sidx]);
                    x = (unsigned long long)freq * (x >> 16) + (x & 0xFFFFull) - start;
                    if (x < (1ull << 31)) {
                        if (pos < a.n_words) x = (x << 32) | a.stream[pos]; else err = 1;
                        ++pos;
                    }
                    int value = sidx;
                    if (value == max_value) {               // bypass digits (rans.cpp:303-325)
                        auto bits4 = [&]() -> int {
                            const int val = (int)(x & 15ull);
                            x >>= 4;
                            if (x < (1ull << 31)) {
                                if (pos < a.n_words) x = (x << 32) | a.stream[pos]; else err = 1;
                                ++pos;
                            }
                            return val;
                        };
                        int val = bits4();
                        int n_bypass = val;
                        while (val == 15) { val = bits4(); n_bypass += val; }
                        int raw = 0;
                        for (int j = 0; j < n_bypass; ++j) raw |= bits4() << (j * 4);
                        value = raw >> 1;
                        if (raw & 1) value = -value - 1; else value += max_value;
                    }
                    const float q = (float)(short)(value + a.offsets[row]);
                    if (tid == 0) newval[pl] = __builtin_rintf(q + mean);
                }
            }
            __syncthreads();
            if (live && co == 0) cur[(long)h * a.W + wq] = newval[plane];
            __syncthreads();
        }
    }
    if (tid == 0) {
        a.state_out[0] = x;
        a.state_out[1] = (unsigned long long)pos;
        a.state_out[2] = (unsigned long long)err;
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Streaming form of the sequential LL decode.  What bounds a position is (a) per output channel one fmaf chain of 640
// terms per type-B layer in the spec's order — it cannot be split — and (b) 1.7 MB of weights through one CU's L2 port.
// The first kernel above paid for much more: five dependent global round trips per position for the causal activations,
// weight loads issued chunk by chunk with nothing in flight behind them, the CDF row of every symbol fetched from L2.
// Here:
//  * 128 threads, one per output channel, each running the chains of ALL planes of the stream (chroma: two chains per
//    weight, the weights are fetched once);
//  * the causal activations never make a round trip on the critical path: a layer's input at (h, w-1) is the value this
//    thread produced one position ago, the three values of row h-1 are a four-slot ring whose next element is requested a
//    full position ahead (the scratch buffers are per-channel private: a thread only reads what it wrote itself);
//  * the weights are one linear stream of 16-byte loads through a ring of five register blocks (four blocks = 128 chain
//    terms, 64 KB per workgroup, in flight), which never drains: the head of the next position is requested while the
//    symbol is decoded;
//  * the CDF table (105 KB), the biases, the per-layer state and the 128 -> 2 head sit in LDS.
// Same chains in the same order: bit-identical to the first kernel and to the encoder's one-shot masked convolutions.
template <int NP>
__global__ __launch_bounds__(NF) void ll_ar_stream_kernel(LLArgs a) {
    // float4 groups per ring block: 8 (32 chain terms) for one plane, 4 for two (the same time per block either way); the
    // ring holds 40 groups either way: 72 KB of weights in flight per workgroup, what it takes to keep the L2 port busy
    constexpr int BLK = NP == 1 ? 8 : 4;
    constexpr int RING = NP == 1 ? 5 : 10;
    constexpr int BLOCKS_B = GROUPS_B / BLK;              // 20 or 40: a multiple of RING
    constexpr int BLOCKS_D = GROUPS_DP / BLK;             // 5 or 10 (the last fifth is padding that is never read)
    constexpr int BLOCKS_TOTAL = 5 * BLOCKS_B + 2 * BLOCKS_D;
    static_assert(BLOCKS_B % RING == 0 && BLOCKS_D % RING == 0, "a layer must start at ring slot 0");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;                          // = output channel
    const int W = a.W, H = a.H;
    // LDS carve-up
    int32_t *l_cdf = (int32_t *)smem;                     // [256][cols]
    int32_t *l_sizes = l_cdf + 256 * a.cols;
    int32_t *l_offs = l_sizes + 256;
    float *l_act = (float *)(l_offs + 256);               // [2][NP][TB * NF]
    float *l_left = l_act + 2 * NP * TB * NF;             // [5][NP][NF]   layer input at (h, w-1)
    float *l_up = l_left + 5 * NP * NF;                   // [5][NP][4][NF] layer inputs of row h-1, slot = column & 3
    float *l_bias = l_up + 5 * NP * 4 * NF;               // [7][NF]
    float *l_p2 = l_bias + 7 * NF;                        // [2][NF] + 2
    float *l_prm = l_p2 + 2 * NF + 2;                     // [NP][2]
    float *l_rows = l_prm + 2 * NP;                       // [NP][2][W + 2]: decoded values of rows h-1 / h, zero-padded
    for (int i = tid; i < 256 * a.cols; i += NF) l_cdf[i] = a.cdf[i];
    for (int i = tid; i < 256; i += NF) { l_sizes[i] = a.sizes[i]; l_offs[i] = a.offsets[i]; }
    for (int i = tid; i < NP * 2 * (W + 2); i += NF) l_rows[i] = 0.0f;
    for (int i = tid; i < 2 * NF + 2; i += NF) l_p2[i] = a.w[S_P2 + i];
    for (int i = tid; i < 7 * NF; i += NF) l_bias[i] = a.w[S_BIAS + i];
    const float *w = a.w;
    const bool blocks = a.blocks != 0;
    const float w00 = w[W_L0 + 0 * NF + tid], w01 = w[W_L0 + 1 * NF + tid], w02 = w[W_L0 + 2 * NF + tid],
                w10 = w[W_L0 + 3 * NF + tid], b0 = w[B_L0 + tid];
    const f32x4 *ws = (const f32x4 *)(w + S_W) + tid;     // group g of this channel: ws[g * NF]
    const long plane_sz = (long)H * W * NF;               // scratch: [5 layers][NP][H][W][NF]
    float *scr = a.bufs + tid;
    f32x4 ring[RING][BLK];
    int nb = 0;                                           // next block of the stream to request
    auto request = [&](int slot) {                        // slot is a compile-time constant at every use
        const f32x4 *src = ws + (long)nb * BLK * NF;
#pragma unroll
        for (int g = 0; g < BLK; ++g) ring[slot][g] = src[(long)g * NF];
        nb = nb + 1 == BLOCKS_TOTAL ? 0 : nb + 1;
    };
#pragma unroll
    for (int b = 0; b < RING - 1; ++b) request(b);
    unsigned long long x = a.x0;
    long pos = a.pos0;
    int err = 0;
    __syncthreads();

    for (int h = 0; h < H; ++h) {
        float *row_prev = l_rows + ((h + 1) & 1) * (W + 2);      // + p * 2 * (W + 2)
        float *row_cur = l_rows + (h & 1) * (W + 2);
        // the row starts: nothing to the left; columns -1, 0, 1 of row h-1 into their slots
#pragma unroll
        for (int l = 0; l < 5; ++l)
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const float *sc = scr + ((long)l * NP + p) * plane_sz + (long)(h - 1) * W * NF;
                float *up = l_up + ((l * NP + p) * 4) * NF + tid;
                l_left[(l * NP + p) * NF + tid] = 0.0f;
                up[3 * NF] = 0.0f;
                up[0 * NF] = h > 0 ? sc[0] : 0.0f;
                up[1 * NF] = (h > 0 && W > 1) ? sc[NF] : 0.0f;
            }
        for (int wq = 0; wq < W; ++wq) {
            // next element of the window over row h-1 (column wq+2): requested now, stored when this position is done
            float un[5][NP];
#pragma unroll
            for (int l = 0; l < 5; ++l)
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    un[l][p] = (h > 0 && wq + 2 < W) ? scr[((long)l * NP + p) * plane_sz + ((long)(h - 1) * W + wq + 2) * NF] : 0.0f;
            // ---- maskedConv1 (type A, 1 -> 128) on the decoded values
            float conv1[NP], xin[NP], xres[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const float *rp = row_prev + p * 2 * (W + 2) + wq, *rc = row_cur + p * 2 * (W + 2) + wq;
                float t = blocks ? 0.0f : b0;             // padded rows: column wq-1 is at index wq
                t = __builtin_fmaf(rp[0], w00, t);
                t = __builtin_fmaf(rp[1], w01, t);
                t = __builtin_fmaf(rp[2], w02, t);
                t = __builtin_fmaf(rc[0], w10, t);
                if (blocks) t = t + b0;
                conv1[p] = t;
                xin[p] = t;
                xres[p] = 0.0f;
            }
            const long spos = ((long)h * W + wq) * NF;
            const int s0 = (wq + 3) & 3, s1 = wq & 3, s2 = (wq + 1) & 3;      // slots of columns wq-1, wq, wq+1
            // ---- five type-B layers
#pragma unroll 1
            for (int layer = 0; layer < 5; ++layer) {
                float *A = l_act + (layer & 1) * NP * TB * NF;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    scr[((long)layer * NP + p) * plane_sz + spos] = xin[p];
                    float *Ap = A + p * TB * NF + tid;
                    const float *up = l_up + ((layer * NP + p) * 4) * NF + tid;
                    float *lf = l_left + (layer * NP + p) * NF + tid;
                    Ap[0 * NF] = up[s0 * NF];
                    Ap[1 * NF] = up[s1 * NF];
                    Ap[2 * NF] = up[s2 * NF];
                    Ap[3 * NF] = lf[0];
                    Ap[4 * NF] = xin[p];
                    lf[0] = xin[p];
                }
                __syncthreads();
                float acc[NP], tot[NP];
                const float lbias = l_bias[layer * NF + tid];
#pragma unroll
                for (int p = 0; p < NP; ++p) { acc[p] = blocks ? 0.0f : lbias; tot[p] = 0.0f; }
                // activations of a block are read from LDS one block ahead of their use (the chain must never wait for them)
                f32x4 ab[2][NP][BLK];
                auto read_act = [&](int blk, int buf) {
#pragma unroll
                    for (int g = 0; g < BLK; ++g) {
                        const int k = (blk * BLK + g) * 4;                 // chain index of the group's first weight
                        const int pair = k / 16, cb = pair / TB, t = pair % TB, ci = k % 16;
#pragma unroll
                        for (int p = 0; p < NP; ++p) ab[buf][p][g] = *(const f32x4 *)(A + p * TB * NF + t * NF + cb * 16 + ci);
                    }
                };
                read_act(0, 0);
#pragma unroll
                for (int blk = 0; blk < BLOCKS_B; ++blk) {
                    request((blk + RING - 1) % RING);
                    if (blk + 1 < BLOCKS_B) read_act(blk + 1, (blk + 1) & 1);
#pragma unroll
                    for (int g = 0; g < BLK; ++g) {
                        const int k0 = (blk * BLK + g) * 4;               // chain index: a 16-channel chunk is 80 terms
                        if (k0 > 0 && k0 % (TB * 16) == 0 && blocks) {    // rule "blocks": the chunk's sum is complete
#pragma unroll
                            for (int p = 0; p < NP; ++p) {
                                tot[p] = k0 == TB * 16 ? acc[p] + lbias : tot[p] + acc[p];
                                acc[p] = 0.0f;
                            }
                        }
                        const f32x4 wv = ring[blk % RING][g];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int p = 0; p < NP; ++p)                   // the planes' chains interleaved: independent
                                acc[p] = __builtin_fmaf(ab[blk & 1][p][g][j], wv[j], acc[p]);
                    }
                    // the chains are pure: without this the optimiser is free to sink a plane's 640 fmaf to the end of the
                    // layer and spill every operand they need on the way
#pragma unroll
                    for (int p = 0; p < NP; ++p) asm volatile("" : "+v"(acc[p]));
                }
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const float o = blocks ? tot[p] + acc[p] : acc[p];
                    if (layer == 0 || layer == 2) { xres[p] = xin[p]; xin[p] = leaky02(o); }
                    else if (layer == 1) xin[p] = o + xres[p];
                    else if (layer == 3) xin[p] = (o + xres[p]) + conv1[p];
                    else xin[p] = leaky02(o);
                }
            }
            // ---- head: 128 -> 128 -> 128 -> 2
#pragma unroll 1
            for (int d = 0; d < 2; ++d) {
                float *A = l_act + ((5 + d) & 1) * NP * TB * NF;
#pragma unroll
                for (int p = 0; p < NP; ++p) A[p * TB * NF + tid] = xin[p];
                __syncthreads();
                float acc[NP], tot[NP];
                const unsigned hmask = a.head_mask[d];
                const int hb = a.head_b[d];
#pragma unroll
                for (int p = 0; p < NP; ++p) { acc[p] = l_bias[(5 + d) * NF + tid]; tot[p] = 0.0f; }
#pragma unroll
                for (int blk = 0; blk < BLOCKS_D; ++blk) {
                    request((blk + RING - 1) % RING);
                    if (blk < GROUPS_D / BLK) {           // the fifth block is padding that keeps the ring aligned
#pragma unroll
                        for (int g = 0; g < BLK; ++g) {
                            const int k = (blk * BLK + g) * 4;
                            if (k % 16 == 0 && ((hmask >> (k / 16)) & 1u)) {      // "reduce-B": a block of the reduction ends
#pragma unroll
                                for (int p = 0; p < NP; ++p) {
                                    tot[p] = k == hb ? acc[p] : tot[p] + acc[p];
                                    acc[p] = 0.0f;
                                }
                            }
                            const f32x4 wv = ring[blk % RING][g];
#pragma unroll
                            for (int p = 0; p < NP; ++p) {
                                const f32x4 av = *(const f32x4 *)(A + p * TB * NF + k);
                                acc[p] = __builtin_fmaf(av[0], wv[0], acc[p]);
                                acc[p] = __builtin_fmaf(av[1], wv[1], acc[p]);
                                acc[p] = __builtin_fmaf(av[2], wv[2], acc[p]);
                                acc[p] = __builtin_fmaf(av[3], wv[3], acc[p]);
                            }
                        }
#pragma unroll
                        for (int p = 0; p < NP; ++p) asm volatile("" : "+v"(acc[p]));
                    }
                }
#pragma unroll
                for (int p = 0; p < NP; ++p) xin[p] = leaky02(hmask ? tot[p] + acc[p] : acc[p]);
            }
            {
                float *A = l_act + (7 & 1) * NP * TB * NF;
#pragma unroll
                for (int p = 0; p < NP; ++p) A[p * TB * NF + tid] = xin[p];
                __syncthreads();
                if (tid < 2 * NP) {                       // (plane, output) pairs: scale and mean of every plane
                    const int p = tid >> 1, o = tid & 1;
                    float acc = l_p2[2 * NF + o], tot = 0.0f;
                    const float *wv = l_p2 + o * NF, *av = A + p * TB * NF;
                    const unsigned hmask = a.head_mask[2];
#pragma unroll 1
                    for (int cb = 0; cb < NF / 16; ++cb) {
                        if ((hmask >> cb) & 1u) {
                            tot = cb * 16 == a.head_b[2] ? acc : tot + acc;
                            acc = 0.0f;
                        }
#pragma unroll
                        for (int k = cb * 16; k < cb * 16 + 16; ++k) acc = __builtin_fmaf(av[k], wv[k], acc);
                    }
                    l_prm[p * 2 + o] = hmask ? tot + acc : acc;
                }
                __syncthreads();
            }
            // ---- entropy decode (wave 0, uniform): plane after plane (pWave.py:566-575)
            if (tid < 64) {
#pragma unroll 1
                for (int pl = 0; pl < NP; ++pl) {
                    float s = l_prm[pl * 2 + 0];
                    const float mean = l_prm[pl * 2 + 1];
                    s = s < 1e-5f ? 1e-5f : s;
                    float iv = (pm::logf_(s) - a.lmin) / a.lstep;
                    iv = iv >= 0.0f ? iv : 0.0f;          // also maps NaN (corrupt stream) to row 0
                    iv = iv > 255.0f ? 255.0f : iv;
                    const int row = (int)iv;
                    const int32_t *cd = l_cdf + row * a.cols;
                    const int size = l_sizes[row];
                    const int max_value = size - 2;
                    const unsigned cum = (unsigned)(x & 0xFFFFull);
                    int cnt = 0;
                    for (int base = 0; base < size; base += 64) {
                        const int i = base + tid;
                        const bool le = i < size && (unsigned)cd[i] <= cum;
                        cnt += __builtin_popcountll(__ballot(le));
                    }
                    const int sidx = cnt - 1;
                    const unsigned start = (unsigned)cd[sidx], freq = (unsigned)(cd[sidx + 1] - cd[sidx]);
                    x = (unsigned long long)freq * (x >> 16) + (x & 0xFFFFull) - start;
                    if (x < (1ull << 31)) {
                        if (pos < a.n_words) x = (x << 32) | a.stream[pos]; else err = 1;
                        ++pos;
                    }
                    int value = sidx;
                    if (value == max_value) {               // bypass digits (rans.cpp:303-325)
                        auto bits4 = [&]() -> int {
                            const int val = (int)(x & 15ull);
                            x >>= 4;
                            if (x < (1ull << 31)) {
                                if (pos < a.n_words) x = (x << 32) | a.stream[pos]; else err = 1;
                                ++pos;
                            }
                            return val;
                        };
                        int val = bits4();
                        int n_bypass = val;
                        while (val == 15) { val = bits4(); n_bypass += val; }
                        int raw = 0;
                        for (int j = 0; j < n_bypass; ++j) raw |= bits4() << (j * 4);
                        value = raw >> 1;
                        if (raw & 1) value = -value - 1; else value += max_value;
                    }
                    const float q = (float)(short)(value + l_offs[row]);
                    if (tid == 0) {
                        const float v = __builtin_rintf(q + mean);
                        row_cur[pl * 2 * (W + 2) + wq + 1] = v;
                        a.ll_out[(long)pl * H * W + (long)h * W + wq] = v;
                    }
                }
            }
            // column wq+2 of row h-1 takes the slot column wq-2 had
#pragma unroll
            for (int l = 0; l < 5; ++l)
#pragma unroll
                for (int p = 0; p < NP; ++p) l_up[((l * NP + p) * 4 + ((wq + 2) & 3)) * NF + tid] = un[l][p];
            __syncthreads();
        }
    }
    if (tid == 0) {
        a.state_out[0] = x;
        a.state_out[1] = (unsigned long long)pos;
        a.state_out[2] = (unsigned long long)err;
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Row-wise form of the sequential LL decode under rule "blocks".  A type-B layer's output at (h, w) is, per 16-channel
// chunk, ONE chain from zero over the taps (h-1, w-1) (h-1, w) (h-1, w+1) (h, w-1) (h, w); the chunk sums are added in
// turn.  The first 48 terms of every chunk's chain depend on row h-1 only: ll_ar_pre_kernel evaluates them for a whole
// row at once (all CUs, 8 positions per workgroup so that a weight is fetched once per 8 chains), the sequential kernel
// continues each chain from that prefix with the 32 terms of the two taps of row h — the same fmaf sequence, the same
// bits, 2.25x fewer weights to stream per position (0.79 instead of 1.7 MB) and chains of 256 instead of 640 terms.
// One launch pair per row; the coder state travels in state_out.
template <int NP>
__global__ __launch_bounds__(NF) void ll_ar_pre_kernel(LLArgs a, int h) {
    __shared__ float act[10][NF];
    const int tid = threadIdx.x;
    const int l = blockIdx.y / NP, p = blockIdx.y % NP, w0 = blockIdx.x * 8;
    const int W = a.W;
    const long plane_sz = (long)a.H * W * NF;
    const float *src = a.bufs + ((long)l * NP + p) * plane_sz + (long)(h - 1) * W * NF;
    for (int i = tid; i < 10 * NF; i += NF) {
        const int col = w0 - 1 + i / NF;
        act[i / NF][i % NF] = (col >= 0 && col < W) ? src[(long)col * NF + (i % NF)] : 0.0f;
    }
    __syncthreads();
    const float *wu = a.w + U_W + (long)l * 8 * 3 * 16 * NF + tid;
    float *P = a.bufs + ll_scratch_acts_dev(a.N, a.H, W) + (((long)l * NP + p) * W) * 8 * NF + tid;
#pragma unroll 1
    for (int cb = 0; cb < 8; ++cb) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int ci = 0; ci < 16; ++ci) {
                const float wv = wu[(long)((cb * 3 + t) * 16 + ci) * NF];
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = __builtin_fmaf(act[j + t][cb * 16 + ci], wv, acc[j]);
            }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (w0 + j < W) P[((long)(w0 + j) * 8 + cb) * NF] = acc[j];
    }
}

template <int NP>
__global__ __launch_bounds__(NF) void ll_ar_row_kernel(LLArgs a, int h) {
    constexpr int BLK = 4;                                // float4 groups per ring block; 40 groups (80 KB) in flight
    constexpr int RING = 10;
    constexpr int BLOCKS_R = GROUPS_R / BLK;              // 8 or 16 blocks per type-B layer
    constexpr int BLOCKS_D = GROUPS_DP / BLK;             // 5 or 10 (the last fifth is padding that is never read)
    constexpr int BLOCKS_TOTAL = 5 * BLOCKS_R + 2 * BLOCKS_D;   // 50 or 100 per position
    static_assert(BLOCKS_TOTAL % RING == 0, "a position must start at ring slot 0");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;                          // = output channel
    const int W = a.W, H = a.H;
    int32_t *l_cdf = (int32_t *)smem;                     // [256][cols]
    int32_t *l_sizes = l_cdf + 256 * a.cols;
    int32_t *l_offs = l_sizes + 256;
    float *l_act = (float *)(l_offs + 256);               // [2][NP][2 * NF]: taps (h, w-1) and (h, w) of the layer's input
    float *l_left = l_act + 2 * NP * 2 * NF;              // [5][NP][NF]   layer input at (h, w-1)
    float *l_bias = l_left + 5 * NP * NF;                 // [7][NF]
    float *l_p2 = l_bias + 7 * NF;                        // [2][NF] + 2
    float *l_prm = l_p2 + 2 * NF + 2;                     // [NP][2]
    float *l_rows = l_prm + 2 * NP;                       // [NP][2][W + 2]: decoded values of rows h-1 / h, zero-padded
    for (int i = tid; i < 256 * a.cols; i += NF) l_cdf[i] = a.cdf[i];
    for (int i = tid; i < 256; i += NF) { l_sizes[i] = a.sizes[i]; l_offs[i] = a.offsets[i]; }
    for (int i = tid; i < NP * 2 * (W + 2); i += NF) {
        const int p = i / (2 * (W + 2)), r = (i / (W + 2)) & 1, c = i % (W + 2);
        l_rows[i] = (r == 0 && h > 0 && c >= 1 && c <= W) ? a.ll_out[(long)p * H * W + (long)(h - 1) * W + c - 1] : 0.0f;
    }
    for (int i = tid; i < 2 * NF + 2; i += NF) l_p2[i] = a.w[S_P2 + i];
    for (int i = tid; i < 7 * NF; i += NF) l_bias[i] = a.w[S_BIAS + i];
    for (int i = tid; i < 5 * NP * NF; i += NF) l_left[i] = 0.0f;
    const float *w = a.w;
    const float w00 = w[W_L0 + 0 * NF + tid], w01 = w[W_L0 + 1 * NF + tid], w02 = w[W_L0 + 2 * NF + tid],
                w10 = w[W_L0 + 3 * NF + tid], b0 = w[B_L0 + tid];
    const f32x4 *ws = (const f32x4 *)(w + R_W) + tid;     // group g of this channel: ws[g * NF]
    const long plane_sz = (long)H * W * NF;               // scratch: [5 layers][NP][H][W][NF]
    float *scr = a.bufs + tid;
    const float *pre = a.bufs + ll_scratch_acts_dev(a.N, H, W) + tid;      // [5][NP][W][8][NF]
    f32x4 ring[RING][BLK];
    int nb = 0;                                           // next block of the stream to request
    auto request = [&](int slot) {                        // slot is a compile-time constant at every use
        // nb is the same constant at a given site for every position; hidden from the optimiser, which otherwise keeps one
        // precomputed 64-bit address per site (100 sites: 200 registers, spilled)
        asm volatile("" : "+s"(nb));
        const f32x4 *src = ws + (long)nb * BLK * NF;
#pragma unroll
        for (int g = 0; g < BLK; ++g) ring[slot][g] = src[(long)g * NF];
        nb = nb + 1 == BLOCKS_TOTAL ? 0 : nb + 1;
    };
#pragma unroll
    for (int b = 0; b < RING - 1; ++b) request(b);
    unsigned long long x = h == 0 ? a.x0 : a.state_out[0];
    long pos = h == 0 ? a.pos0 : (long)a.state_out[1];
    int err = h == 0 ? 0 : (int)a.state_out[2];
    // the next word of the stream is fetched when its predecessor is consumed, a position or more before it is needed
    uint32_t nw = pos < a.n_words ? a.stream[pos] : 0u;
    const float *row_prev = l_rows, *row_cur_c = l_rows + (W + 2);
    float *row_cur = l_rows + (W + 2);
    (void)row_cur_c;
    // chunk prefixes of (layer 0, position 0)
    float pn[NP][8];
    auto fetch_prefix = [&](int layer, int wq) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int cb = 0; cb < 8; ++cb)
                pn[p][cb] = h > 0 ? pre[((((long)layer * NP + p) * W + wq) * 8 + cb) * NF] : 0.0f;
    };
    fetch_prefix(0, 0);
    __syncthreads();

    for (int wq = 0; wq < W; ++wq) {
        // ---- maskedConv1 (type A, 1 -> 128) on the decoded values: one input channel = one block, bias last
        float conv1[NP], xin[NP], xres[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const float *rp = row_prev + p * 2 * (W + 2) + wq, *rc = row_cur + p * 2 * (W + 2) + wq;
            float t = 0.0f;                               // padded rows: column wq-1 is at index wq
            t = __builtin_fmaf(rp[0], w00, t);
            t = __builtin_fmaf(rp[1], w01, t);
            t = __builtin_fmaf(rp[2], w02, t);
            t = __builtin_fmaf(rc[0], w10, t);
            t = t + b0;
            conv1[p] = t;
            xin[p] = t;
            xres[p] = 0.0f;
        }
        const long spos = ((long)h * W + wq) * NF;
        // ---- five type-B layers (the layer index is a compile-time constant: ring slots are registers)
        auto type_b = [&](auto layer_c) {
            constexpr int layer = decltype(layer_c)::value;
            float *A = l_act + (layer & 1) * NP * 2 * NF;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                scr[((long)layer * NP + p) * plane_sz + spos] = xin[p];
                float *lf = l_left + (layer * NP + p) * NF + tid;
                A[p * 2 * NF + tid] = lf[0];
                A[p * 2 * NF + NF + tid] = xin[p];
                lf[0] = xin[p];
            }
            __syncthreads();
            float pc[NP][8];
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) pc[p][cb] = pn[p][cb];
            // the next layer's (or the next position's first layer's) prefixes: requested now, needed a layer later
            if (layer < 4) fetch_prefix(layer + 1, wq);
            else if (wq + 1 < W) fetch_prefix(0, wq + 1);
            const float lbias = l_bias[layer * NF + tid];
            float acc[NP], tot[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) { acc[p] = 0.0f; tot[p] = 0.0f; }
#pragma unroll
            for (int blk = 0; blk < BLOCKS_R; ++blk) {
                request((layer * BLOCKS_R + blk + RING - 1) % RING);
#pragma unroll
                for (int g = 0; g < BLK; ++g) {
                    const int k0 = (blk * BLK + g) * 4;               // chain index among the 256 terms of rows h
                    const int cb = k0 / 32, t = (k0 % 32) / 16, ci = k0 % 16;
                    if (k0 % 32 == 0) {                                // a chunk starts: its chain continues the prefix
#pragma unroll
                        for (int p = 0; p < NP; ++p) {
                            if (cb == 1) tot[p] = acc[p] + lbias;
                            else if (cb > 1) tot[p] = tot[p] + acc[p];
                            acc[p] = pc[p][cb];
                        }
                    }
                    const f32x4 wv = ring[(layer * BLOCKS_R + blk) % RING][g];
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        const f32x4 av = *(const f32x4 *)(A + p * 2 * NF + t * NF + cb * 16 + ci);
                        acc[p] = __builtin_fmaf(av[0], wv[0], acc[p]);
                        acc[p] = __builtin_fmaf(av[1], wv[1], acc[p]);
                        acc[p] = __builtin_fmaf(av[2], wv[2], acc[p]);
                        acc[p] = __builtin_fmaf(av[3], wv[3], acc[p]);
                    }
                }
#pragma unroll
                for (int p = 0; p < NP; ++p) asm volatile("" : "+v"(acc[p]));
                __builtin_amdgcn_sched_barrier(0);     // keep a block's LDS reads with the block (hoisted, they spill)
            }
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const float o = tot[p] + acc[p];
                if (layer == 0 || layer == 2) { xres[p] = xin[p]; xin[p] = leaky02(o); }
                else if (layer == 1) xin[p] = o + xres[p];
                else if (layer == 3) xin[p] = (o + xres[p]) + conv1[p];
                else xin[p] = leaky02(o);
            }
        };
        type_b(std::integral_constant<int, 0>{});
        type_b(std::integral_constant<int, 1>{});
        type_b(std::integral_constant<int, 2>{});
        type_b(std::integral_constant<int, 3>{});
        type_b(std::integral_constant<int, 4>{});
        // ---- head: 128 -> 128 -> 128 -> 2
        auto dense = [&](auto d_c) {
            constexpr int d = decltype(d_c)::value;
            float *A = l_act + ((5 + d) & 1) * NP * 2 * NF;
#pragma unroll
            for (int p = 0; p < NP; ++p) A[p * 2 * NF + tid] = xin[p];
            __syncthreads();
            float acc[NP], tot[NP];
            const unsigned hmask = a.head_mask[d];
            const int hb = a.head_b[d];
#pragma unroll
            for (int p = 0; p < NP; ++p) { acc[p] = l_bias[(5 + d) * NF + tid]; tot[p] = 0.0f; }
#pragma unroll
            for (int blk = 0; blk < BLOCKS_D; ++blk) {
                request((5 * BLOCKS_R + d * BLOCKS_D + blk + RING - 1) % RING);
                if (blk < GROUPS_D / BLK) {               // the fifth block is padding that keeps the ring aligned
#pragma unroll
                    for (int g = 0; g < BLK; ++g) {
                        const int k = (blk * BLK + g) * 4;
                        if (k % 16 == 0 && ((hmask >> (k / 16)) & 1u)) {      // "reduce-B": a block of the reduction ends
#pragma unroll
                            for (int p = 0; p < NP; ++p) {
                                tot[p] = k == hb ? acc[p] : tot[p] + acc[p];
                                acc[p] = 0.0f;
                            }
                        }
                        const f32x4 wv = ring[(5 * BLOCKS_R + d * BLOCKS_D + blk) % RING][g];
#pragma unroll
                        for (int p = 0; p < NP; ++p) {
                            const f32x4 av = *(const f32x4 *)(A + p * 2 * NF + k);
                            acc[p] = __builtin_fmaf(av[0], wv[0], acc[p]);
                            acc[p] = __builtin_fmaf(av[1], wv[1], acc[p]);
                            acc[p] = __builtin_fmaf(av[2], wv[2], acc[p]);
                            acc[p] = __builtin_fmaf(av[3], wv[3], acc[p]);
                        }
                    }
#pragma unroll
                    for (int p = 0; p < NP; ++p) asm volatile("" : "+v"(acc[p]));
                __builtin_amdgcn_sched_barrier(0);     // keep a block's LDS reads with the block (hoisted, they spill)
                }
            }
#pragma unroll
            for (int p = 0; p < NP; ++p) xin[p] = leaky02(hmask ? tot[p] + acc[p] : acc[p]);
        };
        dense(std::integral_constant<int, 0>{});
        dense(std::integral_constant<int, 1>{});
        {
            float *A = l_act + (7 & 1) * NP * 2 * NF;
#pragma unroll
            for (int p = 0; p < NP; ++p) A[p * 2 * NF + tid] = xin[p];
            __syncthreads();
            if (tid < 2 * NP) {                           // (plane, output) pairs: scale and mean of every plane
                const int p = tid >> 1, o = tid & 1;
                float acc = l_p2[2 * NF + o], tot = 0.0f;
                const float *wv = l_p2 + o * NF, *av = A + p * 2 * NF;
                const unsigned hmask = a.head_mask[2];
#pragma unroll 1
                for (int cb = 0; cb < NF / 16; ++cb) {
                    if ((hmask >> cb) & 1u) {
                        tot = cb * 16 == a.head_b[2] ? acc : tot + acc;
                        acc = 0.0f;
                    }
#pragma unroll
                    for (int k = cb * 16; k < cb * 16 + 16; ++k) acc = __builtin_fmaf(av[k], wv[k], acc);
                }
                l_prm[p * 2 + o] = hmask ? tot + acc : acc;
            }
            __syncthreads();
        }
        // ---- entropy decode (wave 0, uniform): plane after plane (pWave.py:566-575)
        if (tid < 64) {
#pragma unroll 1
            for (int pl = 0; pl < NP; ++pl) {
                float s = l_prm[pl * 2 + 0];
                const float mean = l_prm[pl * 2 + 1];
                s = s < 1e-5f ? 1e-5f : s;
                float iv = (pm::logf_(s) - a.lmin) / a.lstep;
                iv = iv >= 0.0f ? iv : 0.0f;              // also maps NaN (corrupt stream) to row 0
                iv = iv > 255.0f ? 255.0f : iv;
                const int row = (int)iv;
                const int32_t *cd = l_cdf + row * a.cols;
                const int size = l_sizes[row];
                const int max_value = size - 2;
                const unsigned cum = (unsigned)(x & 0xFFFFull);
                int cnt = 0;
                for (int base = 0; base < size; base += 64) {
                    const int i = base + tid;
                    const bool le = i < size && (unsigned)cd[i] <= cum;
                    cnt += __builtin_popcountll(__ballot(le));
                }
                const int sidx = cnt - 1;
                const unsigned start = (unsigned)cd[sidx], freq = (unsigned)(cd[sidx + 1] - cd[sidx]);
                x = (unsigned long long)freq * (x >> 16) + (x & 0xFFFFull) - start;
                if (x < (1ull << 31)) {
                    if (pos < a.n_words) x = (x << 32) | nw; else err = 1;
                    ++pos;
                    nw = pos < a.n_words ? a.stream[pos] : 0u;
                }
                int value = sidx;
                if (value == max_value) {                   // bypass digits (rans.cpp:303-325)
                    auto bits4 = [&]() -> int {
                        const int val = (int)(x & 15ull);
                        x >>= 4;
                        if (x < (1ull << 31)) {
                            if (pos < a.n_words) x = (x << 32) | nw; else err = 1;
                            ++pos;
                            nw = pos < a.n_words ? a.stream[pos] : 0u;
                        }
                        return val;
                    };
                    int val = bits4();
                    int n_bypass = val;
                    while (val == 15) { val = bits4(); n_bypass += val; }
                    int raw = 0;
                    for (int j = 0; j < n_bypass; ++j) raw |= bits4() << (j * 4);
                    value = raw >> 1;
                    if (raw & 1) value = -value - 1; else value += max_value;
                }
                const float q = (float)(short)(value + l_offs[row]);
                if (tid == 0) {
                    const float v = __builtin_rintf(q + mean);
                    row_cur[pl * 2 * (W + 2) + wq + 1] = v;
                    a.ll_out[(long)pl * H * W + (long)h * W + wq] = v;
                }
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        a.state_out[0] = x;
        a.state_out[1] = (unsigned long long)pos;
        a.state_out[2] = (unsigned long long)err;
    }
}


// The same row kernel on 256 threads: under rule "blocks" the eight chunk sums of an output channel are independent chains,
// so TWO threads share a channel — chunks 0-3 and 4-7 — each streaming half of the layer's weights through its own register
// ring (twice the bytes in flight, half the chain length, all four SIMDs); the second thread hands its four sums over
// through LDS and the first adds all eight in order ((S_0 + bias) + S_1 + ... + S_7: the same additions).  The dense
// head layers are single chains and stay on the first 128 threads.
template <int NP, int HALF>
__device__ __forceinline__ void ll_ar_row2_body(const LLArgs &a, const int h, float *smem) {
    constexpr int BLK = 4, RING = 10;
    constexpr int BLOCKS_R = GROUPS_R / 2 / BLK;          // 8 blocks per type-B layer and half
    constexpr int BLOCKS_D = GROUPS_DP / BLK;             // 10 (the last fifth is padding that is never read)
    constexpr int BLOCKS_TOTAL = 5 * BLOCKS_R + (HALF == 0 ? 2 * BLOCKS_D : 0);   // 60 or 40 per position
    static_assert(BLOCKS_TOTAL % RING == 0, "a position must start at ring slot 0");
    const int tid = threadIdx.x & (NF - 1);               // = output channel
    const int W = a.W, H = a.H;
    int32_t *l_cdf = (int32_t *)smem;                     // [256][cols]
    int32_t *l_sizes = l_cdf + 256 * a.cols;
    int32_t *l_offs = l_sizes + 256;
    float *l_act = (float *)(l_offs + 256);               // [2][NP][2 * NF]: taps (h, w-1) and (h, w) of the layer's input
    float *l_left = l_act + 2 * NP * 2 * NF;              // [5][NP][NF]   layer input at (h, w-1)
    float *l_bias = l_left + 5 * NP * NF;                 // [7][NF]
    float *l_p2 = l_bias + 7 * NF;                        // [2][NF] + 2
    float *l_prm = l_p2 + 2 * NF + 2;                     // [NP][2]
    float *l_s = l_prm + 2 * NP;                          // [NP][4][NF]: chunk sums 4..7 on their way to the first half
    float *l_rows = l_s + NP * 4 * NF;                    // [NP][2][W + 2]: decoded values of rows h-1 / h, zero-padded
    {
        const int t2 = threadIdx.x;
        for (int i = t2; i < 256 * a.cols; i += 2 * NF) l_cdf[i] = a.cdf[i];
        for (int i = t2; i < 256; i += 2 * NF) { l_sizes[i] = a.sizes[i]; l_offs[i] = a.offsets[i]; }
        for (int i = t2; i < NP * 2 * (W + 2); i += 2 * NF) {
            const int p = i / (2 * (W + 2)), r = (i / (W + 2)) & 1, c = i % (W + 2);
            l_rows[i] = (r == 0 && h > 0 && c >= 1 && c <= W) ? a.ll_out[(long)p * H * W + (long)(h - 1) * W + c - 1] : 0.0f;
        }
        for (int i = t2; i < 2 * NF + 2; i += 2 * NF) l_p2[i] = a.w[S_P2 + i];
        for (int i = t2; i < 7 * NF; i += 2 * NF) l_bias[i] = a.w[S_BIAS + i];
        for (int i = t2; i < 5 * NP * NF; i += 2 * NF) l_left[i] = 0.0f;
    }
    const float *w = a.w;
    const float w00 = w[W_L0 + 0 * NF + tid], w01 = w[W_L0 + 1 * NF + tid], w02 = w[W_L0 + 2 * NF + tid],
                w10 = w[W_L0 + 3 * NF + tid], b0 = w[B_L0 + tid];
    const f32x4 *ws = (const f32x4 *)(w + R_W) + tid;     // group g of this channel: ws[g * NF]
    const long plane_sz = (long)H * W * NF;               // scratch: [5 layers][NP][H][W][NF]
    float *scr = a.bufs + tid;
    const float *pre = a.bufs + ll_scratch_acts_dev(a.N, H, W) + tid;      // [5][NP][W][8][NF]
    f32x4 ring[RING][BLK];
    int nb = 0;                                           // next block of this half's stream to request
    auto request = [&](int slot) {                        // slot is a compile-time constant at every use
        asm volatile("" : "+s"(nb));                      // (hidden from the optimiser: see ll_ar_row_kernel)
        // block nb of this half: layer nb / 8, groups 64 * layer + 32 * HALF + 4 * (nb % 8) ..; then the dense layers
        const int g0 = nb < 5 * BLOCKS_R ? (nb >> 3) * GROUPS_R + HALF * (GROUPS_R / 2) + (nb & 7) * BLK
                                         : 5 * GROUPS_R + (nb - 5 * BLOCKS_R) * BLK;
        const f32x4 *src = ws + (long)g0 * NF;
#pragma unroll
        for (int g = 0; g < BLK; ++g) ring[slot][g] = src[(long)g * NF];
        nb = nb + 1 == BLOCKS_TOTAL ? 0 : nb + 1;
    };
#pragma unroll
    for (int b = 0; b < RING - 1; ++b) request(b);
    unsigned long long x = h == 0 ? a.x0 : a.state_out[0];
    long pos = h == 0 ? a.pos0 : (long)a.state_out[1];
    int err = h == 0 ? 0 : (int)a.state_out[2];
    uint32_t nw = pos < a.n_words ? a.stream[pos] : 0u;
    const float *row_prev = l_rows;
    float *row_cur = l_rows + (W + 2);
    float pn[NP][4];                                      // chunk prefixes of this half's chunks
    auto fetch_prefix = [&](int layer, int wq) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                pn[p][c] = h > 0 ? pre[((((long)layer * NP + p) * W + wq) * 8 + 4 * HALF + c) * NF] : 0.0f;
    };
    fetch_prefix(0, 0);
    __syncthreads();

    for (int wq = 0; wq < W; ++wq) {
        float conv1[NP], xin[NP], xres[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const float *rp = row_prev + p * 2 * (W + 2) + wq, *rc = row_cur + p * 2 * (W + 2) + wq;
            float t = 0.0f;                               // padded rows: column wq-1 is at index wq
            t = __builtin_fmaf(rp[0], w00, t);
            t = __builtin_fmaf(rp[1], w01, t);
            t = __builtin_fmaf(rp[2], w02, t);
            t = __builtin_fmaf(rc[0], w10, t);
            t = t + b0;
            conv1[p] = t;
            xin[p] = t;
            xres[p] = 0.0f;
        }
        const long spos = ((long)h * W + wq) * NF;
        auto type_b = [&](auto layer_c) {
            constexpr int layer = decltype(layer_c)::value;
            float *A = l_act + (layer & 1) * NP * 2 * NF;
            if constexpr (HALF == 0) {
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    scr[((long)layer * NP + p) * plane_sz + spos] = xin[p];
                    float *lf = l_left + (layer * NP + p) * NF + tid;
                    A[p * 2 * NF + tid] = lf[0];
                    A[p * 2 * NF + NF + tid] = xin[p];
                    lf[0] = xin[p];
                }
            }
            __syncthreads();
            float S[NP][4];
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int c = 0; c < 4; ++c) S[p][c] = pn[p][c];
            if (layer < 4) fetch_prefix(layer + 1, wq);
            else if (wq + 1 < W) fetch_prefix(0, wq + 1);
#pragma unroll
            for (int blk = 0; blk < BLOCKS_R; ++blk) {
                request((layer * BLOCKS_R + blk + RING - 1) % RING);
#pragma unroll
                for (int g = 0; g < BLK; ++g) {
                    const int k0 = (blk * BLK + g) * 4;               // chain index among this half's 128 terms of row h
                    const int c = k0 / 32, t = (k0 % 32) / 16, ci = k0 % 16;
                    const f32x4 wv = ring[(layer * BLOCKS_R + blk) % RING][g];
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        const f32x4 av = *(const f32x4 *)(A + p * 2 * NF + t * NF + (4 * HALF + c) * 16 + ci);
                        S[p][c] = __builtin_fmaf(av[0], wv[0], S[p][c]);
                        S[p][c] = __builtin_fmaf(av[1], wv[1], S[p][c]);
                        S[p][c] = __builtin_fmaf(av[2], wv[2], S[p][c]);
                        S[p][c] = __builtin_fmaf(av[3], wv[3], S[p][c]);
                    }
                }
#pragma unroll
                for (int p = 0; p < NP; ++p)
#pragma unroll
                    for (int c = 0; c < 4; ++c) asm volatile("" : "+v"(S[p][c]));
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (HALF == 1) {
#pragma unroll
                for (int p = 0; p < NP; ++p)
#pragma unroll
                    for (int c = 0; c < 4; ++c) l_s[(p * 4 + c) * NF + tid] = S[p][c];
            }
            __syncthreads();
            if constexpr (HALF == 0) {
                const float lbias = l_bias[layer * NF + tid];
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    float o = S[p][0] + lbias;
                    o = o + S[p][1];
                    o = o + S[p][2];
                    o = o + S[p][3];
#pragma unroll
                    for (int c = 0; c < 4; ++c) o = o + l_s[(p * 4 + c) * NF + tid];
                    if (layer == 0 || layer == 2) { xres[p] = xin[p]; xin[p] = leaky02(o); }
                    else if (layer == 1) xin[p] = o + xres[p];
                    else if (layer == 3) xin[p] = (o + xres[p]) + conv1[p];
                    else xin[p] = leaky02(o);
                }
            }
        };
        type_b(std::integral_constant<int, 0>{});
        type_b(std::integral_constant<int, 1>{});
        type_b(std::integral_constant<int, 2>{});
        type_b(std::integral_constant<int, 3>{});
        type_b(std::integral_constant<int, 4>{});
        // ---- head: 128 -> 128 -> 128 -> 2 (single chains: the first half; the second half keeps the barriers)
        auto dense = [&](auto d_c) {
            constexpr int d = decltype(d_c)::value;
            float *A = l_act + ((5 + d) & 1) * NP * 2 * NF;
            if constexpr (HALF == 0) {
#pragma unroll
                for (int p = 0; p < NP; ++p) A[p * 2 * NF + tid] = xin[p];
            }
            __syncthreads();
            if constexpr (HALF == 0) {
                float acc[NP], tot[NP];
                const unsigned hmask = a.head_mask[d];
                const int hb = a.head_b[d];
#pragma unroll
                for (int p = 0; p < NP; ++p) { acc[p] = l_bias[(5 + d) * NF + tid]; tot[p] = 0.0f; }
#pragma unroll
                for (int blk = 0; blk < BLOCKS_D; ++blk) {
                    request((5 * BLOCKS_R + d * BLOCKS_D + blk + RING - 1) % RING);
                    if (blk < GROUPS_D / BLK) {           // the fifth block is padding that keeps the ring aligned
#pragma unroll
                        for (int g = 0; g < BLK; ++g) {
                            const int k = (blk * BLK + g) * 4;
                            if (k % 16 == 0 && ((hmask >> (k / 16)) & 1u)) {      // "reduce-B": a block of the reduction ends
#pragma unroll
                                for (int p = 0; p < NP; ++p) {
                                    tot[p] = k == hb ? acc[p] : tot[p] + acc[p];
                                    acc[p] = 0.0f;
                                }
                            }
                            const f32x4 wv = ring[(5 * BLOCKS_R + d * BLOCKS_D + blk) % RING][g];
#pragma unroll
                            for (int p = 0; p < NP; ++p) {
                                const f32x4 av = *(const f32x4 *)(A + p * 2 * NF + k);
                                acc[p] = __builtin_fmaf(av[0], wv[0], acc[p]);
                                acc[p] = __builtin_fmaf(av[1], wv[1], acc[p]);
                                acc[p] = __builtin_fmaf(av[2], wv[2], acc[p]);
                                acc[p] = __builtin_fmaf(av[3], wv[3], acc[p]);
                            }
                        }
#pragma unroll
                        for (int p = 0; p < NP; ++p) asm volatile("" : "+v"(acc[p]));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int p = 0; p < NP; ++p) xin[p] = leaky02(hmask ? tot[p] + acc[p] : acc[p]);
            }
        };
        dense(std::integral_constant<int, 0>{});
        dense(std::integral_constant<int, 1>{});
        {
            float *A = l_act + (7 & 1) * NP * 2 * NF;
            if constexpr (HALF == 0) {
#pragma unroll
                for (int p = 0; p < NP; ++p) A[p * 2 * NF + tid] = xin[p];
            }
            __syncthreads();
            if (HALF == 0 && tid < 2 * NP) {              // (plane, output) pairs: scale and mean of every plane
                const int p = tid >> 1, o = tid & 1;
                float acc = l_p2[2 * NF + o], tot = 0.0f;
                const float *wv = l_p2 + o * NF, *av = A + p * 2 * NF;
                const unsigned hmask = a.head_mask[2];
#pragma unroll 1
                for (int cb = 0; cb < NF / 16; ++cb) {
                    if ((hmask >> cb) & 1u) {
                        tot = cb * 16 == a.head_b[2] ? acc : tot + acc;
                        acc = 0.0f;
                    }
#pragma unroll
                    for (int k = cb * 16; k < cb * 16 + 16; ++k) acc = __builtin_fmaf(av[k], wv[k], acc);
                }
                l_prm[p * 2 + o] = hmask ? tot + acc : acc;
            }
            __syncthreads();
        }
        // ---- entropy decode (wave 0 = the first 64 threads of the first half, uniform): plane after plane
        if (HALF == 0 && tid < 64) {
#pragma unroll 1
            for (int pl = 0; pl < NP; ++pl) {
                float s = l_prm[pl * 2 + 0];
                const float mean = l_prm[pl * 2 + 1];
                s = s < 1e-5f ? 1e-5f : s;
                float iv = (pm::logf_(s) - a.lmin) / a.lstep;
                iv = iv >= 0.0f ? iv : 0.0f;              // also maps NaN (corrupt stream) to row 0
                iv = iv > 255.0f ? 255.0f : iv;
                const int row = (int)iv;
                const int32_t *cd = l_cdf + row * a.cols;
                const int size = l_sizes[row];
                const int max_value = size - 2;
                const unsigned cum = (unsigned)(x & 0xFFFFull);
                int cnt = 0;
                for (int base = 0; base < size; base += 64) {
                    const int i = base + tid;
                    const bool le = i < size && (unsigned)cd[i] <= cum;
                    cnt += __builtin_popcountll(__ballot(le));
                }
                const int sidx = cnt - 1;
                const unsigned start = (unsigned)cd[sidx], freq = (unsigned)(cd[sidx + 1] - cd[sidx]);
                x = (unsigned long long)freq * (x >> 16) + (x & 0xFFFFull) - start;
                if (x < (1ull << 31)) {
                    if (pos < a.n_words) x = (x << 32) | nw; else err = 1;
                    ++pos;
                    nw = pos < a.n_words ? a.stream[pos] : 0u;
                }
                int value = sidx;
                if (value == max_value) {                   // bypass digits (rans.cpp:303-325)
                    auto bits4 = [&]() -> int {
                        const int val = (int)(x & 15ull);
                        x >>= 4;
                        if (x < (1ull << 31)) {
                            if (pos < a.n_words) x = (x << 32) | nw; else err = 1;
                            ++pos;
                            nw = pos < a.n_words ? a.stream[pos] : 0u;
                        }
                        return val;
                    };
                    int val = bits4();
                    int n_bypass = val;
                    while (val == 15) { val = bits4(); n_bypass += val; }
                    int raw = 0;
                    for (int j = 0; j < n_bypass; ++j) raw |= bits4() << (j * 4);
                    value = raw >> 1;
                    if (raw & 1) value = -value - 1; else value += max_value;
                }
                const float q = (float)(short)(value + l_offs[row]);
                if (tid == 0) {
                    const float v = __builtin_rintf(q + mean);
                    row_cur[pl * 2 * (W + 2) + wq + 1] = v;
                    a.ll_out[(long)pl * H * W + (long)h * W + wq] = v;
                }
            }
        }
        __syncthreads();
    }
    if (HALF == 0 && tid == 0) {
        a.state_out[0] = x;
        a.state_out[1] = (unsigned long long)pos;
        a.state_out[2] = (unsigned long long)err;
    }
}

// The two halves are two instantiations (their weight streams have different lengths, and ring slots must be compile-time
// constants): the branch below is uniform per wave (waves 0-1 / 2-3), and both bodies execute the SAME sequence of
// s_barrier — one in the prologue, fifteen per position (two per type-B layer, one per dense layer, two around the 128 -> 2
// head, one after the entropy decode) — every one of them outside any `if constexpr (HALF ...)`.  Keep it that way: a
// barrier added to one half only deadlocks the workgroup.
template <int NP>
__global__ __launch_bounds__(2 * NF) void ll_ar_row2_kernel(LLArgs a, int h) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (threadIdx.x < NF) ll_ar_row2_body<NP, 0>(a, h, smem);
    else ll_ar_row2_body<NP, 1>(a, h, smem);
}

// four-step decompress: CDF rows of step k (0 off the mask), then x_hat at the mask positions
__device__ __forceinline__ int scale_index(float s, float lmin, float step) {
    s = s < 1e-5f ? 1e-5f : s;
    float v = (pm::logf_(s) - lmin) / step;
    v = v >= 0.0f ? v : 0.0f;      // also maps NaN (corrupt stream) to row 0
    v = v > 255.0f ? 255.0f : v;
    return (int)v;
}

__global__ void fourstep_indexes_kernel(const float *__restrict__ params, short *idx, int N, int H, int W, int k,
                                        int psub, float lmin, float step) {
    const long total = (long)N * H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int xw = (int)(i % W);
        const int yy = (int)((i / W) % H);
        short r = 0;
        if ((yy & 1) * 2 + (xw & 1) == k) {
            const long pi = psub ? ((i / ((long)W * H)) * (H >> 1) + (yy >> 1)) * (W >> 1) + (xw >> 1) : i;
            r = (short)scale_index(params[pi * 2], lmin, step);
        }
        idx[i] = r;
    }
}

__global__ void fourstep_dequant_kernel(const short *__restrict__ sym, const float *__restrict__ params, float *so_far,
                                        int N, int H, int W, int k, int psub) {
    const long total = (long)N * H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int xw = (int)(i % W);
        const int yy = (int)((i / W) % H);
        if ((yy & 1) * 2 + (xw & 1) == k) {
            const long pi = psub ? ((i / ((long)W * H)) * (H >> 1) + (yy >> 1)) * (W >> 1) + (xw >> 1) : i;
            so_far[i] = (float)sym[i] + params[pi * 2 + 1];
        } else if (k == 0) {
            so_far[i] = 0.0f;
        }
    }
}

__constant__ int MVD_PERM[4][4] = {{0, 1, 2, 3}, {3, 2, 1, 0}, {2, 3, 0, 1}, {1, 0, 3, 2}};

__global__ void mv_fourpart_indexes_kernel(const float *__restrict__ common, const float *__restrict__ sp, short *idx,
                                           int H, int W, int t, float lmin, float step) {
    const long HW = (long)H * W;
    const long total = HW * 16;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cc = (int)(i / HW);
        const long p = i - (long)cc * HW;
        const int xw = (int)(p % W), yy = (int)(p / W);
        const int cls = (yy & 1) * 2 + (xw & 1);
        int g = 0;
        for (int q = 0; q < 4; ++q) if (MVD_PERM[t][q] == cls) g = q;
        const float scale = t == 0 ? common[p * 192 + 64 + g * 16 + cc] : sp[p * 128 + g * 16 + cc];
        idx[i] = (short)scale_index(scale, lmin, step);
    }
}

__global__ void mv_fourpart_dequant_kernel(const short *__restrict__ sym, const float *__restrict__ common,
                                           const float *__restrict__ sp, float *so_far, int H, int W, int t) {
    const long HW = (long)H * W;
    const long total = HW * 64;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i & 63);
        const long p = i >> 6;
        const int xw = (int)(p % W), yy = (int)(p / W);
        const int cls = (yy & 1) * 2 + (xw & 1);
        const int g = c >> 4, cc = c & 15;
        if (MVD_PERM[t][g] == cls) {
            const float mean = t == 0 ? common[p * 192 + 128 + c] : sp[p * 128 + 64 + g * 16 + cc];
            so_far[i] = (float)sym[(long)cc * HW + p] + mean;
        } else if (t == 0) {
            so_far[i] = 0.0f;
        }
    }
}

// int16 NCHW symbols -> float NHWC (MV hyper latent z_hat after host decode)
__global__ void sym_to_nhwc_kernel(const short *__restrict__ sym, float *out, int HW, int C) {
    const long total = (long)HW * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long p = i / C;
        out[i] = (float)sym[(long)c * HW + p];
    }
}

}  // namespace

extern "C" int64_t pmctf_ll_ar_packed_size(void) { return W_TOTAL; }

// HOST: pack the (already masked) filters of ContextFusionSubband into the layout of the decode kernel.
//  w_a [128][1][3][3], b_a[128]; w_b[5] = {res0.conv1, res0.conv2, res1.conv1, res1.conv2, maskedConv2}: [128][128][3][3] + bias;
//  w_p0, w_p1: [128][128] (1x1), w_p2: [2][128]
extern "C" int pmctf_ll_ar_pack_weights(const float *w_a, const float *b_a, const float *const *w_b, const float *const *b_b,
                                        const float *w_p0, const float *b_p0, const float *w_p1, const float *b_p1,
                                        const float *w_p2, const float *b_p2, float *out) {
    if (!w_a || !b_a || !w_b || !b_b || !w_p0 || !b_p0 || !w_p1 || !b_p1 || !w_p2 || !b_p2 || !out) return PMCTF_EINVAL;
    static const int TA[4][2] = {{0, 0}, {0, 1}, {0, 2}, {1, 0}};
    static const int TBT[5][2] = {{0, 0}, {0, 1}, {0, 2}, {1, 0}, {1, 1}};
    for (int t = 0; t < 4; ++t)
        for (int co = 0; co < NF; ++co) out[W_L0 + t * NF + co] = w_a[co * 9 + TA[t][0] * 3 + TA[t][1]];
    for (int co = 0; co < NF; ++co) out[B_L0 + co] = b_a[co];
    for (int l = 0; l < 5; ++l) {
        float *o = out + W_MB + l * SZ_MB;
        for (int cb = 0; cb < 8; ++cb)
            for (int t = 0; t < TB; ++t)
                for (int ci = 0; ci < 16; ++ci)
                    for (int co = 0; co < NF; ++co)
                        o[(long)((cb * TB + t) * 16 + ci) * NF + co] =
                            w_b[l][((long)co * NF + cb * 16 + ci) * 9 + TBT[t][0] * 3 + TBT[t][1]];
        for (int co = 0; co < NF; ++co) o[(long)8 * TB * 16 * NF + co] = b_b[l][co];
    }
    const float *wp[2] = {w_p0, w_p1};
    const float *bp[2] = {b_p0, b_p1};
    const long offs[2] = {W_P0, W_P1};
    for (int l = 0; l < 2; ++l) {
        for (int k = 0; k < NF; ++k)
            for (int co = 0; co < NF; ++co) out[offs[l] + (long)k * NF + co] = wp[l][(long)co * NF + k];
        for (int co = 0; co < NF; ++co) out[offs[l] + (long)NF * NF + co] = bp[l][co];
    }
    for (int k = 0; k < NF; ++k)
        for (int co = 0; co < 2; ++co) out[W_P2 + (long)k * 2 + co] = w_p2[(long)co * NF + k];
    out[W_P2 + NF * 2 + 0] = b_p2[0];
    out[W_P2 + NF * 2 + 1] = b_p2[1];
    // the stream layout of ll_ar_stream_kernel: [group][co][4] over the chain index k of each layer
    long g0 = 0;
    for (int l = 0; l < 5; ++l) {
        for (int k = 0; k < 8 * TB * 16; ++k) {
            const int pair = k / 16, cb = pair / TB, t = pair % TB, ci = k % 16;
            for (int co = 0; co < NF; ++co)
                out[S_W + ((g0 + k / 4) * NF + co) * 4 + (k & 3)] =
                    w_b[l][((long)co * NF + cb * 16 + ci) * 9 + TBT[t][0] * 3 + TBT[t][1]];
        }
        for (int co = 0; co < NF; ++co) out[S_BIAS + l * NF + co] = b_b[l][co];
        g0 += GROUPS_B;
    }
    for (int l = 0; l < 2; ++l) {
        for (int k = 0; k < NF; ++k)
            for (int co = 0; co < NF; ++co) out[S_W + ((g0 + k / 4) * NF + co) * 4 + (k & 3)] = wp[l][(long)co * NF + k];
        for (int co = 0; co < NF; ++co) out[S_BIAS + (5 + l) * NF + co] = bp[l][co];
        for (long i = (g0 + GROUPS_D) * NF * 4; i < (g0 + GROUPS_DP) * NF * 4; ++i) out[S_W + i] = 0.0f;
        g0 += GROUPS_DP;
    }
    for (int o = 0; o < 2; ++o)
        for (int k = 0; k < NF; ++k) out[S_P2 + o * NF + k] = w_p2[(long)o * NF + k];
    out[S_P2 + 2 * NF + 0] = b_p2[0];
    out[S_P2 + 2 * NF + 1] = b_p2[1];
    // the row-wise layouts
    long r0 = 0;
    for (int l = 0; l < 5; ++l) {
        for (int cb = 0; cb < 8; ++cb)
            for (int t = 0; t < TB; ++t)
                for (int ci = 0; ci < 16; ++ci)
                    for (int co = 0; co < NF; ++co) {
                        const float v = w_b[l][((long)co * NF + cb * 16 + ci) * 9 + TBT[t][0] * 3 + TBT[t][1]];
                        if (t < 3) {
                            out[U_W + ((((long)l * 8 + cb) * 3 + t) * 16 + ci) * NF + co] = v;
                        } else {
                            const int k2 = cb * 32 + (t - 3) * 16 + ci;
                            out[R_W + ((r0 + k2 / 4) * NF + co) * 4 + (k2 & 3)] = v;
                        }
                    }
        r0 += GROUPS_R;
    }
    for (int l = 0; l < 2; ++l) {
        for (int k = 0; k < NF; ++k)
            for (int co = 0; co < NF; ++co) out[R_W + ((r0 + k / 4) * NF + co) * 4 + (k & 3)] = wp[l][(long)co * NF + k];
        for (long i = (r0 + GROUPS_D) * NF * 4; i < (r0 + GROUPS_DP) * NF * 4; ++i) out[R_W + i] = 0.0f;
        r0 += GROUPS_DP;
    }
    return PMCTF_OK;
}

// [5 layers][N][H+1][W+2][NF] layer inputs (the kernels address it as [5][N][H][W][NF]) + [5][N][W][8][NF] chunk prefixes of a row
inline int64_t ll_scratch_acts(int N, int H, int W) { return (int64_t)ll_scratch_acts_dev(N, H, W); }
extern "C" int64_t pmctf_ll_ar_scratch_floats(int N, int H, int W) {
    return ll_scratch_acts(N, H, W) + (int64_t)5 * N * W * 8 * NF;
}

extern "C" int pmctf_ll_ar_decode_f32(const float *w_packed, const uint32_t *stream_words, int64_t n_words, uint64_t x0,
                                      int64_t pos0, const int32_t *cdf, const int32_t *sizes, const int32_t *offsets,
                                      int cdf_cols, float log_scale_min, float log_scale_step, float *ll_out,
                                      float *scratch_zeroed, int N, int H, int W, uint64_t *state_out, void *stream) {
    return pmctf_ll_ar_decode_rules_f32(w_packed, stream_words, n_words, x0, pos0, cdf, sizes, offsets, cdf_cols, log_scale_min,
                                        log_scale_step, ll_out, scratch_zeroed, N, H, W, state_out, PMCTF_SUM_CHAIN,
                                        PMCTF_SUM_CHAIN, PMCTF_SUM_CHAIN, stream);
}

extern "C" int pmctf_ll_ar_decode_rules_f32(const float *w_packed, const uint32_t *stream_words, int64_t n_words, uint64_t x0,
                                            int64_t pos0, const int32_t *cdf, const int32_t *sizes, const int32_t *offsets,
                                            int cdf_cols, float log_scale_min, float log_scale_step, float *ll_out,
                                            float *scratch_zeroed, int N, int H, int W, uint64_t *state_out,
                                            int sum_rule_3x3, int sum_rule_head, int sum_rule_head_out, void *stream) {
    if ((sum_rule_3x3 != PMCTF_SUM_CHAIN && sum_rule_3x3 != PMCTF_SUM_BLOCKS) ||
        (sum_rule_head != PMCTF_SUM_CHAIN && (sum_rule_head < 16 || sum_rule_head % 16)) ||
        (sum_rule_head_out != PMCTF_SUM_CHAIN && (sum_rule_head_out < 16 || sum_rule_head_out % 16)))
        return PMCTF_EINVAL;
    if (!w_packed || !stream_words || !cdf || !sizes || !offsets || !ll_out || !scratch_zeroed || !state_out || N < 1 ||
        N > LL_MAX_PLANES || H < 1 || W < 1 || n_words < 0 || cdf_cols < 3)
        return PMCTF_EINVAL;
    LLArgs a;
    a.w = w_packed; a.stream = stream_words; a.n_words = n_words; a.x0 = x0; a.pos0 = pos0;
    a.cdf = cdf; a.sizes = sizes; a.offsets = offsets; a.cols = cdf_cols;
    a.lmin = log_scale_min; a.lstep = log_scale_step;
    a.ll_out = ll_out; a.bufs = scratch_zeroed; a.N = N; a.H = H; a.W = W;
    a.state_out = (unsigned long long *)state_out;
    a.blocks = sum_rule_3x3 == PMCTF_SUM_BLOCKS;
    a.head_b[0] = a.head_b[1] = sum_rule_head;
    a.head_b[2] = sum_rule_head_out;
    for (int i = 0; i < 3; ++i) a.head_mask[i] = reduce_mask(a.head_b[i]);
    const size_t smem = ((size_t)256 * cdf_cols + 512) * sizeof(int32_t) +
                        ((size_t)2 * N * TB * NF + 5 * N * NF + 5 * N * 4 * NF + 7 * NF + 2 * NF + 2 + 2 * N +
                         (size_t)N * 2 * (W + 2)) * sizeof(float) + 64;
    static const bool v1 = getenv("PMCTF_LL_AR_V1") != nullptr;      // the first kernel, kept for A/B measurements
    static const bool v2 = getenv("PMCTF_LL_AR_V2") != nullptr;      // the streaming kernel where the row-wise form applies
    if (!v1 && !v2 && a.blocks && N <= 2 && smem <= 150 * 1024) {
        // rule "blocks": the row-wise form — per row the chunk prefixes of the whole row (all CUs), then the sequential kernel
        static std::once_flag once_r[4];
        static const bool one_thread = getenv("PMCTF_LL_AR_ROW1") != nullptr;    // one thread per channel (A/B measurements)
        hipStream_t st = (hipStream_t)stream;
        auto row = [&](auto pre_k, auto row_k, std::once_flag &flag, int threads, size_t lds) {
            std::call_once(flag, [row_k] {
                (void)hipFuncSetAttribute((const void *)row_k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            });
            for (int h = 0; h < H; ++h) {
                if (h > 0) PM_LAUNCH(pre_k, dim3((W + 7) / 8, 5 * N), dim3(NF), 0, st, a, h);
                PM_LAUNCH(row_k, dim3(1), dim3(threads), lds, st, a, h);
            }
        };
        const size_t smem2 = smem + (size_t)N * 4 * NF * sizeof(float);
        if (one_thread || smem2 > 150 * 1024) {
            if (N == 1) row(ll_ar_pre_kernel<1>, ll_ar_row_kernel<1>, once_r[0], NF, smem);
            else row(ll_ar_pre_kernel<2>, ll_ar_row_kernel<2>, once_r[1], NF, smem);
        } else {
            if (N == 1) row(ll_ar_pre_kernel<1>, ll_ar_row2_kernel<1>, once_r[2], 2 * NF, smem2);
            else row(ll_ar_pre_kernel<2>, ll_ar_row2_kernel<2>, once_r[3], 2 * NF, smem2);
        }
        return launch_ok();
    }
    if (!v1 && N <= 2 && smem <= 150 * 1024) {       // Y and UV streams; three or four planes (RGB stills) keep the first kernel
        static std::once_flag once[LL_MAX_PLANES];
        auto go = [&](auto kernel, std::once_flag &flag) {
            std::call_once(flag, [kernel] {
                (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            });
            PM_LAUNCH(kernel, dim3(1), dim3(NF), smem, (hipStream_t)stream, a);
        };
        if (N == 1) go(ll_ar_stream_kernel<1>, once[0]);
        else go(ll_ar_stream_kernel<2>, once[1]);
        return launch_ok();
    }
    PM_LAUNCH(ll_ar_decode_kernel, dim3(1), dim3(128 * N), 0, (hipStream_t)stream, a);
    return launch_ok();
}

extern "C" int pmctf_fourstep_indexes_f32(const float *params, int16_t *idx, int N, int H, int W, int k, int params_sub,
                                          float log_scale_min, float log_scale_step, void *stream) {
    if (!params || !idx || N <= 0 || H <= 0 || W <= 0 || k < 0 || k > 3 || (params_sub && ((H | W) & 1))) return PMCTF_EINVAL;
    PM_LAUNCH(fourstep_indexes_kernel, dim3(grid_for((long)N * H * W)), dim3(256), 0, (hipStream_t)stream, params, idx, N,
              H, W, k, params_sub, log_scale_min, log_scale_step);
    return launch_ok();
}

extern "C" int pmctf_fourstep_dequant_f32(const int16_t *sym, const float *params, float *so_far, int N, int H, int W,
                                          int k, int params_sub, void *stream) {
    if (!sym || !params || !so_far || N <= 0 || H <= 0 || W <= 0 || k < 0 || k > 3 || (params_sub && ((H | W) & 1)))
        return PMCTF_EINVAL;
    PM_LAUNCH(fourstep_dequant_kernel, dim3(grid_for((long)N * H * W)), dim3(256), 0, (hipStream_t)stream, sym, params,
              so_far, N, H, W, k, params_sub);
    return launch_ok();
}

extern "C" int pmctf_mv_fourpart_indexes_f32(const float *common, const float *sp, int16_t *idx, int H, int W, int t,
                                             float log_scale_min, float log_scale_step, void *stream) {
    if (!common || !idx || H <= 0 || W <= 0 || t < 0 || t > 3 || (t > 0 && !sp)) return PMCTF_EINVAL;
    PM_LAUNCH(mv_fourpart_indexes_kernel, dim3(grid_for((long)H * W * 16)), dim3(256), 0, (hipStream_t)stream, common, sp,
              idx, H, W, t, log_scale_min, log_scale_step);
    return launch_ok();
}

extern "C" int pmctf_mv_fourpart_dequant_f32(const int16_t *sym, const float *common, const float *sp, float *so_far,
                                             int H, int W, int t, void *stream) {
    if (!sym || !common || !so_far || H <= 0 || W <= 0 || t < 0 || t > 3 || (t > 0 && !sp)) return PMCTF_EINVAL;
    PM_LAUNCH(mv_fourpart_dequant_kernel, dim3(grid_for((long)H * W * 64)), dim3(256), 0, (hipStream_t)stream, sym, common,
              sp, so_far, H, W, t);
    return launch_ok();
}

extern "C" int pmctf_sym_to_nhwc_f32(const int16_t *sym, float *out, int HW, int C, void *stream) {
    if (!sym || !out || HW <= 0 || C <= 0) return PMCTF_EINVAL;
    PM_LAUNCH(sym_to_nhwc_kernel, dim3(grid_for((long)HW * C)), dim3(256), 0, (hipStream_t)stream, sym, out, HW, C);
    return launch_ok();
}
