// conv_mfma.hip — implicit-GEMM convolution on the gfx950 matrix cores, exact f32.
//
// Replaces F.conv2d behind the reference's nn.Conv2d layers (pmctf_hip.h lists the
// call sites).  D[cout x pixel] += A[cout x k] * B[k x pixel] with
// v_mfma_f32_16x16x4_f32, whose result is bit-for-bit a k-ordered fmaf chain, so the
// sum order of the PM-F32 spec is reproduced exactly:
//   rule 0 ("chain"):  acc = bias; for 16-channel chunk cb: for ky: for kx: for ci in chunk: fmaf
//   rule 1 ("blocks"): per chunk a chain from zero S_cb; out = (..((S_0 + bias) + S_1) + ..) + S_last  — what ATen's CPU
//                      path (oneDNN direct convolution) computes for KH*KW > 1; kernels instantiated with BSUM = true
//                      (ConvArgs::bsum; every chunk's partial sum is added to a second accumulator set)
// Workgroup = 4 waves (256 threads) = one TH x TW output tile of one image, all MT
// 16-row cout tiles of one M-block.  Per chunk the input patch (tile + halo, 16
// channels) is staged NHWC -> LDS once and shared by the 4 waves; each wave owns
// NT 16-pixel row segments and keeps MT*NT accumulators (4 VGPRs each).  Weight
// fragments are pre-packed on the host in lane order and read straight from L2
// (every workgroup reads the same <=0.5 MB).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include "pm_device_math.h"
#include "conv_epilogue.h"
#include "launch.h"
#include "../../include/pmctf_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CB = 16;       // channels per chunk
constexpr int CP = 18;       // LDS pixel stride in words (16 + 2: conflict-free b32 reads, 8-B aligned)
constexpr int WAVES = 4;

struct ConvArgs {
    const float *x, *wp, *bp, *res1, *res2;
    float *y;
    int N, H, W, Cin, Cout, KH, KW, S, pad_h, pad_w, Ho, Wo;
    int tiles_x, tiles_y, ncb, act;
    float slope;
    // mtp: cout tiles per packed M-block (the weight layout); a workgroup handles MT <= mtp of them, blockIdx.z
    // enumerates groups of MT tiles.  [oy_base, oy_end): output rows this launch covers.
    int mtp, oy_base, oy_end;
};

// Summation rule of a launch (see the header), a SEPARATE kernel argument of the kernels that implement rule 1: the
// register-tight rule-0 kernels (conv3x3s1_wave_kernel<7,2> sits at the VGPR and SGPR limits) must not grow by a word.
// bsum 0 = one chain from the bias; bsum 1 = the reduction is cut into blocks of `bchunks` 16-channel chunks whose sums
// are added in turn — bias_first 0: every block's chain starts at zero and the bias is added to the first block's sum
// (ATen's KH*KW > 1 layers, bchunks = 1); bias_first 1: the first block's chain starts at the bias, later blocks at zero
// (ATen's 1x1 layers with a blocked reduction, bchunks = block / 16)
struct SumCfg { int bsum, bchunks, bias_first; };
thread_local SumCfg t_sum = {0, 1, 0};      // of the convolution call in progress on this host thread

// MT: cout tiles per workgroup, NT: pixel tiles per wave, TW16: 16-pixel segments per tile row.
template <int MT, int NT, int TW16, bool BSUM>
__device__ __forceinline__ void conv_mfma_body(const ConvArgs &a, const SumCfg sc);

template <int MT, int NT, int TW16>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs a) {
    conv_mfma_body<MT, NT, TW16, false>(a, SumCfg{0, 1, 0});
}
template <int MT, int NT, int TW16>
__global__ __launch_bounds__(256) void conv_mfma_bsum_kernel(ConvArgs a, SumCfg sc) {
    conv_mfma_body<MT, NT, TW16, true>(a, sc);
}

template <int MT, int NT, int TW16, bool BSUM>
__device__ __forceinline__ void conv_mfma_body(const ConvArgs &a, const SumCfg sc) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int TW = TW16 * 16;
    constexpr int TH = NT * WAVES / TW16;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int tile = blockIdx.x;
    const int n = blockIdx.y;
    const int mtile0 = blockIdx.z * MT;                 // first cout tile of this workgroup
    const int mb = mtile0 / a.mtp, mtin = mtile0 - mb * a.mtp;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int oy0 = a.oy_base + ty * TH, ox0 = tx * TW;
    const int LH = (TH - 1) * a.S + a.KH, LW = (TW - 1) * a.S + a.KW;
    const int iy0 = oy0 * a.S - a.pad_h, ix0 = ox0 * a.S - a.pad_w;
    const int taps = a.KH * a.KW;

    // rule 0: accumulators start at the bias: acc = bias[co]; rule 1 (BSUM): every chunk starts from zero
    f32x4 acc[MT][NT];
    f32x4 tot[BSUM ? MT : 1][BSUM ? NT : 1];
    {
        const float *bp = a.bp + (size_t)mtile0 * 16 + 4 * (lane >> 4);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x4 b = (BSUM && !sc.bias_first) ? f32x4{0.f, 0.f, 0.f, 0.f} : *(const f32x4 *)(bp + mt * 16);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = b;
        }
    }
    // per-lane LDS word offset of (pixel lane&15 of segment nt, channel lane>>4)
    int boff[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int seg = wave * NT + nt;
        const int r = seg / TW16, c16 = seg - r * TW16;
        boff[nt] = ((r * a.S) * LW + (c16 * 16 + (lane & 15)) * a.S) * CP + (lane >> 4);
    }
    const int E = LH * LW * 4;  // float4 units per chunk
    const int wstride = a.mtp * 256;
    const float *wbase = a.wp + (size_t)mb * a.ncb * taps * wstride + mtin * 256 + lane * 4;

    for (int cb = 0; cb < a.ncb; ++cb) {
        __syncthreads();  // previous chunk fully consumed
        for (int e = tid; e < E; e += 256) {
            const int pix = e >> 2, part = e & 3;
            const int ly = pix / LW, lx = pix - ly * LW;
            const int gy = iy0 + ly, gx = ix0 + lx;
            const int c = cb * CB + part * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && c < a.Cin)
                v = *(const f32x4 *)(a.x + (((size_t)n * a.H + gy) * a.W + gx) * a.Cin + c);
            float2 *dst = (float2 *)(lds + pix * CP + part * 4);
            dst[0] = make_float2(v.x, v.y);
            dst[1] = make_float2(v.z, v.w);
        }
        __syncthreads();
        const int crem = a.Cin - cb * CB;
        const int ksteps = crem >= CB ? 4 : (crem + 3) >> 2;
        const float *wc = wbase + (size_t)cb * taps * wstride;
        int tap = 0;
        for (int ky = 0; ky < a.KH; ++ky) {
            for (int kx = 0; kx < a.KW; ++kx, ++tap) {
                f32x4 af[MT];  // af[mt][ks]: A fragment of k-step ks
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) af[mt] = *(const f32x4 *)(wc + (size_t)tap * wstride + mt * 256);
                const int toff = (ky * LW + kx) * CP;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    if (ks < ksteps) {
                        float bf[NT];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) bf[nt] = lds[boff[nt] + toff + ks * 4];
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt][ks], bf[nt], acc[mt][nt], 0, 0, 0);
                    }
                }
            }
        }
        if constexpr (BSUM) {
            if ((cb + 1) % sc.bchunks == 0 || cb + 1 == a.ncb) {     // a block's sum is complete (wave-uniform)
                const bool first = cb < sc.bchunks;
                const float *bp = a.bp + (size_t)mtile0 * 16 + 4 * (lane >> 4);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const f32x4 b = *(const f32x4 *)(bp + mt * 16);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        tot[mt][nt] = first ? (sc.bias_first ? acc[mt][nt] : acc[mt][nt] + b) : tot[mt][nt] + acc[mt][nt];
                        acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
        }
    }
    if constexpr (BSUM) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = tot[mt][nt];
    }

    // epilogue: lane holds couts co..co+3 of pixel (lane&15) of each segment
    PM_EPILOGUE(a,
_Pragma("unroll")
    for (int nt = 0; nt < NT; ++nt) {
        const int seg = wave * NT + nt;
        const int r = seg / TW16, c16 = seg - r * TW16;
        const int oy = oy0 + r, ox = ox0 + c16 * 16 + (lane & 15);
        if (oy >= a.oy_end || ox >= a.Wo) continue;
        const size_t pbase = (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.Cout;
_Pragma("unroll")
        for (int mt = 0; mt < MT; ++mt) {
            const int co = (mtile0 + mt) * 16 + 4 * (lane >> 4);
            if (co >= a.Cout) continue;
            store_frag<ACT, RES>(a, acc[mt][nt], pbase, co);
        }
    })
}


// ---------------------------------------------------------------------------------------------------
// Pipelined variant (same arithmetic, same sum order): the input patch of chunk cb+1 is fetched into
// registers while chunk cb is being multiplied and written to the second LDS buffer afterwards (one
// barrier per chunk); the weight fragments of the next tap are loaded one tap ahead; <= 256 VGPRs so
// that two workgroups share a CU (2 waves/SIMD) and fill each other's stalls.
template <int MT, int NT, int TW16, int MAXP>
__global__ __launch_bounds__(256, 2) void conv_mfma_pipe_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int TW = TW16 * 16;
    constexpr int TH = NT * WAVES / TW16;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int tile = blockIdx.x;
    const int n = blockIdx.y;
    const int mtile0 = blockIdx.z * MT;                 // first cout tile of this workgroup
    const int mb = mtile0 / a.mtp, mtin = mtile0 - mb * a.mtp;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int oy0 = a.oy_base + ty * TH, ox0 = tx * TW;
    const int LH = (TH - 1) * a.S + a.KH, LW = (TW - 1) * a.S + a.KW;
    const int iy0 = oy0 * a.S - a.pad_h, ix0 = ox0 * a.S - a.pad_w;
    const int taps = a.KH * a.KW;
    const int bufsz = LH * LW * CP;

    f32x4 acc[MT][NT];
    {
        const float *bp = a.bp + (size_t)mtile0 * 16 + 4 * (lane >> 4);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x4 b = *(const f32x4 *)(bp + mt * 16);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = b;
        }
    }
    int boff[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int seg = wave * NT + nt;
        const int r = seg / TW16, c16 = seg - r * TW16;
        boff[nt] = ((r * a.S) * LW + (c16 * 16 + (lane & 15)) * a.S) * CP + (lane >> 4);
    }
    const int E = LH * LW * 4;
    // staging: thread `tid` owns float4 slots e = tid + 256*j of the patch (pixel e>>2, channel quad e&3)
    f32x4 pre[MAXP];
    auto fetch = [&](int cb) {
#pragma unroll
        for (int j = 0; j < MAXP; ++j) {
            const int e = tid + 256 * j;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (e < E) {
                const int pix = e >> 2, part = e & 3;
                const int ly = pix / LW, lx = pix - ly * LW;
                const int gy = iy0 + ly, gx = ix0 + lx;
                if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                    v = *(const f32x4 *)(a.x + (((size_t)n * a.H + gy) * a.W + gx) * a.Cin + cb * CB + part * 4);
            }
            pre[j] = v;
        }
    };
    auto stash = [&](float *buf) {
#pragma unroll
        for (int j = 0; j < MAXP; ++j) {
            const int e = tid + 256 * j;
            if (e < E) {
                float2 *dst = (float2 *)(buf + (e >> 2) * CP + (e & 3) * 4);
                dst[0] = make_float2(pre[j].x, pre[j].y);
                dst[1] = make_float2(pre[j].z, pre[j].w);
            }
        }
    };
    const int wstride = a.mtp * 256;
    const float *wq = a.wp + (size_t)mb * a.ncb * taps * wstride + mtin * 256 + lane * 4;   // walks [cb][tap]
    const long wsteps = (long)a.ncb * taps;
    f32x4 a_cur[MT], a_nxt[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a_cur[mt] = *(const f32x4 *)(wq + mt * 256);
    fetch(0);
    stash(lds);
    __syncthreads();
    long wstep = 0;
    // B operands: two register sets; set (ks&1) feeds k-step ks while the other set receives the next k-step
    // (or the next tap's first k-step), so every LDS read is issued one full k-step (MT*NT MFMAs) ahead of its use.
    float b0[NT], b1[NT];
    for (int cb = 0; cb < a.ncb; ++cb) {
        const float *cur = lds + (cb & 1) * bufsz;
        const bool more = cb + 1 < a.ncb;
        if (more) fetch(cb + 1);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b0[nt] = cur[boff[nt]];
        int ky = 0, kx = 0;
        for (int tap = 0; tap < taps; ++tap) {
            ++wstep;
            {   // next tap's weight fragments, always loaded (the last step re-reads its own block: no branch)
                const long wn = wstep < wsteps ? wstep : wsteps - 1;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a_nxt[mt] = *(const f32x4 *)(wq + wn * wstride + mt * 256);
            }
            const float *bbase = cur + (ky * LW + kx) * CP;
            if (++kx == a.KW) { kx = 0; ++ky; }
            // first k-step of the next tap (the last tap of a chunk harmlessly re-reads its own first k-step)
            const float *bnext = tap + 1 < taps ? cur + (ky * LW + kx) * CP : bbase;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b1[nt] = bbase[boff[nt] + 4];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[mt][0], b0[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b0[nt] = bbase[boff[nt] + 8];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[mt][1], b1[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b1[nt] = bbase[boff[nt] + 12];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[mt][2], b0[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b0[nt] = bnext[boff[nt]];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[mt][3], b1[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a_cur[mt] = a_nxt[mt];
        }
        if (more) stash(lds + ((cb + 1) & 1) * bufsz);
        __syncthreads();
    }

    PM_EPILOGUE_NOTRANS(a,
_Pragma("unroll")
    for (int nt = 0; nt < NT; ++nt) {
        const int seg = wave * NT + nt;
        const int r = seg / TW16, c16 = seg - r * TW16;
        const int oy = oy0 + r, ox = ox0 + c16 * 16 + (lane & 15);
        if (oy >= a.oy_end || ox >= a.Wo) continue;
        const size_t pbase = (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.Cout;
_Pragma("unroll")
        for (int mt = 0; mt < MT; ++mt) {
            const int co = (mtile0 + mt) * 16 + 4 * (lane >> 4);
            if (co >= a.Cout) continue;
            store_frag<ACT, RES>(a, acc[mt][nt], pbase, co);
        }
    })
}

// ---------------------------------------------------------------------------------------------------
// Barrier-free variant for large planes: every WAVE owns a 4x16-pixel output tile (NT = 4 rows of 16 px), stages
// its own input patch ((3*S+KH) x (15*S+KW) pixels x 16 channels) into a wave-private, double-buffered LDS region
// and never synchronises with the other waves of the workgroup (the LDS operations of one wave are ordered), so the
// two waves sharing a SIMD drift apart and keep the matrix pipe busy across chunk boundaries.  Same arithmetic and
// sum order as the other variants.  The workgroup (4 waves = 8x32 pixels) only groups waves for dispatch.
// NBUF = 2: double-buffered patch (62 KB of LDS per workgroup for 3x3: two workgroups per CU, what the 238-VGPR
// 112-cout instance can use anyway).  NBUF = 1 (narrow layers, <= 168 VGPRs): one buffer per wave — the next chunk is
// written after the last read of the current one has been issued, in-order LDS makes that safe — so that THREE workgroups
// fit a CU and a third wave per SIMD covers the shorter chunks' staging.
template <int MT, int MAXP, int NBUF = 2>
__global__ __launch_bounds__(256, NBUF == 1 ? 3 : 2) void conv_mfma_wave_kernel(ConvArgs a) {
    constexpr int NT = 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x;
    const int n = blockIdx.y;
    const int mtile0 = blockIdx.z * MT;
    const int mb = mtile0 / a.mtp, mtin = mtile0 - mb * a.mtp;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int oy0 = a.oy_base + ty * 8 + (wave >> 1) * 4, ox0 = tx * 32 + (wave & 1) * 16;
    const int LH = 3 * a.S + a.KH, LW = 15 * a.S + a.KW;
    const int iy0 = oy0 * a.S - a.pad_h, ix0 = ox0 * a.S - a.pad_w;
    const int taps = a.KH * a.KW;
    const int bufsz = LH * LW * CP;
    float *wlds = lds + wave * NBUF * bufsz;

    f32x4 acc[MT][NT];
    {
        const float *bp = a.bp + (size_t)mtile0 * 16 + 4 * (lane >> 4);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x4 b = *(const f32x4 *)(bp + mt * 16);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = b;
        }
    }
    int boff[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        boff[nt] = ((nt * a.S) * LW + (lane & 15) * a.S) * CP + (lane >> 4);
    }
    const int E = LH * LW * 4;
    // staging: lane owns float4 slots e = lane + 64*j of this wave's patch (pixel e>>2, channel quad e&3)
    f32x4 pre[MAXP];
    auto fetch = [&](int cb) {
#pragma unroll
        for (int j = 0; j < MAXP; ++j) {
            const int e = lane + 64 * j;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (e < E) {
                const int pix = e >> 2, part = e & 3;
                const int ly = pix / LW, lx = pix - ly * LW;
                const int gy = iy0 + ly, gx = ix0 + lx;
                if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                    v = *(const f32x4 *)(a.x + (((size_t)n * a.H + gy) * a.W + gx) * a.Cin + cb * CB + part * 4);
            }
            pre[j] = v;
        }
    };
    auto stash = [&](float *buf) {
#pragma unroll
        for (int j = 0; j < MAXP; ++j) {
            const int e = lane + 64 * j;
            if (e < E) {
                float2 *dst = (float2 *)(buf + (e >> 2) * CP + (e & 3) * 4);
                dst[0] = make_float2(pre[j].x, pre[j].y);
                dst[1] = make_float2(pre[j].z, pre[j].w);
            }
        }
    };
    const int wstride = a.mtp * 256;
    const float *wq = a.wp + (size_t)mb * a.ncb * taps * wstride + mtin * 256 + lane * 4;   // walks [cb][tap]
    const long wsteps = (long)a.ncb * taps;
    f32x4 a_cur[MT], a_nxt[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a_cur[mt] = *(const f32x4 *)(wq + mt * 256);
    fetch(0);
    stash(wlds);
    long wstep = 0;
    // B operands: two register sets; set (ks&1) feeds k-step ks while the other set receives the next k-step
    // (or the next tap's first k-step), so every LDS read is issued one full k-step (MT*NT MFMAs) ahead of its use.
    float b0[NT], b1[NT];
    for (int cb = 0; cb < a.ncb; ++cb) {
        const float *cur = wlds + (NBUF == 2 ? (cb & 1) * bufsz : 0);
        const bool more = cb + 1 < a.ncb;
        if (more) fetch(cb + 1);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b0[nt] = cur[boff[nt]];
        int ky = 0, kx = 0;
        for (int tap = 0; tap < taps; ++tap) {
            ++wstep;
            {   // next tap's weight fragments, always loaded (the last step re-reads its own block: no branch)
                const long wn = wstep < wsteps ? wstep : wsteps - 1;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a_nxt[mt] = *(const f32x4 *)(wq + wn * wstride + mt * 256);
            }
            const float *bbase = cur + (ky * LW + kx) * CP;
            if (++kx == a.KW) { kx = 0; ++ky; }
            // first k-step of the next tap (the last tap of a chunk harmlessly re-reads its own first k-step)
            const float *bnext = tap + 1 < taps ? cur + (ky * LW + kx) * CP : bbase;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b1[nt] = bbase[boff[nt] + 4];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[mt][0], b0[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b0[nt] = bbase[boff[nt] + 8];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[mt][1], b1[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b1[nt] = bbase[boff[nt] + 12];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[mt][2], b0[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b0[nt] = bnext[boff[nt]];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[mt][3], b1[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a_cur[mt] = a_nxt[mt];
        }
        // the other buffer was last read in chunk cb-1 by this same wave (NBUF = 1: this buffer, all of whose reads have
        // been issued by now): in-order LDS makes the overwrite safe
        if (more) stash(wlds + (NBUF == 2 ? ((cb + 1) & 1) * bufsz : 0));
    }

    PM_EPILOGUE_NOTRANS(a,
_Pragma("unroll")
    for (int nt = 0; nt < NT; ++nt) {
        const int oy = oy0 + nt, ox = ox0 + (lane & 15);
        if (oy >= a.oy_end || ox >= a.Wo) continue;
        const size_t pbase = (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.Cout;
_Pragma("unroll")
        for (int mt = 0; mt < MT; ++mt) {
            const int co = (mtile0 + mt) * 16 + 4 * (lane >> 4);
            if (co >= a.Cout) continue;
            store_frag<ACT, RES>(a, acc[mt][nt], pbase, co);
        }
    })
}


// ---------------------------------------------------------------------------------------------------
// 3x3 / stride 1 specialisation of the barrier-free wave-private variant (the shape of 70 % of the encode's FLOPs).
// On gfx950 a v_mfma_f32_16x16x4_f32 and vector-ALU instructions do NOT overlap on a SIMD (tools/mfma_valu_overlap.hip:
// time = matrix cycles + vector cycles, within a wave and across the waves of a SIMD), so every v_mov / address
// computation of the inner loop is matrix time lost.  Here the nine taps are unrolled (the two weight-fragment sets
// alternate without copies, one copy per 16-channel chunk remains), every LDS address is ONE per-wave base register plus an
// immediate, the staging slots' global offsets and bounds are computed once per tile, and the 16-channel chunk advances
// a scalar base.  Same arithmetic and sum order as every other variant (chunk, ky, kx, ci ascending).
// BSUM: summation rule "blocks" (per-chunk sums from zero, added in turn to a second accumulator set).  Two sets of
// MT x 4 quads do not fit 256 registers at MT = 7: that instantiation runs ONE workgroup per CU (512 registers per lane, the
// compiler keeps one set in AGPRs); the rule-0 instantiation measured the same 137 TFLOP/s at one workgroup per CU
// (docs/history.md section 9), i.e. the kernel is MFMA-bound, not occupancy-bound.
// MOFF >= 0: the workgroup takes MT cout tiles starting at tile MOFF of packed M-block blockIdx.z (a 7-tile block run as
// 4 + 3 tiles in two launches: both accumulator sets of rule "blocks" then fit 256 registers at two workgroups per CU).
template <int MT, int NBUF, bool BSUM, int MOFF>
__device__ __forceinline__ void conv3x3s1_wave_body(const ConvArgs &a, const int tile, float *lds) {
    constexpr int NT = 4, LH = 6, LW = 18, MAXP = 7;
    constexpr int BUFSZ = LH * LW * CP;
    constexpr int E = LH * LW * 4;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.y;
    const int mtile0 = MOFF >= 0 ? blockIdx.z * a.mtp + MOFF : blockIdx.z * MT;
    const int mb = mtile0 / a.mtp, mtin = mtile0 - mb * a.mtp;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int oy0 = a.oy_base + ty * 8 + (wave >> 1) * 4, ox0 = tx * 32 + (wave & 1) * 16;
    if (oy0 >= a.oy_end || ox0 >= a.Wo) return;          // this wave's 4x16 tile lies outside the plane (no barriers here)
    const int iy0 = oy0 - a.pad_h, ix0 = ox0 - a.pad_w;
    float *wlds = lds + wave * NBUF * BUFSZ;

    // BSUM: `tot` starts as the bias and takes every chunk's sum when it is complete (S_0 + bias = bias + S_0 exactly).
    // With 7 cout tiles the two sets are 224 registers: `tot` is pinned to the accumulation registers (AGPRs) by giving
    // every access of it an "a" operand — left to itself the allocator keeps both sets in VGPRs and spills inside the loop.
    constexpr bool TOT_AGPR = BSUM && MT >= 7;
    f32x4 acc[MT][NT];
    f32x4 tot[BSUM && !TOT_AGPR ? MT : 1][BSUM && !TOT_AGPR ? NT : 1];
    float tota[TOT_AGPR ? MT : 1][TOT_AGPR ? NT : 1][4];
    {
        const float *bp = a.bp + (size_t)mtile0 * 16 + 4 * (lane >> 4);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x4 b = *(const f32x4 *)(bp + mt * 16);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if constexpr (TOT_AGPR) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) asm("v_accvgpr_write_b32 %0, %1" : "=a"(tota[mt][nt][e]) : "v"(b[e]));
                } else if constexpr (BSUM) {
                    tot[mt][nt] = b;
                }
                acc[mt][nt] = b;
            }
        }
    }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // staging: lane owns float4 slots e = lane + 64*j of this wave's patch (pixel e>>2, channel quad e&3); bounds and
    // byte offsets inside image n are fixed for the tile, slots outside the image stay zero for ever
    bool ok[MAXP];
    unsigned goff[MAXP];
    f32x4 pre[MAXP];
#pragma unroll
    for (int j = 0; j < MAXP; ++j) {
        const int e = lane + 64 * j;
        const int pix = e >> 2, part = e & 3;
        const int ly = pix / LW, lx = pix - ly * LW;
        const int gy = iy0 + ly, gx = ix0 + lx;
        ok[j] = e < E && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        goff[j] = ok[j] ? (unsigned)(((gy * a.W + gx) * a.Cin + part * 4) * 4) : 0u;
        pre[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const char *xn = (const char *)(a.x + (size_t)n * a.H * a.W * a.Cin);
    auto fetch = [&](int cb) {
        const char *base = xn + cb * (CB * 4);
#pragma unroll
        for (int j = 0; j < MAXP; ++j)
            if (ok[j]) pre[j] = *(const f32x4 *)(base + goff[j]);
    };
    float *sdst = wlds + (lane >> 2) * CP + (lane & 3) * 4;
    auto stash = [&](int buf) {
        float *d0 = sdst + buf * BUFSZ;
#pragma unroll
        for (int j = 0; j < MAXP; ++j) {
            if (64 * j + 63 < E || lane + 64 * j < E) {
                float2 *dst = (float2 *)(d0 + j * 16 * CP);
                dst[0] = make_float2(pre[j].x, pre[j].y);
                dst[1] = make_float2(pre[j].z, pre[j].w);
            }
        }
    };
    const int wstride = a.mtp * 256;
    const float *wq = a.wp + (size_t)mb * a.ncb * 9 * wstride + mtin * 256 + lane * 4;   // walks [cb][tap]
    const long wsteps = (long)a.ncb * 9;
    f32x4 wf[2][MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) wf[0][mt] = *(const f32x4 *)(wq + mt * 256);
    fetch(0);
    stash(0);
    long wstep = 0;
    const float *bl = wlds + (lane & 15) * CP + (lane >> 4);
    float b0[NT], b1[NT];
    for (int cb = 0; cb < a.ncb; ++cb) {
        const float *cur = bl + (NBUF == 2 ? (cb & 1) * BUFSZ : 0);
        const bool more = cb + 1 < a.ncb;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b0[nt] = cur[nt * LW * CP];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int toff = ((t / 3) * LW + (t % 3)) * CP;
            const int tnext = (((t + 1) / 3) * LW + ((t + 1) % 3)) * CP;
            ++wstep;
            {   // next tap's weight fragments into the other set (the last step re-reads its own block: no branch)
                const long wn = wstep < wsteps ? wstep : wsteps - 1;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) wf[(t + 1) & 1][mt] = *(const f32x4 *)(wq + wn * wstride + mt * 256);
            }
            // Vector-memory loads return in order: the next chunk's patch (HBM, microseconds) is requested AFTER tap 1's
            // weights so that it cannot hold them back; the first loads queued behind it are tap 2's, needed two taps later.
            // (one patch load per tap instead, each in front of a tap's weights: 114 instead of 142 TFLOP/s)
            if (t == 0 && more) fetch(cb + 1);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b1[nt] = cur[nt * LW * CP + toff + 4];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t & 1][mt][0], b0[nt],
                                                                        (BSUM && t == 0) ? zero4 : acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b0[nt] = cur[nt * LW * CP + toff + 8];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t & 1][mt][1], b1[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b1[nt] = cur[nt * LW * CP + toff + 12];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t & 1][mt][2], b0[nt], acc[mt][nt], 0, 0, 0);
            if (t < 8) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) b0[nt] = cur[nt * LW * CP + tnext];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t & 1][mt][3], b1[nt], acc[mt][nt], 0, 0, 0);
        }
        // nine taps: the next chunk's first fragments arrived in set 1
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) wf[0][mt] = wf[1][mt];
        if constexpr (BSUM) {       // the chunk's sums are complete: (bias + S_0), + S_1, ...; the next chunk starts from zero
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if constexpr (TOT_AGPR) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float t;
                            asm("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(tota[mt][nt][e]));
                            t = t + acc[mt][nt][e];
                            asm("v_accvgpr_write_b32 %0, %1" : "=a"(tota[mt][nt][e]) : "v"(t));
                        }
                    } else {
                        tot[mt][nt] = tot[mt][nt] + acc[mt][nt];
                    }
                }
        }
        // the other buffer was last read in chunk cb-1 by this same wave (NBUF = 1: this buffer, all of whose reads have
        // been issued by now): in-order LDS makes the overwrite safe
        if (more) stash(NBUF == 2 ? ((cb + 1) & 1) : 0);
    }
    if constexpr (BSUM) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if constexpr (TOT_AGPR) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) asm("v_accvgpr_read_b32 %0, %1" : "=v"(acc[mt][nt][e]) : "a"(tota[mt][nt][e]));
                } else {
                    acc[mt][nt] = tot[mt][nt];
                }
            }
    }

    PM_EPILOGUE_NOTRANS(a,
_Pragma("unroll")
    for (int nt = 0; nt < NT; ++nt) {
        const int oy = oy0 + nt, ox = ox0 + (lane & 15);
        if (oy >= a.oy_end || ox >= a.Wo) continue;
        const size_t pbase = (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.Cout;
_Pragma("unroll")
        for (int mt = 0; mt < MT; ++mt) {
            const int co = (mtile0 + mt) * 16 + 4 * (lane >> 4);
            if (co >= a.Cout) continue;
            store_frag<ACT, RES>(a, acc[mt][nt], pbase, co);
        }
    })
}

template <int MT, int NBUF, bool BSUM = false, int MOFF = -1>
__global__ __launch_bounds__(256, (BSUM && MT >= 7) ? 1 : (NBUF == 1 ? 3 : 2)) void conv3x3s1_wave_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    conv3x3s1_wave_body<MT, NBUF, BSUM, MOFF>(a, blockIdx.x, lds);
}

// ---------------------------------------------------------------------------------------------------
// 3x3 / stride 1 specialisation of the pipelined 4x16-tile variant (mid-size planes and the remainder rows of the big
// launches): same treatment as conv3x3s1_wave_kernel — unrolled taps, alternating weight-fragment sets, LDS addresses
// = one register + immediates, staging offsets and bounds computed once — with the workgroup-shared, double-buffered
// 6x18 patch and one barrier per 16-channel chunk of conv_mfma_pipe_kernel<MT,1,1,.>.  A wave owns one tile row.
// S = 2: the stride-2 form of the quarter-resolution context convolutions (9x33 patch).
template <int MT, int S = 1, bool BSUM = false>
__global__ __launch_bounds__(256, 2) void conv3x3s1_pipe_kernel(ConvArgs a) {
    constexpr int LH = 3 * S + 3, LW = 15 * S + 3;
    constexpr int BUFSZ = LH * LW * CP;
    constexpr int E = LH * LW * 4, MAXP = (E + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x;
    const int n = blockIdx.y;
    const int mtile0 = blockIdx.z * MT;
    const int mb = mtile0 / a.mtp, mtin = mtile0 - mb * a.mtp;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int oy0 = a.oy_base + ty * 4, ox0 = tx * 16;
    const int iy0 = oy0 * S - a.pad_h, ix0 = ox0 * S - a.pad_w;

    f32x4 acc[MT];
    f32x4 tot[BSUM ? MT : 1];          // rule "blocks": starts as the bias, takes every chunk's sum when it is complete
    {
        const float *bp = a.bp + (size_t)mtile0 * 16 + 4 * (lane >> 4);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            acc[mt] = *(const f32x4 *)(bp + mt * 16);
            if constexpr (BSUM) tot[mt] = acc[mt];
        }
    }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    bool ok[MAXP];
    unsigned goff[MAXP];
    f32x4 pre[MAXP];
#pragma unroll
    for (int j = 0; j < MAXP; ++j) {
        const int e = tid + 256 * j;
        const int pix = e >> 2, part = e & 3;
        const int ly = pix / LW, lx = pix - ly * LW;
        const int gy = iy0 + ly, gx = ix0 + lx;
        ok[j] = e < E && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        goff[j] = ok[j] ? (unsigned)(((gy * a.W + gx) * a.Cin + part * 4) * 4) : 0u;
        pre[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const char *xn = (const char *)(a.x + (size_t)n * a.H * a.W * a.Cin);
    auto fetch = [&](int cb) {
        const char *base = xn + cb * (CB * 4);
#pragma unroll
        for (int j = 0; j < MAXP; ++j)
            if (ok[j]) pre[j] = *(const f32x4 *)(base + goff[j]);
    };
    float *sdst = lds + (tid >> 2) * CP + (tid & 3) * 4;
    auto stash = [&](int buf) {
        float *d0 = sdst + buf * BUFSZ;
#pragma unroll
        for (int j = 0; j < MAXP; ++j) {
            if (256 * j + 255 < E || tid + 256 * j < E) {
                float2 *dst = (float2 *)(d0 + j * 64 * CP);
                dst[0] = make_float2(pre[j].x, pre[j].y);
                dst[1] = make_float2(pre[j].z, pre[j].w);
            }
        }
    };
    const int wstride = a.mtp * 256;
    const float *wq = a.wp + (size_t)mb * a.ncb * 9 * wstride + mtin * 256 + lane * 4;   // walks [cb][tap]
    const long wsteps = (long)a.ncb * 9;
    f32x4 wf[2][MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) wf[0][mt] = *(const f32x4 *)(wq + mt * 256);
    fetch(0);
    stash(0);
    __syncthreads();
    long wstep = 0;
    const float *bl = lds + (wave * S * LW + (lane & 15) * S) * CP + (lane >> 4);
    float b0, b1;
    for (int cb = 0; cb < a.ncb; ++cb) {
        const float *cur = bl + (cb & 1) * BUFSZ;
        const bool more = cb + 1 < a.ncb;
        b0 = cur[0];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int toff = ((t / 3) * LW + (t % 3)) * CP;
            const int tnext = (((t + 1) / 3) * LW + ((t + 1) % 3)) * CP;
            ++wstep;
            {
                const long wn = wstep < wsteps ? wstep : wsteps - 1;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) wf[(t + 1) & 1][mt] = *(const f32x4 *)(wq + wn * wstride + mt * 256);
            }
            if (t == 0 && more) fetch(cb + 1);         // after tap 1's weights: loads return in order (see the wave kernel)
            b1 = cur[toff + 4];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t & 1][mt][0], b0, (BSUM && t == 0) ? zero4 : acc[mt], 0, 0, 0);
            b0 = cur[toff + 8];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t & 1][mt][1], b1, acc[mt], 0, 0, 0);
            b1 = cur[toff + 12];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t & 1][mt][2], b0, acc[mt], 0, 0, 0);
            if (t < 8) b0 = cur[tnext];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t & 1][mt][3], b1, acc[mt], 0, 0, 0);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) wf[0][mt] = wf[1][mt];
        if constexpr (BSUM) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) tot[mt] = tot[mt] + acc[mt];
        }
        if (more) stash((cb + 1) & 1);
        __syncthreads();
    }
    if constexpr (BSUM) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = tot[mt];
    }

    {
        const int oy = oy0 + wave, ox = ox0 + (lane & 15);
        if (oy < a.oy_end && ox < a.Wo) {
            const size_t pbase = (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.Cout;
            PM_EPILOGUE_NOTRANS(a,
_Pragma("unroll")
            for (int mt = 0; mt < MT; ++mt) {
                const int co = (mtile0 + mt) * 16 + 4 * (lane >> 4);
                if (co >= a.Cout) continue;
                store_frag<ACT, RES>(a, acc[mt], pbase, co);
            })
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// 7x7 / stride 1 on 8x32 tiles (SpyNet's 32->64, 64->32, 32->16 layers on the large pyramid levels): the pipelined
// workgroup-shared variant with the treatment of the 3x3 kernels above — the seven taps of a filter row unrolled with
// alternating weight-fragment sets (one copy per filter row), LDS addresses = one register per filter row + immediates,
// staging offsets and bounds computed once per tile, the next chunk's patch fetched into registers while the current
// one is multiplied (the generic single-buffer kernel waits for every chunk's loads and spends a quarter of its issue
// slots on address arithmetic when MT = 1).  Same sums, same order.
// NT = 4: 8x32 tiles (a wave owns two rows); NT = 1: 4x16 tiles (a wave owns one 16-pixel row) for the small pyramid
// levels, where the generic kernel waits for every tap's weights (105 us on a 36x60 plane that holds 3 us of work).
template <int MT, int NT = 4, bool BSUM = false>
__global__ __launch_bounds__(256, 2) void conv7x7s1_pipe_kernel(ConvArgs a) {
    constexpr int K = 7, TH = NT == 4 ? 8 : 4, TW = NT == 4 ? 32 : 16, LH = TH + K - 1, LW = TW + K - 1;
    constexpr int BUFSZ = LH * LW * CP;
    constexpr int E = LH * LW * 4, MAXP = (E + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x;
    const int n = blockIdx.y;
    const int mtile0 = blockIdx.z * MT;
    const int mb = mtile0 / a.mtp, mtin = mtile0 - mb * a.mtp;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int oy0 = a.oy_base + ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 - a.pad_h, ix0 = ox0 - a.pad_w;

    f32x4 acc[MT][NT];
    f32x4 tot[BSUM ? MT : 1][BSUM ? NT : 1];            // rule 1: running output, a chunk's sum is added when complete
    {
        const float *bp = a.bp + (size_t)mtile0 * 16 + 4 * (lane >> 4);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x4 b = BSUM ? f32x4{0.f, 0.f, 0.f, 0.f} : *(const f32x4 *)(bp + mt * 16);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = b;
        }
    }
    bool ok[MAXP];
    unsigned goff[MAXP];
    f32x4 pre[MAXP];
#pragma unroll
    for (int j = 0; j < MAXP; ++j) {
        const int e = tid + 256 * j;
        const int pix = e >> 2, part = e & 3;
        const int ly = pix / LW, lx = pix - ly * LW;
        const int gy = iy0 + ly, gx = ix0 + lx;
        ok[j] = e < E && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        goff[j] = ok[j] ? (unsigned)(((gy * a.W + gx) * a.Cin + part * 4) * 4) : 0u;
        pre[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const char *xn = (const char *)(a.x + (size_t)n * a.H * a.W * a.Cin);
    auto fetch = [&](int cb) {
        const char *base = xn + cb * (CB * 4);
#pragma unroll
        for (int j = 0; j < MAXP; ++j)
            if (ok[j]) pre[j] = *(const f32x4 *)(base + goff[j]);
    };
    float *sdst = lds + (tid >> 2) * CP + (tid & 3) * 4;
    auto stash = [&](int buf) {
        float *d0 = sdst + buf * BUFSZ;
#pragma unroll
        for (int j = 0; j < MAXP; ++j) {
            if (256 * j + 255 < E || tid + 256 * j < E) {
                float2 *dst = (float2 *)(d0 + j * 64 * CP);
                dst[0] = make_float2(pre[j].x, pre[j].y);
                dst[1] = make_float2(pre[j].z, pre[j].w);
            }
        }
    };
    const int wstride = a.mtp * 256;
    const float *wq = a.wp + (size_t)mb * a.ncb * (K * K) * wstride + mtin * 256 + lane * 4;   // walks [cb][tap]
    const long wsteps = (long)a.ncb * (K * K);
    f32x4 wf[2][MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) wf[0][mt] = *(const f32x4 *)(wq + mt * 256);
    fetch(0);
    stash(0);
    __syncthreads();
    long wstep = 0;
    // NT = 4: segment nt of this wave = row 2*wave + (nt >> 1), columns 16*(nt & 1) ..; NT = 1: row wave
    const float *bl = lds + (((NT == 4 ? 2 : 1) * wave) * LW + (lane & 15)) * CP + (lane >> 4);
    constexpr int SEG[4] = {0, 16 * CP, LW * CP, LW * CP + 16 * CP};
    float b0[NT], b1[NT];
    for (int cb = 0; cb < a.ncb; ++cb) {
        const float *cur = bl + (cb & 1) * BUFSZ;
        const bool more = cb + 1 < a.ncb;
#pragma unroll 1
        for (int ky = 0; ky < K; ++ky) {
            const float *row = cur + ky * (LW * CP);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b0[nt] = row[SEG[nt]];
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const int toff = kx * CP;
                ++wstep;
                {
                    const long wn = wstep < wsteps ? wstep : wsteps - 1;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        wf[(kx + 1) & 1][mt] = *(const f32x4 *)(wq + wn * wstride + mt * 256);
                }
                if (kx == 0 && ky == 0 && more) fetch(cb + 1);     // after the next tap's weights: loads return in order
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) b1[nt] = row[SEG[nt] + toff + 4];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[kx & 1][mt][0], b0[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) b0[nt] = row[SEG[nt] + toff + 8];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[kx & 1][mt][1], b1[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) b1[nt] = row[SEG[nt] + toff + 12];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[kx & 1][mt][2], b0[nt], acc[mt][nt], 0, 0, 0);
                if (kx + 1 < K) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) b0[nt] = row[SEG[nt] + toff + CP];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[kx & 1][mt][3], b1[nt], acc[mt][nt], 0, 0, 0);
            }
            // seven taps: the next row's first fragments arrived in set 1
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) wf[0][mt] = wf[1][mt];
        }
        if constexpr (BSUM) {
            const float *bp = a.bp + (size_t)mtile0 * 16 + 4 * (lane >> 4);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const f32x4 b = *(const f32x4 *)(bp + mt * 16);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    tot[mt][nt] = cb == 0 ? acc[mt][nt] + b : tot[mt][nt] + acc[mt][nt];
                    acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        if (more) stash((cb + 1) & 1);
        __syncthreads();
    }
    if constexpr (BSUM) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = tot[mt][nt];
    }

    PM_EPILOGUE_NOTRANS(a,
_Pragma("unroll")
    for (int nt = 0; nt < NT; ++nt) {
        const int oy = oy0 + (NT == 4 ? 2 * wave + (nt >> 1) : wave), ox = ox0 + 16 * (nt & 1) + (lane & 15);
        if (oy >= a.oy_end || ox >= a.Wo) continue;
        const size_t pbase = (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.Cout;
_Pragma("unroll")
        for (int mt = 0; mt < MT; ++mt) {
            const int co = (mtile0 + mt) * 16 + 4 * (lane >> 4);
            if (co >= a.Cout) continue;
            store_frag<ACT, RES>(a, acc[mt][nt], pbase, co);
        }
    })
}


// ---------------------------------------------------------------------------------------------------
// Resident-patch variant for small planes (same arithmetic, same sum order): the workgroup stages the input patch of
// its 4x16 tile for ALL input channels at once (one global-load latency instead of one per 16-channel chunk, a single
// barrier), then every wave runs its 16 pixels x MT cout tiles straight through.  Small planes are latency-bound in
// the chunked variants: a chunk's 36 MFMAs per wave are shorter than the load that feeds the next chunk.
// Measured (tools/bench_conv.py, knob RES): with one cout tile per workgroup the 7x re-staged patch saturates L2->LDS
// (56 vs 78 TFLOP/s on 144x240); with all cout tiles (RES=2) it wins only on 2x144x240 (93 vs 82).  Off by default.
template <int MT>
__global__ __launch_bounds__(256) void conv_mfma_res_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int tile = blockIdx.x;
    const int n = blockIdx.y;
    const int mtile0 = blockIdx.z * MT;
    const int mb = mtile0 / a.mtp, mtin = mtile0 - mb * a.mtp;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int oy0 = a.oy_base + ty * 4, ox0 = tx * 16;
    const int LH = 3 * a.S + a.KH, LW = 15 * a.S + a.KW;
    const int iy0 = oy0 * a.S - a.pad_h, ix0 = ox0 * a.S - a.pad_w;
    const int taps = a.KH * a.KW;
    const int CPR = a.Cin + 2;                // LDS pixel stride in words (even: 8-byte aligned rows of float2)
    const int q4 = a.Cin >> 2;                // float4 units per pixel
    const int E = LH * LW * q4;

    // stage the whole patch; 8 loads in flight per thread and pass.  Element e = tid + 256*k is float4 `part` of patch
    // pixel (ly, lx); the three indices advance incrementally (no divisions in the loop).
    {
        const int dpix = 256 / q4, dpart = 256 - dpix * q4;      // e += 256  ->  pix += dpix, part += dpart (+ carry)
        const int drow = dpix / LW, dcol = dpix - drow * LW;      // pix += dpix ->  ly += drow, lx += dcol (+ carry)
        int pix = tid / q4, part = tid - pix * q4;
        int ly = pix / LW, lx = pix - ly * LW;
        for (int e0 = 0; e0 < E; e0 += 256 * 8) {
            f32x4 pre[8];
            int spix[8], spart[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int e = e0 + tid + 256 * j;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                spix[j] = pix; spart[j] = part;
                if (e < E) {
                    const int gy = iy0 + ly, gx = ix0 + lx;
                    if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                        v = *(const f32x4 *)(a.x + (((size_t)n * a.H + gy) * a.W + gx) * a.Cin + part * 4);
                }
                pre[j] = v;
                part += dpart; pix += dpix; lx += dcol; ly += drow;
                if (part >= q4) { part -= q4; ++pix; ++lx; }
                if (lx >= LW) { lx -= LW; ++ly; }
                if (lx >= LW) { lx -= LW; ++ly; }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int e = e0 + tid + 256 * j;
                if (e < E) {
                    float2 *dst = (float2 *)(lds + spix[j] * CPR + spart[j] * 4);
                    dst[0] = make_float2(pre[j].x, pre[j].y);
                    dst[1] = make_float2(pre[j].z, pre[j].w);
                }
            }
        }
    }
    f32x4 acc[MT];
    {
        const float *bp = a.bp + (size_t)mtile0 * 16 + 4 * (lane >> 4);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = *(const f32x4 *)(bp + mt * 16);
    }
    const int wstride = a.mtp * 256;
    const float *wq = a.wp + (size_t)mb * a.ncb * taps * wstride + mtin * 256 + lane * 4;   // walks [cb][tap]
    const long wsteps = (long)a.ncb * taps;
    f32x4 a_cur[MT], a_nxt[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a_cur[mt] = *(const f32x4 *)(wq + mt * 256);
    const float *bl = lds + ((wave * a.S) * LW + (lane & 15) * a.S) * CPR + (lane >> 4);
    __syncthreads();
    long wstep = 0;
    for (int cb = 0; cb < a.ncb; ++cb) {
        int ky = 0, kx = 0;
        for (int tap = 0; tap < taps; ++tap) {
            ++wstep;
            {
                const long wn = wstep < wsteps ? wstep : wsteps - 1;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a_nxt[mt] = *(const f32x4 *)(wq + wn * wstride + mt * 256);
            }
            const float *bb = bl + (ky * LW + kx) * CPR + cb * CB;
            if (++kx == a.KW) { kx = 0; ++ky; }
            const float b0 = bb[0], b1 = bb[4], b2 = bb[8], b3 = bb[12];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[mt][0], b0, acc[mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[mt][1], b1, acc[mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[mt][2], b2, acc[mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[mt][3], b3, acc[mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a_cur[mt] = a_nxt[mt];
        }
    }

    const int oy = oy0 + wave, ox = ox0 + (lane & 15);
    if (oy >= a.oy_end || ox >= a.Wo) return;
    const size_t pbase = (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.Cout;
    PM_EPILOGUE_NOTRANS(a,
_Pragma("unroll")
    for (int mt = 0; mt < MT; ++mt) {
        const int co = (mtile0 + mt) * 16 + 4 * (lane >> 4);
        if (co >= a.Cout) continue;
        store_frag<ACT, RES>(a, acc[mt], pbase, co);
    })
}


// ---------------------------------------------------------------------------------------------------
// 16 -> 16 channels, 3x3 (the PredictUpdate trunk, lifting_1d.py:39-40, 45-47): one chunk, one cout tile, so all 36
// weight fragments stay in registers and a wave is PERSISTENT: it walks over 4x16-pixel tiles (grid-stride), fetching
// the next tile's patch into registers while the current one is multiplied, wave-private LDS, no barriers.  The layer
// moves 128 B per pixel for 9.2 kFLOP, i.e. it is HBM-bound once the weights stop being re-read.  Same sums, same order.
__global__ __launch_bounds__(256) void conv16_persistent_kernel(ConvArgs a, int tiles_total, int bsum) {
    constexpr int NT = 4, LH = 6, LW = 18, E = LH * LW * 4, MAXP = (E + 63) / 64;      // 432 float4 slots, 7 per lane
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int bufsz = LH * LW * CP;
    float *wlds = lds + wave * 2 * bufsz;
    const int tiles_x = a.tiles_x, tiles_y = a.tiles_y;         // in 4x16 wave tiles
    const int per_img = tiles_x * tiles_y;

    f32x4 af[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) af[t] = *(const f32x4 *)(a.wp + t * 256 + lane * 4);
    const f32x4 bias = *(const f32x4 *)(a.bp + 4 * (lane >> 4));
    int boff[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) boff[nt] = (nt * LW + (lane & 15)) * CP + (lane >> 4);

    // staging slots e = lane + 64*j: patch pixel (ly, lx) and byte offset relative to the patch origin are the same for
    // every tile; per tile only the (wave-uniform, scalar) origin moves and two bounds compares per slot remain
    f32x4 pre[MAXP];
    int sly[MAXP], slx[MAXP], rel[MAXP];
#pragma unroll
    for (int j = 0; j < MAXP; ++j) {
        const int e = lane + 64 * j;
        const int pix = e >> 2, part = e & 3;
        sly[j] = pix / LW;
        slx[j] = pix - sly[j] * LW;
        rel[j] = ((sly[j] * a.W + slx[j]) * 16 + part * 4) * 4;
        if (e >= E) sly[j] = -(1 << 20);                   // never inside the image
    }
    auto fetch = [&](int tile) {
        const int n = tile / per_img, r = tile - n * per_img;
        const int ty = r / tiles_x, tx = r - ty * tiles_x;
        const int iy0 = ty * 4 - 1, ix0 = tx * 16 - 1;
        const char *base = (const char *)a.x + (((long)n * a.H + iy0) * a.W + ix0) * 64;     // scalar; may lie before row 0
#pragma unroll
        for (int j = 0; j < MAXP; ++j) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)(iy0 + sly[j]) < (unsigned)a.H && (unsigned)(ix0 + slx[j]) < (unsigned)a.W)
                v = *(const f32x4 *)(base + rel[j]);
            pre[j] = v;
        }
    };
    auto stash = [&](float *buf) {
#pragma unroll
        for (int j = 0; j < MAXP; ++j) {
            const int e = lane + 64 * j;
            if (e < E) {
                float2 *dst = (float2 *)(buf + (e >> 2) * CP + (e & 3) * 4);
                dst[0] = make_float2(pre[j].x, pre[j].y);
                dst[1] = make_float2(pre[j].z, pre[j].w);
            }
        }
    };

    const int stride = gridDim.x * WAVES;
    int tile = blockIdx.x * WAVES + wave;
    if (tile >= tiles_total) return;
    fetch(tile);
    stash(wlds);
    int parity = 0;
    for (; tile < tiles_total; tile += stride) {
        const float *cur = wlds + parity * bufsz;
        const bool more = tile + stride < tiles_total;
        if (more) fetch(tile + stride);
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = bsum ? f32x4{0.f, 0.f, 0.f, 0.f} : bias;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float *bb = cur + ((t / 3) * LW + (t % 3)) * CP;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                float b[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) b[nt] = bb[boff[nt] + ks * 4];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[t][ks], b[nt], acc[nt], 0, 0, 0);
            }
        }
        if (bsum) {             // rule 1, one chunk: the chain ran from zero, the bias is added last
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = acc[nt] + bias;
        }
        {
            const int n = tile / per_img, r = tile - n * per_img;
            const int ty = r / tiles_x, tx = r - ty * tiles_x;
            const int ox = tx * 16 + (lane & 15);
            const int co = 4 * (lane >> 4);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int oy = ty * 4 + nt;
                if (oy < a.Ho && ox < a.Wo) {
                    const size_t o = (((size_t)n * a.Ho + oy) * a.Wo + ox) * 16 + co;
                    f32x4 v = acc[nt];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = pm::apply_act(v[i], a.act, a.slope);
                    if (a.res1) { const f32x4 r1 = *(const f32x4 *)(a.res1 + o); v = v + r1; }
                    if (a.res2) { const f32x4 r2 = *(const f32x4 *)(a.res2 + o); v = v + r2; }
                    *(f32x4 *)(a.y + o) = v;
                }
            }
        }
        if (more) stash(wlds + (parity ^ 1) * bufsz);
        parity ^= 1;
    }
}


// The same walk at a higher occupancy (round 3): the patch of the NEXT tile waits in registers until the matrix work of
// the current one has been issued, so ONE wave-private LDS buffer is enough (8.6 KB per wave instead of 17: OCC workgroups
// per CU instead of two); tiles inside the image take their patch without per-slot bounds tests; and each XCD walks its
// own contiguous band of tiles, so the halo rows of a tile are found in the L2 that fetched them one round earlier.
// Same sums, same order.
template <int OCC>
__global__ __launch_bounds__(256, OCC) void conv16_band_kernel(ConvArgs a, int tiles_total, int bsum) {
    constexpr int NT = 4, LH = 6, LW = 18, E = LH * LW * 4, MAXP = (E + 63) / 64;      // 432 float4 slots, 7 per lane
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ uint4 tanh_tab[pm::TANH_LDS_UINT4];              // the PredictUpdate trunk's activation (one barrier, at the start)
    if (a.act == pm::ACT_TANH) {                                // wave-uniform
        pm::tanh_rows_to_lds(tanh_tab, threadIdx.x, 256);
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int bufsz = LH * LW * CP;
    float *wlds = lds + wave * bufsz;
    const int tiles_x = a.tiles_x, tiles_y = a.tiles_y;         // in 4x16 wave tiles
    const int per_img = tiles_x * tiles_y;

    f32x4 af[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) af[t] = *(const f32x4 *)(a.wp + t * 256 + lane * 4);
    const f32x4 bias = *(const f32x4 *)(a.bp + 4 * (lane >> 4));
    const int boff = (lane & 15) * CP + (lane >> 4);
    const unsigned yoff = ((lane & 15) * 16 + 4 * (lane >> 4)) * 4;       // byte offset of this lane inside a 16-pixel output row

    f32x4 pre[MAXP];
    unsigned rel[MAXP];
#pragma unroll
    for (int j = 0; j < MAXP; ++j) {
        const int e = lane + 64 * j;
        const int pix = e >> 2, part = e & 3;
        const int ly = pix / LW, lx = pix - ly * LW;
        rel[j] = ((ly * a.W + lx) * 16 + part * 4) * 4;
    }
    auto fetch = [&](int tile) {
        const int n = tile / per_img, r = tile - n * per_img;
        const int ty = r / tiles_x, tx = r - ty * tiles_x;
        const int iy0 = ty * 4 - 1, ix0 = tx * 16 - 1;
        if (iy0 >= 0 && ix0 >= 0 && iy0 + LH <= a.H && ix0 + LW <= a.W) {                      // wave-uniform: inside
            const char *base = (const char *)a.x + (((long)n * a.H + iy0) * a.W + ix0) * 64;     // scalar
#pragma unroll
            for (int j = 0; j < MAXP; ++j)
                if (j < MAXP - 1 || lane + 64 * j < E) pre[j] = *(const f32x4 *)(base + rel[j]);
        } else {
            const char *img = (const char *)a.x + (long)n * a.H * a.W * 64;
#pragma unroll
            for (int j = 0; j < MAXP; ++j) {
                const int e = lane + 64 * j;
                const int pix = e >> 2;
                const int ly = pix / LW, lx = pix - ly * LW;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (e < E && (unsigned)(iy0 + ly) < (unsigned)a.H && (unsigned)(ix0 + lx) < (unsigned)a.W)
                    v = *(const f32x4 *)(img + ((long)(iy0 + ly) * a.W + (ix0 + lx)) * 64 + (e & 3) * 16);
                pre[j] = v;
            }
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int j = 0; j < MAXP; ++j) {
            const int e = lane + 64 * j;
            if (e < E) {
                float2 *dst = (float2 *)(wlds + (e >> 2) * CP + (e & 3) * 4);
                dst[0] = make_float2(pre[j].x, pre[j].y);
                dst[1] = make_float2(pre[j].z, pre[j].w);
            }
        }
    };
    f32x4 acc[NT];
    // results of a tile leave one tile later, behind the patch of the next one: the wait for that patch (the newest
    // vector-memory operations of the wave) then never includes stores that have just been issued
    auto finish = [&](int tile) {
        const int n = tile / per_img, r = tile - n * per_img;
        const int ty = r / tiles_x, tx = r - ty * tiles_x;
        const size_t row0 = (((size_t)n * a.Ho + ty * 4) * a.Wo + tx * 16) * 16;          // scalar, in floats
        if (ty * 4 + NT <= a.Ho && tx * 16 + 16 <= a.Wo && a.act == pm::ACT_NONE && !a.res2) {     // wave-uniform
            char *yb = (char *)(a.y + row0);
            if (a.res1) {
                const char *rb = (const char *)(a.res1 + row0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const f32x4 r1 = *(const f32x4 *)(rb + (unsigned)(nt * a.Wo * 64) + yoff);
                    *(f32x4 *)(yb + (unsigned)(nt * a.Wo * 64) + yoff) = acc[nt] + r1;
                }
            } else {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) *(f32x4 *)(yb + (unsigned)(nt * a.Wo * 64) + yoff) = acc[nt];
            }
            return;
        }
        const int ox = tx * 16 + (lane & 15);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int oy = ty * 4 + nt;
            if (oy < a.Ho && ox < a.Wo) {
                const size_t o = row0 + (size_t)nt * a.Wo * 16 + (yoff >> 2);
                f32x4 v = acc[nt];
                if (a.act == pm::ACT_TANH) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = pm::tanhf_rows(v[i], tanh_tab);
                } else if (a.act != pm::ACT_NONE) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = pm::apply_act(v[i], a.act, a.slope);
                }
                if (a.res1) { const f32x4 r1 = *(const f32x4 *)(a.res1 + o); v = v + r1; }
                if (a.res2) { const f32x4 r2 = *(const f32x4 *)(a.res2 + o); v = v + r2; }
                *(f32x4 *)(a.y + o) = v;
            }
        }
    };

    // band of this XCD (workgroups are dealt round-robin to the 8 XCDs; the grid is a multiple of 8)
    const int xcd = blockIdx.x & 7, wg_in = blockIdx.x >> 3;
    const int per_xcd = (tiles_total + 7) >> 3;
    const int t_end = min(tiles_total, (xcd + 1) * per_xcd);
    const int stride = (gridDim.x >> 3) * WAVES;
    int tile = xcd * per_xcd + wg_in * WAVES + wave;
    if (tile >= t_end) return;
    fetch(tile);
    int prev = -1;
    for (; tile < t_end; tile += stride) {
        stash();
        if (prev >= 0) finish(prev);
        if (tile + stride < t_end) fetch(tile + stride);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = bsum ? f32x4{0.f, 0.f, 0.f, 0.f} : bias;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float *bb = wlds + ((t / 3) * LW + (t % 3)) * CP + boff;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                float b[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) b[nt] = bb[nt * LW * CP + ks * 4];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[t][ks], b[nt], acc[nt], 0, 0, 0);
            }
        }
        if (bsum) {             // rule 1, one chunk: the chain ran from zero, the bias is added last
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = acc[nt] + bias;
        }
        prev = tile;
    }
    finish(prev);
}

// ---------------------------------------------------------------------------------------------------
// 1x1 convolution (stride 1, no padding) as a plain GEMM over the flattened pixels: the MV codec's 192 <-> 768 layers
// on 72x120 planes and the 64 <-> 256 layers on 576x960.  The tap-oriented kernels above stage a 16-channel chunk per
// barrier pair and get 4*MT matrix instructions out of it for a 1x1 filter: they spend their time waiting.  Here a
// workgroup stages 64 channels of 16*NT pixels at a time (double-buffered, one barrier per 64 channels), every wave
// multiplies all the pixels by its own MTW cout tiles (16*MTW*NT matrix instructions per k-chunk and wave), LDS
// addresses are one register plus immediates.  Same sum order as everywhere: acc = bias, channels ascending.
// RB: "reduce-B" (PMCTF_SUM_*: the reduction in blocks of `bchunks` 16-channel chunks, the first block's chain from the
// bias, later blocks from zero, block results added in turn) with a second accumulator set.
template <int MTW, int NT, bool RB = false>
__global__ __launch_bounds__(256) void conv1x1_kernel(ConvArgs a, long P, int tiles_total, int bchunks) {
    constexpr int PX = 16 * NT, KC = 64, KP = KC + 2, SLOTS = PX * (KC / 4) / 256;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long p0 = (long)blockIdx.x * PX;
    const int t0 = (blockIdx.y * WAVES + wave) * MTW;            // first cout tile of this wave
    const int nchunk = (a.ncb + 3) >> 2;

    f32x4 acc[MTW][NT];
    f32x4 tot[RB ? MTW : 1][RB ? NT : 1];
    int left = bchunks;                                          // 16-channel chunks until the current reduction block ends
    bool first = true;
    const float *wt[MTW];                                        // fragment base of each tile: + cb * mtp * 256 per chunk
    bool live[MTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        const int t = t0 + i;
        live[i] = t < tiles_total;
        const int tt = live[i] ? t : 0;
        const int mb = tt / a.mtp, mtin = tt - mb * a.mtp;
        wt[i] = a.wp + ((size_t)mb * a.ncb * a.mtp + mtin) * 256 + lane * 4;
        const f32x4 b = *(const f32x4 *)(a.bp + (size_t)tt * 16 + 4 * (lane >> 4));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[i][nt] = b;
    }
    const int wstride = a.mtp * 256;

    // staging: thread owns float4 slots e = tid + 256*j: pixel e >> 4, channel quad e & 15 of the 64-channel chunk
    f32x4 pre[SLOTS];
    auto fetch = [&](int c) {
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) {
            const int e = tid + 256 * j;
            const long p = p0 + (e >> 4);
            const int ch = c * KC + (e & 15) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (p < P && ch < a.Cin) v = *(const f32x4 *)(a.x + p * a.Cin + ch);
            pre[j] = v;
        }
    };
    auto stash = [&](float *buf) {
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) {
            const int e = tid + 256 * j;
            float2 *dst = (float2 *)(buf + (e >> 4) * KP + (e & 15) * 4);
            dst[0] = make_float2(pre[j].x, pre[j].y);
            dst[1] = make_float2(pre[j].z, pre[j].w);
        }
    };
    fetch(0);
    stash(lds);
    __syncthreads();
    f32x4 af[2][MTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i) af[0][i] = *(const f32x4 *)(wt[i]);
    const float *bl = lds + (lane & 15) * KP + (lane >> 4);
    for (int c = 0; c < nchunk; ++c) {
        const float *cur = bl + (c & 1) * (PX * KP);
        const bool more = c + 1 < nchunk;
        if (more) fetch(c + 1);
        const int nsub = a.ncb - 4 * c < 4 ? a.ncb - 4 * c : 4;      // 16-channel blocks in this chunk (uniform)
#pragma unroll
        for (int sb = 0; sb < 4; ++sb) {
            if (sb < nsub) {
                if constexpr (RB) {
                    if (left == 0) {                              // a reduction block is complete
#pragma unroll
                        for (int i = 0; i < MTW; ++i)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) {
                                tot[i][nt] = first ? acc[i][nt] : tot[i][nt] + acc[i][nt];
                                acc[i][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                            }
                        first = false;
                        left = bchunks;
                    }
                    --left;
                }
                {   // next 16-channel block's fragments (the last block re-reads its own: no branch)
                    const int cbn = 4 * c + sb + 1 < a.ncb ? 4 * c + sb + 1 : a.ncb - 1;
#pragma unroll
                    for (int i = 0; i < MTW; ++i) af[(sb + 1) & 1][i] = *(const f32x4 *)(wt[i] + (size_t)cbn * wstride);
                }
                float b[4][NT];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) b[ks][nt] = cur[nt * 16 * KP + sb * 16 + ks * 4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int i = 0; i < MTW; ++i)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[sb & 1][i][ks], b[ks][nt], acc[i][nt], 0, 0, 0);
            }
        }
        if (nsub & 1) {                                           // odd block count: the prefetched set is set 1
#pragma unroll
            for (int i = 0; i < MTW; ++i) af[0][i] = af[1][i];
        }
        if (more) stash(lds + ((c + 1) & 1) * (PX * KP));
        __syncthreads();
    }
    if constexpr (RB) {
        if (!first) {
#pragma unroll
            for (int i = 0; i < MTW; ++i)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[i][nt] = tot[i][nt] + acc[i][nt];
        }
    }

    PM_EPILOGUE_NOTRANS(a,
_Pragma("unroll")
    for (int nt = 0; nt < NT; ++nt) {
        const long p = p0 + nt * 16 + (lane & 15);
        if (p >= P) continue;
        const size_t pbase = (size_t)p * a.Cout;
_Pragma("unroll")
        for (int i = 0; i < MTW; ++i) {
            const int co = (t0 + i) * 16 + 4 * (lane >> 4);
            if (!live[i] || co >= a.Cout) continue;
            store_frag<ACT, RES>(a, acc[i][nt], pbase, co);
        }
    })
}


// choose cout tiles per workgroup (MT in {1,2,4,7,8}) and the number of M-blocks
void choose_mt(int Cout, int &MT, int &MB) {
    const int tiles = (Cout + 15) / 16;
    static const int allowed[5] = {1, 2, 4, 7, 8};
    int best_mt = 8, best_mb = (tiles + 7) / 8, best_cost = 1 << 30;
    for (int i = 0; i < 5; ++i) {
        const int mt = allowed[i];
        const int mb = (tiles + mt - 1) / mt;
        const int cost = mt * mb * 8 + mb;  // padded MFMA work first, then restaging
        if (cost < best_cost) { best_cost = cost; best_mt = mt; best_mb = mb; }
    }
    MT = best_mt; MB = best_mb;
}

// Tuning knobs: environment variable PMCTF_CONV_<NAME> at first use, or pmctf_conv2d_set_option("<NAME>", v).
struct Knob { const char *name; long value; bool set; };
Knob g_knobs[] = {{"WAVE", 1, false}, {"NT", 0, false}, {"MSPLIT_PX", 70000, false}, {"SPLIT", 1, false},
                  {"BIGPX", 131072, false}, {"RES", 0, false}, {"MSPLIT_NT", 1, false}, {"C16", 1, false}, {"C16_WGS", 512, false}, {"C16_OCC", 2, false}, {"V1", 0, false}, {"V2", 0, false}, {"NBUF1", 1, false}, {"K33", 1, false}, {"WAVE_SMALL", 1, false}, {"K11", 1, false}, {"K77", 1, false}, {"K33_SMALL", 1, false}, {"K11_MIN_TILES", 7, false}, {"BIGPX_NOSPLIT", 200000, false}, {"BSUM_SPLIT", 1, false}};
std::once_flag g_knobs_once;
inline long knob(const char *name) {
    // one-time, thread-safe read of the environment (ctypes callers may launch from several host threads)
    std::call_once(g_knobs_once, [] {
        for (Knob &k : g_knobs)
            if (!k.set) {
                char env[64];
                snprintf(env, sizeof env, "PMCTF_CONV_%s", k.name);
                const char *v = getenv(env);
                if (v) k.value = atol(v);
                k.set = true;
            }
    });
    for (Knob &k : g_knobs)
        if (!strcmp(k.name, name)) return k.value;
    return 0;
}

// dynamic LDS above the 64 KB default needs the function attribute: set it once per kernel instantiation, thread-safe
template <typename K>
inline void allow_big_lds(K kernel, std::once_flag &once) {
    std::call_once(once, [kernel] {
        (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
}

// wave-private variant usable? (stride-1-ish patch of one wave must fit 7 staging slots and 80 KB for the workgroup)
inline bool wave_eligible(const ConvArgs &a) {
    const bool use_wave = knob("WAVE") != 0;
    const int PH = 3 * a.S + a.KH, PW = 15 * a.S + a.KW;
    const size_t wsmem = (size_t)PH * PW * CP * sizeof(float) * 2 * WAVES;
    const int wslots = (PH * PW * 4 + 63) / 64;
    return use_wave && (a.Cin % CB) == 0 && wsmem <= 80 * 1024 && wslots <= 7;
}

// One launch over output rows [r0, r1) with MT cout tiles per workgroup (gz = cout-tile groups in grid.z).
// What the last convolution call of this thread launched (kernel expression, tile parameters, grid): bench.py reports the
// kernel behind its roofline figure from here instead of from a string literal.
thread_local char g_last_launch[512];
thread_local int g_last_len = 0;
void note_launch(const char *kernel, int mt, int nt, int tw16, dim3 g) {
    if (g_last_len >= (int)sizeof(g_last_launch) - 1) return;
    const int n = snprintf(g_last_launch + g_last_len, sizeof(g_last_launch) - (size_t)g_last_len,
                           "%s%s [MT=%d NT=%d TW16=%d] grid %ux%ux%u", g_last_len ? " + " : "", kernel, mt, nt, tw16, g.x, g.y, g.z);
    if (n > 0) g_last_len = g_last_len + n < (int)sizeof(g_last_launch) ? g_last_len + n : (int)sizeof(g_last_launch) - 1;
}
#define CONV_LAUNCH(kernel, grid, block, smem, stream, ...)          \
    do {                                                             \
        note_launch(#kernel, MT, NT, TW16, grid);                    \
        PM_LAUNCH(kernel, grid, block, smem, stream, __VA_ARGS__);   \
    } while (0)

template <int MT, int NT, int TW16>
int launch(const ConvArgs &a, int gz, hipStream_t st, int r0, int r1) {
    constexpr int TW = TW16 * 16;
    constexpr int TH = NT * WAVES / TW16;
    ConvArgs b = a;
    b.oy_base = r0;
    b.oy_end = r1;
    b.tiles_x = (a.Wo + TW - 1) / TW;
    b.tiles_y = (r1 - r0 + TH - 1) / TH;
    const int LH = (TH - 1) * a.S + a.KH, LW = (TW - 1) * a.S + a.KW;
    const size_t smem = (size_t)LH * LW * CP * sizeof(float);
    if (smem > 160 * 1024) return PMCTF_EINVAL;
    dim3 grid(b.tiles_x * b.tiles_y, a.N, gz);
    const SumCfg sc = t_sum;
    if (a.act > pm::ACT_LEAKY && !sc.bsum) {    // tanh / sigmoid epilogues live in the generic kernel only (conv_epilogue.h)
        static std::once_flag once_t;
        allow_big_lds(conv_mfma_kernel<MT, NT, TW16>, once_t);
        CONV_LAUNCH((conv_mfma_kernel<MT, NT, TW16>), grid, dim3(256), smem, st, b);
        return pm_launch_status();
    }
    if (sc.bsum) {
        // rule 1 (per-chunk sums): the 7x7 pipelined kernel where it applies (SpyNet), else the single-buffer kernel;
        // the specialised 3x3 / wave-private kernels carry rule 0 only (their layers of the path that need rule 1 have at
        // most 64 couts and are a few per cent of the work)
        if constexpr (((NT == 4 && TW16 == 2) || (NT == 1 && TW16 == 1)) && MT <= 4) {
            if (a.act <= pm::ACT_LEAKY && sc.bchunks == 1 && !sc.bias_first && a.KH == 7 && a.KW == 7 && a.S == 1 && (a.Cin % CB) == 0 &&
                knob("K77") != 0 && (size_t)a.H * a.W * a.Cin * sizeof(float) < (1ull << 32)) {
                static std::once_flag once_7b;
                allow_big_lds(conv7x7s1_pipe_kernel<MT, NT, true>, once_7b);
                CONV_LAUNCH((conv7x7s1_pipe_kernel<MT, NT, true>), grid, dim3(256), 2 * smem, st, b);
                return pm_launch_status();
            }
        }
        const bool plain_blocks = sc.bchunks == 1 && !sc.bias_first && a.act <= pm::ACT_LEAKY && a.KH == 3 && a.KW == 3 &&
                                  knob("K33") != 0 && (size_t)a.H * a.W * a.Cin * sizeof(float) < (1ull << 32);
        if constexpr (NT == 4 && TW16 == 2 && MT == 7) {
            // 7 cout tiles as 4 + 3 in two launches (both accumulator sets in VGPRs, two workgroups per CU, the chunk sums
            // folded with packed adds) instead of one launch with the running sums in AGPRs at one workgroup per CU.
            // (Both halves in ONE kernel, a tile's halves adjacent on one XCD so that the second finds the patch in L2:
            // measured 125 against 138 TFLOP/s — 246 registers and twice the code in one kernel.)
            if (plain_blocks && a.S == 1 && wave_eligible(a) && knob("BSUM_SPLIT") != 0 && a.mtp == 7) {
                const int PH = 3 * a.S + a.KH, PW = 15 * a.S + a.KW;
                const size_t wsmem = (size_t)PH * PW * CP * sizeof(float) * 2 * WAVES;
                static std::once_flag once_k4, once_k3;
                allow_big_lds(conv3x3s1_wave_kernel<4, 2, true, 0>, once_k4);
                allow_big_lds(conv3x3s1_wave_kernel<3, 2, true, 4>, once_k3);
                CONV_LAUNCH((conv3x3s1_wave_kernel<4, 2, true, 0>), grid, dim3(256), wsmem, st, b);
                { const int rc = pm_launch_status(); if (rc != PMCTF_OK) return rc; }
                CONV_LAUNCH((conv3x3s1_wave_kernel<3, 2, true, 4>), grid, dim3(256), wsmem, st, b);
                return pm_launch_status();
            }
        }
        if constexpr (NT == 4 && TW16 == 2 && (MT == 1 || MT == 2 || MT == 4 || MT == 7)) {
            if (plain_blocks && a.S == 1 && wave_eligible(a)) {
                const int PH = 3 * a.S + a.KH, PW = 15 * a.S + a.KW;
                constexpr int NB = MT <= 2 ? 1 : 2;       // narrow layers: single patch buffer, three workgroups per CU
                const size_t wsmem = (size_t)PH * PW * CP * sizeof(float) * NB * WAVES;
                static std::once_flag once_kb;
                allow_big_lds(conv3x3s1_wave_kernel<MT, NB, true>, once_kb);
                CONV_LAUNCH((conv3x3s1_wave_kernel<MT, NB, true>), grid, dim3(256), wsmem, st, b);
                return pm_launch_status();
            }
        }
        if constexpr (NT == 1 && TW16 == 1 && (MT == 1 || MT == 7 || MT == 8)) {
            // 4x16 tiles: the pipelined 3x3 kernels (mid-size planes, remainder rows, the stride-2 quarter-resolution form)
            const int slots = (LH * LW * 4 + 255) / 256;
            if (plain_blocks && (a.S == 1 || a.S == 2) && (a.Cin % CB) == 0 && 2 * smem <= 80 * 1024 && slots <= 9) {
                if (a.S == 1) CONV_LAUNCH((conv3x3s1_pipe_kernel<MT, 1, true>), grid, dim3(256), 2 * smem, st, b);
                else CONV_LAUNCH((conv3x3s1_pipe_kernel<MT, 2, true>), grid, dim3(256), 2 * smem, st, b);
                return pm_launch_status();
            }
        }
        if constexpr (MT <= 4 || NT <= 2) {
            static std::once_flag once_sb;
            allow_big_lds(conv_mfma_bsum_kernel<MT, NT, TW16>, once_sb);
            CONV_LAUNCH((conv_mfma_bsum_kernel<MT, NT, TW16>), grid, dim3(256), smem, st, b, sc);
            return pm_launch_status();
        } else {
            return launch<MT, 2, 1>(a, gz, st, r0, r1);     // wide M-blocks: 2 segments per wave keep both accumulator sets in registers
        }
    }
    if constexpr (NT == 4 && TW16 == 2) {   // barrier-free wave-private variant (8x32 workgroup tile = 2x2 wave tiles of 4x16)
        if ((MT >= 4 || ((MT == 2 || (MT == 1 && knob("WAVE_SMALL") != 0)) && a.KH == 3 && a.KW == 3 && a.S == 1)) && wave_eligible(a)) {
            const int PH = 3 * a.S + a.KH, PW = 15 * a.S + a.KW;
            const size_t wsmem = (size_t)PH * PW * CP * sizeof(float) * 2 * WAVES;
            const bool k33 = a.KH == 3 && a.KW == 3 && a.S == 1 && knob("K33") != 0 &&
                             (size_t)a.H * a.W * a.Cin * sizeof(float) < (1ull << 32);
            if constexpr (MT <= 4) {
                if (knob("NBUF1") != 0) {       // narrow layers: single patch buffer, three workgroups per CU
                    if (k33) CONV_LAUNCH((conv3x3s1_wave_kernel<MT, 1>), grid, dim3(256), wsmem / 2, st, b);
                    else CONV_LAUNCH((conv_mfma_wave_kernel<MT, 7, 1>), grid, dim3(256), wsmem / 2, st, b);
                    return pm_launch_status();
                }
            }
            if (k33) {
                static std::once_flag once_k;
                allow_big_lds(conv3x3s1_wave_kernel<MT, 2>, once_k);
                CONV_LAUNCH((conv3x3s1_wave_kernel<MT, 2>), grid, dim3(256), wsmem, st, b);
                return pm_launch_status();
            }
            static std::once_flag once_w;
            allow_big_lds(conv_mfma_wave_kernel<MT, 7>, once_w);
            CONV_LAUNCH((conv_mfma_wave_kernel<MT, 7>), grid, dim3(256), wsmem, st, b);
            return pm_launch_status();
        }
    }
    if constexpr (((NT == 4 && TW16 == 2) || (NT == 1 && TW16 == 1)) && MT <= 4) {
        if (a.KH == 7 && a.KW == 7 && a.S == 1 && (a.Cin % CB) == 0 && knob("K77") != 0 &&
            (size_t)a.H * a.W * a.Cin * sizeof(float) < (1ull << 32)) {
            static std::once_flag once_7;
            allow_big_lds(conv7x7s1_pipe_kernel<MT, NT>, once_7);
            CONV_LAUNCH((conv7x7s1_pipe_kernel<MT, NT>), grid, dim3(256), 2 * smem, st, b);
            return pm_launch_status();
        }
    }
    {   // pipelined variant: double-buffered patch (2*smem <= 80 KB so two workgroups fit a CU), <= 9 slots/thread
        const bool v1_only = knob("V1") != 0, v2_only = knob("V2") != 0;
        const int slots = (LH * LW * 4 + 255) / 256;
        // measured (tools/bench_conv.py): the pipelined variant wins for the wide-cout tiles (MT>=7) and for 1x1
        // filters; the single-buffer variant keeps 3 waves/SIMD for MT<=4 and wins on 3x3/7x7 there.
        const bool prefer_v2 = v2_only || MT >= 7 || (a.KH == 1 && a.KW == 1) ||
                               (MT == 1 && NT == 1 && a.KH == 3 && a.KW == 3 && knob("K33_SMALL") != 0);
        if (!v1_only && prefer_v2 && (a.Cin % CB) == 0 && 2 * smem <= 80 * 1024 && slots <= 9) {
            if constexpr (NT == 1 && TW16 == 1) {
                if (a.KH == 3 && a.KW == 3 && a.S == 1 && knob("K33") != 0 &&
                    (size_t)a.H * a.W * a.Cin * sizeof(float) < (1ull << 32)) {
                    CONV_LAUNCH((conv3x3s1_pipe_kernel<MT>), grid, dim3(256), 2 * smem, st, b);
                    return pm_launch_status();
                }
                if (a.KH == 3 && a.KW == 3 && a.S == 2 && knob("K33") != 0 &&
                    (size_t)a.H * a.W * a.Cin * sizeof(float) < (1ull << 32)) {
                    CONV_LAUNCH((conv3x3s1_pipe_kernel<MT, 2>), grid, dim3(256), 2 * smem, st, b);
                    return pm_launch_status();
                }
            }
            static std::once_flag once_p6, once_p9;
            if (slots <= 6) {
                allow_big_lds(conv_mfma_pipe_kernel<MT, NT, TW16, 6>, once_p6);
                CONV_LAUNCH((conv_mfma_pipe_kernel<MT, NT, TW16, 6>), grid, dim3(256), 2 * smem, st, b);
            } else {
                allow_big_lds(conv_mfma_pipe_kernel<MT, NT, TW16, 9>, once_p9);
                CONV_LAUNCH((conv_mfma_pipe_kernel<MT, NT, TW16, 9>), grid, dim3(256), 2 * smem, st, b);
            }
            return pm_launch_status();
        }
    }
    static std::once_flag once_s;
    allow_big_lds(conv_mfma_kernel<MT, NT, TW16>, once_s);
    CONV_LAUNCH((conv_mfma_kernel<MT, NT, TW16>), grid, dim3(256), smem, st, b);
    return pm_launch_status();
}

// resident-patch launch (4x16 tiles, MT cout tiles per workgroup); returns PMCTF_EINVAL when the patch does not fit
template <int MT>
int launch_res(const ConvArgs &a, int gz, hipStream_t st) {
    ConvArgs b = a;
    b.oy_base = 0;
    b.oy_end = a.Ho;
    b.tiles_x = (a.Wo + 15) / 16;
    b.tiles_y = (a.Ho + 3) / 4;
    const int LH = 3 * a.S + a.KH, LW = 15 * a.S + a.KW;
    const size_t smem = (size_t)LH * LW * (a.Cin + 2) * sizeof(float);
    if ((a.Cin % CB) != 0 || smem > 52 * 1024) return PMCTF_EINVAL;
    dim3 grid(b.tiles_x * b.tiles_y, a.N, gz);
    note_launch("conv_mfma_res_kernel<MT>", MT, 1, 1, grid);
    PM_LAUNCH((conv_mfma_res_kernel<MT>), grid, dim3(256), smem, st, b);
    return pm_launch_status();
}

// Launch-shape options of ONE call, resolved: a caller's pmctf_conv_launch_opts field >= 0, else the process-wide knob.
struct LaunchOpts { long split, msplit_px; };
inline LaunchOpts resolve_opts(const pmctf_conv_launch_opts *o) {
    LaunchOpts r;
    r.split = (o && o->split >= 0) ? o->split : knob("SPLIT");
    r.msplit_px = (o && o->msplit_px >= 0) ? o->msplit_px : knob("MSPLIT_PX");
    return r;
}

// MTP = cout tiles per packed M-block (fixed by the weight layout); chooses the tile shape and how the launch is cut.
template <int MTP>
int dispatch_tile(ConvArgs a, int MB, hipStream_t st, const LaunchOpts &lo) {
    a.mtp = MTP;
    const long px = (long)a.Ho * a.Wo * a.N;
    const long force_nt = knob("NT");
    if (force_nt == 4 && MTP < 8 && a.S == 1) return launch<MTP, 4, 2>(a, MB, st, 0, a.Ho);
    if (force_nt == 2) return launch<MTP, 2, 1>(a, MB, st, 0, a.Ho);
    if (force_nt == 1) return launch<MTP, 1, 1>(a, MB, st, 0, a.Ho);
    // (0) 16 -> 16 channels 3x3 'same' (PredictUpdate trunk): persistent kernel with register-resident weights
    if constexpr (MTP == 1) {
        if (knob("C16") != 0 && a.Cin == 16 && a.Cout == 16 && a.KH == 3 && a.KW == 3 && a.S == 1 && a.pad_h == 1 &&
            a.pad_w == 1 && a.Ho == a.H && a.Wo == a.W && px >= 16384) {
            ConvArgs b = a;
            b.tiles_x = (a.Wo + 15) / 16;
            b.tiles_y = (a.Ho + 3) / 4;
            const long tiles = (long)b.tiles_x * b.tiles_y * a.N;
            if (tiles < (1L << 30)) {
                long wgs = (tiles + WAVES - 1) / WAVES;
                const long cap = knob("C16_WGS");
                if (wgs > cap) wgs = cap;
                const long occ = knob("C16_OCC");
                if (occ >= 2) {
                    long g = ((tiles + WAVES - 1) / WAVES + 7) & ~7L;
                    if (g > 256 * occ) g = 256 * occ;
                    const size_t smem1 = (size_t)6 * 18 * CP * sizeof(float) * WAVES;
                    note_launch("conv16_band_kernel", (int)occ, 1, 1, dim3((unsigned)g));
                    if (occ == 2) { PM_LAUNCH(conv16_band_kernel<2>, dim3((unsigned)g), dim3(256), smem1, st, b, (int)tiles, t_sum.bsum); }
                    else if (occ == 3) { PM_LAUNCH(conv16_band_kernel<3>, dim3((unsigned)g), dim3(256), smem1, st, b, (int)tiles, t_sum.bsum); }
                    else { PM_LAUNCH(conv16_band_kernel<4>, dim3((unsigned)g), dim3(256), smem1, st, b, (int)tiles, t_sum.bsum); }
                    return pm_launch_status();
                }
                const size_t smem = (size_t)6 * 18 * CP * sizeof(float) * 2 * WAVES;
                note_launch("conv16_persistent_kernel", 1, 1, 1, dim3((unsigned)wgs));
                PM_LAUNCH(conv16_persistent_kernel, dim3((unsigned)wgs), dim3(256), smem, st, b, (int)tiles, t_sum.bsum);
                return pm_launch_status();
            }
        }
    }
    // (1) small planes: too few 16-pixel segments to occupy 1024 SIMDs with whole M-blocks -> one cout tile per
    //     workgroup, MTP x more (and MTP x shorter) workgroups.  Same sums, same order.
    const long msplit_px = lo.msplit_px;
    // stride-2 3x3 with 112 couts (the quarter-resolution context convolutions on the small DWT levels): the specialised
    // pipelined kernel with all cout tiles per workgroup beats the cout-split generic one from 8 000 output pixels up
    // (2x288x480 -> 2x144x240: 324 -> 177 us; tools/bench_conv.py "s2small")
    const bool s2_whole = MTP >= 7 && a.S == 2 && a.KH == 3 && a.KW == 3 && (a.Cin % CB) == 0 && knob("K33") != 0 && px >= 8000;
    if (MTP >= 2 && px <= msplit_px && !s2_whole) {
        if (knob("RES") == 1 && !t_sum.bsum && a.act <= pm::ACT_LEAKY) {
            const int rc = launch_res<1>(a, MTP * MB, st);
            if (rc != PMCTF_EINVAL) return rc;
        }
        if (knob("RES") == 2 && !t_sum.bsum && a.act <= pm::ACT_LEAKY) {
            const int rc = launch_res<MTP>(a, MB, st);
            if (rc != PMCTF_EINVAL) return rc;
        }
        const long mnt = knob("MSPLIT_NT");
        // 3x3 stride 1 whose planes fill the waves' 4x16 tiles to >= 90 %: the barrier-free wave-private kernel with one cout tile
        // per workgroup (a wave owns a 4x16 tile and its own patch; 4x fewer, 4x longer wave tasks than the 4x16
        // workgroup tiles below and no barriers).  tools/bench_conv.py: 144x240 79 -> 103 TFLOP/s, 8x72x120 81 -> 107,
        // 2x72x120 67 -> 87; planes that fit the tiles badly (36x60: 18 % padding) stay on the fine tiles.
        if (knob("WAVE_SMALL") != 0 && mnt == 1 && a.KH == 3 && a.KW == 3 && a.S == 1 && wave_eligible(a)) {
            const long hw = (long)a.Ho * a.Wo;
            const long padded = (long)((a.Ho + 3) / 4) * 4 * ((a.Wo + 15) / 16) * 16;     // waves without a tile exit at once
            const long padded_wg = (long)((a.Ho + 7) / 8) * 8 * ((a.Wo + 31) / 32) * 32;
            const long tasks = (long)a.N * (padded / 64) * MTP * MB;                       // wave tasks for 1024 SIMDs
            if (padded * 10 <= hw * 11 && (padded_wg * 10 <= hw * 11 || tasks >= 3500))
                return launch<1, 4, 2>(a, MTP * MB, st, 0, a.Ho);
        }
        if (mnt == 2) return launch<1, 2, 1>(a, MTP * MB, st, 0, a.Ho);
        if (mnt == 4 && a.S == 1) return launch<1, 4, 2>(a, MTP * MB, st, 0, a.Ho);
        return launch<1, 1, 1>(a, MTP * MB, st, 0, a.Ho);
    }
    // (2) large planes, stride 1: 8x32 tiles.  The wave-private kernel holds 2 workgroups per CU = 512 slots; rows
    //     that fill whole rounds of 512 go to it, the remaining rows (a partial round) are cut 4x finer (4x16 tiles)
    //     so the tail of the launch costs a quarter of a round instead of a full one.
    const bool split = lo.split != 0;
    // without the cut (launch plans: another stream fills the tail of a launch) the 8x32-tile kernel pays from 200 000
    // pixels on (2x288x480), measured on the harness loop; 1x288x480 (1.05 rounds of 512 workgroups) stays on 4x16 tiles
    const long big_px = split ? knob("BIGPX") : knob("BIGPX_NOSPLIT");
    if (MTP < 8 && a.S == 1 && a.KH <= 7 && px >= big_px) {
        if (split && MTP >= 7 && wave_eligible(a)) {
            // a CU works through its workgroups two at a time: the launch ends when the CU with the most workgroups
            // does, so only cut when the last partial round would idle more than ~3 % of the machine
            const long per_band = (long)a.N * ((a.Wo + 31) / 32) * MB, bands = (a.Ho + 7) / 8, slots = 512;
            const long total = bands * per_band, rounds = total / slots;
            const long padded = (total + 255) / 256 * 256;
            const long bands_a = rounds * slots / per_band;
            if (rounds >= 1 && bands_a >= 1 && bands_a < bands && (padded - total) * 100 > 3 * total) {
                const int rows_a = (int)bands_a * 8;
                const int rc = launch<MTP, 4, 2>(a, MB, st, 0, rows_a);
                if (rc != PMCTF_OK) return rc;
                return launch<MTP, 1, 1>(a, MB, st, rows_a, a.Ho);
            }
            if (rounds < 1 || bands_a < 1) return launch<MTP, 1, 1>(a, MB, st, 0, a.Ho);
        }
        if (px >= 400000L || (MTP >= 4 && wave_eligible(a))) return launch<MTP, 4, 2>(a, MB, st, 0, a.Ho);
        if (MTP == 1 && a.KH == 7 && a.KW == 7 && (a.Cin % CB) == 0 && knob("K77") != 0)
            return launch<MTP, 4, 2>(a, MB, st, 0, a.Ho);      // 32->16 on the 288x480 level: 121 -> 89 us
    }
    // measured on MI355X (tools/bench_conv.py): the wide-cout kernels (MT>=7) prefer 4x16 tiles on everything
    // smaller (more workgroups -> less tail quantisation).
    if (MTP >= 7) return launch<MTP, 1, 1>(a, MB, st, 0, a.Ho);
    if (MTP <= 4 && px < 131072 && a.KH == 7 && a.KW == 7 && a.S == 1 && (a.Cin % CB) == 0 && knob("K77") != 0)
        return launch<MTP, 1, 1>(a, MB, st, 0, a.Ho);          // small SpyNet levels: 4x16 tiles of the 7x7 kernel
    if (px >= 64L * 64 * 4) return launch<MTP, 2, 1>(a, MB, st, 0, a.Ho);
    return launch<MTP, 1, 1>(a, MB, st, 0, a.Ho);
}

// 1x1 GEMM launch: 64 pixels per workgroup on large planes, 32 on small ones; as many cout tiles per wave (<= 4) as
// still leave >= 512 workgroups.
template <int MTW, int NT>
int launch_1x1_t(const ConvArgs &a, long P, int tiles, hipStream_t st) {
    constexpr int PX = 16 * NT;
    const long gx = (P + PX - 1) / PX;
    const int gy = (tiles + WAVES * MTW - 1) / (WAVES * MTW);
    if (gx > 0x7fffffffL) return PMCTF_EINVAL;
    const size_t smem = (size_t)2 * PX * 66 * sizeof(float);
    const SumCfg sc = t_sum;
    if (sc.bsum) {
        note_launch("conv1x1_kernel<MTW, NT, true>", MTW, NT, 1, dim3((unsigned)gx, gy));
        PM_LAUNCH((conv1x1_kernel<MTW, NT, true>), dim3((unsigned)gx, gy), dim3(256), smem, st, a, P, tiles, sc.bchunks);
        return pm_launch_status();
    }
    note_launch("conv1x1_kernel<MTW, NT>", MTW, NT, 1, dim3((unsigned)gx, gy));
    PM_LAUNCH((conv1x1_kernel<MTW, NT>), dim3((unsigned)gx, gy), dim3(256), smem, st, a, P, tiles, 0);
    return pm_launch_status();
}
int launch_1x1(const ConvArgs &a, int tiles, hipStream_t st) {
    const long P = (long)a.N * a.H * a.W;
    const bool big = P >= 32768;
    const long gx = (P + (big ? 63 : 31)) / (big ? 64 : 32);
    int mtw = 1;
    for (int m = tiles >= 16 ? 4 : (tiles + 3) / 4; m >= 1; --m)
        if (gx * ((tiles + WAVES * m - 1) / (WAVES * m)) >= 512 || m == 1) { mtw = m; break; }
    if (big) {
        switch (mtw) {
        case 4: return launch_1x1_t<4, 4>(a, P, tiles, st);
        case 3: return launch_1x1_t<3, 4>(a, P, tiles, st);
        case 2: return launch_1x1_t<2, 4>(a, P, tiles, st);
        default: return launch_1x1_t<1, 4>(a, P, tiles, st);
        }
    }
    switch (mtw) {
    case 4: return launch_1x1_t<4, 2>(a, P, tiles, st);
    case 3: return launch_1x1_t<3, 2>(a, P, tiles, st);
    case 2: return launch_1x1_t<2, 2>(a, P, tiles, st);
    default: return launch_1x1_t<1, 2>(a, P, tiles, st);
    }
}

}  // namespace

extern "C" int pmctf_conv2d_set_option(const char *name, long value) {
    if (!name) return PMCTF_EINVAL;
    for (Knob &k : g_knobs)
        if (!strcmp(k.name, name)) { k.value = value; k.set = true; return PMCTF_OK; }
    return PMCTF_EINVAL;
}

extern "C" long pmctf_conv2d_get_option(const char *name) {
    if (!name) return -1;
    for (Knob &k : g_knobs)
        if (!strcmp(k.name, name)) return knob(name);
    return -1;
}

extern "C" int64_t pmctf_conv2d_packed_bias_size(int Cout) {
    int MT, MB;
    choose_mt(Cout, MT, MB);
    return (int64_t)MT * MB * 16;
}

extern "C" int64_t pmctf_conv2d_packed_size(int Cout, int Cin, int KH, int KW) {
    int MT, MB;
    choose_mt(Cout, MT, MB);
    const int ncb = (Cin + CB - 1) / CB;
    return (int64_t)MB * ncb * KH * KW * MT * 256;
}

// layout: [mb][cb][tap][mt][lane][ks]  value = W[co=(mb*MT+mt)*16+(lane&15)][ci=cb*16+ks*4+(lane>>4)][tap]
extern "C" int pmctf_conv2d_pack_weights(const float *w, const float *bias, int Cout, int Cin, int KH, int KW,
                                         float *wp, float *bp) {
    int MT, MB;
    choose_mt(Cout, MT, MB);
    const int ncb = (Cin + CB - 1) / CB, taps = KH * KW;
    for (int mb = 0; mb < MB; ++mb)
        for (int cb = 0; cb < ncb; ++cb)
            for (int t = 0; t < taps; ++t)
                for (int mt = 0; mt < MT; ++mt)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int ks = 0; ks < 4; ++ks) {
                            const int co = (mb * MT + mt) * 16 + (lane & 15);
                            const int ci = cb * CB + ks * 4 + (lane >> 4);
                            float v = 0.f;
                            if (co < Cout && ci < Cin) v = w[((size_t)co * Cin + ci) * taps + t];
                            wp[(((((size_t)mb * ncb + cb) * taps + t) * MT + mt) * 64 + lane) * 4 + ks] = v;
                        }
    for (int i = 0; i < MT * MB * 16; ++i) bp[i] = (bias && i < Cout) ? bias[i] : 0.f;
    return PMCTF_OK;
}

extern "C" int pmctf_conv2d_nhwc_geom_opts_f32(const float *x, const float *wp, const float *bp, const float *res1,
                                               const float *res2, float *y, int N, int H, int W, int Cin, int Cout,
                                               int KH, int KW, int stride, int pad_top, int pad_left, int Ho, int Wo,
                                               int act, float slope, int sum_rule, const pmctf_conv_launch_opts *opts,
                                               void *stream) {
    if (!x || !wp || !bp || !y || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || (Cin & 3) || Cout <= 0 ||
        KH <= 0 || KW <= 0 || stride <= 0 || pad_top < 0 || pad_left < 0 || Ho <= 0 || Wo <= 0 ||
        (sum_rule != PMCTF_SUM_CHAIN && sum_rule != PMCTF_SUM_BLOCKS && (sum_rule < 16 || (sum_rule & 15))))
        return PMCTF_EINVAL;
    ConvArgs a;
    a.x = x; a.wp = wp; a.bp = bp; a.res1 = res1; a.res2 = res2; a.y = y;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.KH = KH; a.KW = KW; a.S = stride;
    a.pad_h = pad_top; a.pad_w = pad_left;
    a.Ho = Ho;
    a.Wo = Wo;
    a.ncb = (Cin + CB - 1) / CB;
    a.act = act; a.slope = slope;
    a.tiles_x = a.tiles_y = 0;
    a.mtp = 1; a.oy_base = 0; a.oy_end = Ho;
    SumCfg sc = {0, 1, 0};
    if (sum_rule == PMCTF_SUM_BLOCKS) sc.bsum = 1;
    else if (sum_rule >= 16 && sum_rule < Cin) { sc.bsum = 1; sc.bchunks = sum_rule / CB; sc.bias_first = 1; }   // >= Cin: one block = the chain
    t_sum = sc;
    int MT, MB;
    choose_mt(Cout, MT, MB);
    hipStream_t st = (hipStream_t)stream;
    g_last_len = 0;
    g_last_launch[0] = 0;
    if ((!sc.bsum || sc.bias_first) && act <= pm::ACT_LEAKY && KH == 1 && KW == 1 && stride == 1 && pad_top == 0 && pad_left == 0 && Ho == H && Wo == W && (Cin % CB) == 0 &&
        knob("K11") != 0 && MT * MB >= (sc.bsum ? 4 : knob("K11_MIN_TILES"))) {      // waves split the cout tiles: needs >= 2 tiles
        // per wave to pay against the pipelined tap kernels; a blocked reduction has only the generic kernel as the alternative
        // (256 -> 64 on 576x960 with blocks of 96 channels: 33 -> 68 TFLOP/s)
        a.mtp = MT;
        return launch_1x1(a, MT * MB, st);
    }
    const LaunchOpts lo = resolve_opts(opts);
    switch (MT) {
    case 1: return dispatch_tile<1>(a, MB, st, lo);
    case 2: return dispatch_tile<2>(a, MB, st, lo);
    case 4: return dispatch_tile<4>(a, MB, st, lo);
    case 7: return dispatch_tile<7>(a, MB, st, lo);
    default: return dispatch_tile<8>(a, MB, st, lo);
    }
}

extern "C" int pmctf_conv2d_nhwc_geom_f32(const float *x, const float *wp, const float *bp, const float *res1,
                                          const float *res2, float *y, int N, int H, int W, int Cin, int Cout,
                                          int KH, int KW, int stride, int pad_top, int pad_left, int Ho, int Wo,
                                          int act, float slope, void *stream) {
    return pmctf_conv2d_nhwc_geom_opts_f32(x, wp, bp, res1, res2, y, N, H, W, Cin, Cout, KH, KW, stride, pad_top, pad_left,
                                           Ho, Wo, act, slope, PMCTF_SUM_CHAIN, nullptr, stream);
}

extern "C" int pmctf_conv2d_last_launch(char *buf, int capacity) {
    if (!buf || capacity <= 0) return PMCTF_EINVAL;
    snprintf(buf, (size_t)capacity, "%s", g_last_launch);
    return PMCTF_OK;
}

extern "C" int pmctf_conv2d_nhwc_f32(const float *x, const float *wp, const float *bp, const float *res1,
                                     const float *res2, float *y, int N, int H, int W, int Cin, int Cout,
                                     int KH, int KW, int stride, int pad_h, int pad_w, int act, float slope,
                                     void *stream) {
    if (stride <= 0 || KH <= 0 || KW <= 0) return PMCTF_EINVAL;
    const int Ho = (H + 2 * pad_h - KH) / stride + 1;
    const int Wo = (W + 2 * pad_w - KW) / stride + 1;
    return pmctf_conv2d_nhwc_geom_opts_f32(x, wp, bp, res1, res2, y, N, H, W, Cin, Cout, KH, KW, stride, pad_h, pad_w, Ho,
                                           Wo, act, slope, PMCTF_SUM_CHAIN, nullptr, stream);
}

extern "C" int pmctf_conv2d_nhwc_opts_f32(const float *x, const float *wp, const float *bp, const float *res1,
                                          const float *res2, float *y, int N, int H, int W, int Cin, int Cout,
                                          int KH, int KW, int stride, int pad_h, int pad_w, int act, float slope,
                                          int sum_rule, const pmctf_conv_launch_opts *opts, void *stream) {
    if (stride <= 0 || KH <= 0 || KW <= 0) return PMCTF_EINVAL;
    const int Ho = (H + 2 * pad_h - KH) / stride + 1;
    const int Wo = (W + 2 * pad_w - KW) / stride + 1;
    return pmctf_conv2d_nhwc_geom_opts_f32(x, wp, bp, res1, res2, y, N, H, W, Cin, Cout, KH, KW, stride, pad_h, pad_w, Ho,
                                           Wo, act, slope, sum_rule, opts, stream);
}
