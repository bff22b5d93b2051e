// conv_split.hip — REDUCED-PRECISION profile of the dense 3x3 convolutions (auxiliary; never the parity path).
//
// SURVEY.md §7 step 5 plans two conv variants: exact f32 on v_mfma_f32_16x16x4_f32 (conv_mfma.hip, the product's
// arithmetic, bit-exact against the oracle) and a bf16-input / f32-accumulate one for throughput.  This is the second:
// both operands are split into NS bf16 planes (x = x0 + x1 + x2, each the round-to-nearest bf16 of what the previous
// planes leave) and the product is evaluated as the NS(NS+1)/2 largest partial products on v_mfma_f32_16x16x32_bf16
//   NS = 3: a2b0 + a0b2 + a1b1 + a1b0 + a0b1 + a0b0   (~2^-22 relative per product; 6 MFMAs, 2.67x the f32 MFMA rate)
//   NS = 2: a1b0 + a0b1 + a0b0                         (~2^-16;                      3 MFMAs, 5.3x)
//   NS = 1: a0b0                                       (~2^-8, plain bf16;           1 MFMA,  16x)
// accumulated in f32.  Results are deterministic (encoder and decoder of this build agree bit for bit) but differ from
// PM-F32 in the last bits, so a build that selects this profile reports its own statistics and earns no parity claim.
//
// Two kernels, same arithmetic (chosen per shape, dispatch_ns): the barrier-free wave-private form further down, and the
// workgroup form described next.  What shapes both is the WEIGHT-fragment bandwidth: per MFMA cycle the split kernel consumes 4x the A bytes of the
// f32 kernel, more than a CU's L1 delivers if every wave fetched its own.  So a 512-thread workgroup (8 waves, 16x32
// output pixels, all cout tiles of one M-block) stages the fragments of one K block ONCE in LDS (double-buffered, one
// barrier per stage) and its eight waves share them; activations stay wave-private (a 6x18-pixel patch per 16-channel
// chunk, split into bf16 planes while it is staged, no barrier).  K block = 32 = two filter taps x 16 channels (nine taps
// -> five blocks per chunk, the tenth half-block has zero weights).  The patch is stored as NS planes x two half-channel
// regions of 16-byte units whose bases differ by multiples of 256 B: the 16 lanes the hardware groups in a ds_read_b128
// (lane groups mix the two channel halves) then hit 16 different 16-byte bank groups.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include "pm_device_math.h"
#include "conv_epilogue.h"
#include "launch.h"
#include "../../include/pmctf_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int THREADS = 512, NWAVES = 8;
constexpr int PH = 6, PW = 18, NPIX = PH * PW;            // wave patch: 4x16 outputs + 3x3 halo
constexpr int REGION = 1792;                              // NPIX * 16 B rounded up to a multiple of 256 B
constexpr int KB_PER_CHUNK = 5;                           // ceil(9 taps / 2)
// wave patch of the barrier-free kernel at stride S: 4x16 outputs need (3S+3) x (15S+3) input pixels
constexpr int patch_h(int S) { return 3 * S + 3; }
constexpr int patch_w(int S) { return 15 * S + 3; }
constexpr int region_bytes(int S) { return (patch_h(S) * patch_w(S) * 16 + 255) / 256 * 256; }
static_assert(patch_h(1) == PH && patch_w(1) == PW && region_bytes(1) == REGION, "stride-1 patch");

struct SplitArgs {
    const float *x, *bp, *res1, *res2;
    const uint16_t *wp;
    float *y;
    int N, H, W, Cin, Cout, tiles_x, tiles_y, ncb, act;
    float slope;
    int Ho, Wo, pad_h, pad_w;            // output size and top/left padding (stride 1: H, W, 1, 1)
};

// partial products kept for NS planes, smallest first: (plane of A, plane of B)
__host__ __device__ constexpr int n_terms(int ns) { return ns * (ns + 1) / 2; }
__host__ __device__ constexpr int term_a(int ns, int t) {
    return ns == 3 ? (t == 0 ? 2 : (t == 2 || t == 3) ? 1 : 0) : ns == 2 ? (t == 0 ? 1 : 0) : 0;
}
__host__ __device__ constexpr int term_b(int ns, int t) {
    return ns == 3 ? (t == 1 ? 2 : (t == 2 || t == 4) ? 1 : 0) : ns == 2 ? (t == 1 ? 1 : 0) : 0;
}

template <int NS>
__device__ __forceinline__ void split4(const f32x4 v, bf16x4 out[NS]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float r = v[i];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const __bf16 h = (__bf16)r;                 // v_cvt_pk_bf16_f32: round to nearest even
            out[s][i] = h;
            r = r - (float)h;                           // exact in f32
        }
    }
}

// MT cout tiles, NS bf16 planes per operand, KBPS K blocks of weight fragments per LDS stage
template <int MT, int NS, int KBPS>
__global__ __launch_bounds__(THREADS, 2) void conv3x3_split_kernel(SplitArgs a) {
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    constexpr int KBLOCK_BYTES = MT * NS * 1024;              // one K block of fragments: [mt][plane][lane][8 bf16]
    constexpr int ASTAGE = KBPS * KBLOCK_BYTES;
    constexpr int APF = (ASTAGE / 16 + THREADS - 1) / THREADS;  // 16-byte pieces per thread and stage
    constexpr int PATCH = NS * 2 * REGION;
    unsigned char *ldsA = smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char *patch = smem + 2 * ASTAGE + wave * PATCH;
    const int p = lane & 15, g = lane >> 4;
    const int n = blockIdx.y, mb = blockIdx.z;
    const int ty = blockIdx.x / a.tiles_x, tx = blockIdx.x - ty * a.tiles_x;
    const int oy0 = ty * 16 + (wave >> 1) * 4, ox0 = tx * 32 + (wave & 1) * 16;
    const int iy0 = oy0 - 1, ix0 = ox0 - 1;

    f32x4 acc[MT][4];
    {
        const float *bp = a.bp + (size_t)mb * MT * 16 + 4 * g;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x4 b = *(const f32x4 *)(bp + mt * 16);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = b;
        }
    }

    // ---- wave-private activation patch: fetch (global -> registers) and stash (split -> LDS planes)
    constexpr int E = NPIX * 4, MAXP = (E + 63) / 64;
    f32x4 pre[MAXP];
    auto fetch = [&](int cb) {
#pragma unroll
        for (int j = 0; j < MAXP; ++j) {
            const int e = lane + 64 * j;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (e < E) {
                const int pix = e >> 2, part = e & 3;
                const int ly = pix / PW, lx = pix - ly * PW;
                const int gy = iy0 + ly, gx = ix0 + lx;
                if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                    v = *(const f32x4 *)(a.x + (((size_t)n * a.H + gy) * a.W + gx) * a.Cin + cb * 16 + part * 4);
            }
            pre[j] = v;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int j = 0; j < MAXP; ++j) {
            const int e = lane + 64 * j;
            if (e < E) {
                const int pix = e >> 2, part = e & 3;
                bf16x4 pl[NS];
                split4<NS>(pre[j], pl);
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    *(bf16x4 *)(patch + (s * 2 + (part >> 1)) * REGION + pix * 16 + (part & 1) * 8) = pl[s];
            }
        }
    };
    // ---- workgroup-shared weight fragments: stage st = K blocks [st*KBPS, (st+1)*KBPS) of this M-block
    const int nkb = a.ncb * KB_PER_CHUNK, nstages = nkb / KBPS;
    const unsigned char *wsrc = (const unsigned char *)a.wp + (size_t)mb * nkb * KBLOCK_BYTES;
    uint4 apre[APF];
    auto a_fetch = [&](int st) {
        const uint4 *src = (const uint4 *)(wsrc + (size_t)st * ASTAGE);
#pragma unroll
        for (int j = 0; j < APF; ++j) {
            const int c = tid + THREADS * j;
            if (c < ASTAGE / 16) apre[j] = src[c];
        }
    };
    auto a_stash = [&](int buf) {
        uint4 *dst = (uint4 *)(ldsA + buf * ASTAGE);
#pragma unroll
        for (int j = 0; j < APF; ++j) {
            const int c = tid + THREADS * j;
            if (c < ASTAGE / 16) dst[c] = apre[j];
        }
    };

    a_fetch(0);
    fetch(0);
    a_stash(0);
    stash();
    __syncthreads();
    int buf = 0;
    for (int st = 0; st < nstages; ++st) {
        if (st + 1 < nstages) a_fetch(st + 1);
#pragma unroll 1
        for (int kk = 0; kk < KBPS; ++kk) {
            const int kbi = st * KBPS + kk;
            const int cb = kbi / KB_PER_CHUNK, kb = kbi - cb * KB_PER_CHUNK;
            if (kb == 0 && cb + 1 < a.ncb) fetch(cb + 1);
            // this lane's tap of the K block (lanes 0-31: tap 2kb, lanes 32-63: tap 2kb+1; the tenth half-block
            // re-reads tap 8 against zero weights) and channel half
            int tap = 2 * kb + (g >> 1);
            tap = tap > 8 ? 8 : tap;
            const int ky = tap / 3, kx = tap - ky * 3;
            const unsigned char *bsrc = patch + (g & 1) * REGION + ((ky * PW) + p + kx) * 16;
            const unsigned char *asrc = ldsA + buf * ASTAGE + kk * KBLOCK_BYTES + lane * 16;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                bf16x8 b[2][NS];
#pragma unroll
                for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
                    for (int s = 0; s < NS; ++s)
                        b[n2][s] = *(const bf16x8 *)(bsrc + s * 2 * REGION + (half * 2 + n2) * PW * 16);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    bf16x8 af[NS];
#pragma unroll
                    for (int s = 0; s < NS; ++s) af[s] = *(const bf16x8 *)(asrc + (mt * NS + s) * 1024);
#pragma unroll
                    for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
                        for (int t = 0; t < n_terms(NS); ++t)
                            acc[mt][half * 2 + n2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                af[term_a(NS, t)], b[n2][term_b(NS, t)], acc[mt][half * 2 + n2], 0, 0, 0);
                }
            }
            if (kb == KB_PER_CHUNK - 1 && cb + 1 < a.ncb) stash();      // in-order LDS: after this wave's last read
        }
        if (st + 1 < nstages) a_stash(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    // ---- epilogue (as conv_mfma.hip): lane holds couts 4g..4g+3 of pixel p of each of its four rows
    PM_EPILOGUE_NOTRANS(a,
_Pragma("unroll")
    for (int nt = 0; nt < 4; ++nt) {
        const int oy = oy0 + nt, ox = ox0 + p;
        if (oy >= a.Ho || ox >= a.Wo) continue;
        const size_t pbase = (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.Cout;
_Pragma("unroll")
        for (int mt = 0; mt < MT; ++mt) {
            const int co = (mb * MT + mt) * 16 + 4 * g;
            if (co >= a.Cout) continue;
            store_frag<ACT, RES>(a, acc[mt][nt], pbase, co);
        }
    })
}

// ---------------------------------------------------------------------------------------------------
// Barrier-free variant: the structure of conv_mfma_wave_kernel (every wave owns a 4x16-pixel tile and a private patch,
// workgroups only group waves for dispatch) with the split bf16 products.  Weight fragments come straight from L1/L2 into
// registers, one cout tile ahead of their use: with two waves per SIMD a K block lasts twice its MFMA time, which halves
// the fragment bandwidth a CU needs (about 31 B/clk for three planes, within L1's 64) — and without barriers the waves
// drift apart, so one wave's patch staging (fetch, split, LDS writes) hides behind its SIMD partner's MFMAs.
template <int MT, int NS, int OCC = 2, int S = 1>
__global__ __launch_bounds__(256, OCC) void conv3x3_split_wave_kernel(SplitArgs a) {
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    // S = 2: the quarter-resolution context convolutions (stride 2, top/left padding 1 - parity: ops.conv_at_class)
    constexpr int PH = patch_h(S), PW = patch_w(S), NPIX = PH * PW, REGION = region_bytes(S);
    constexpr int PATCH = NS * 2 * REGION;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char *patch = smem + wave * PATCH;
    const int p = lane & 15, g = lane >> 4;
    const int n = blockIdx.y, mb = blockIdx.z;
    const int ty = blockIdx.x / a.tiles_x, tx = blockIdx.x - ty * a.tiles_x;
    const int oy0 = ty * 8 + (wave >> 1) * 4, ox0 = tx * 32 + (wave & 1) * 16;
    const int iy0 = oy0 * S - a.pad_h, ix0 = ox0 * S - a.pad_w;

    f32x4 acc[MT][4];
    {
        const float *bp = a.bp + (size_t)mb * MT * 16 + 4 * g;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x4 b = *(const f32x4 *)(bp + mt * 16);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = b;
        }
    }
    constexpr int E = NPIX * 4, MAXP = (E + 63) / 64;
    f32x4 pre[MAXP];
    auto fetch = [&](int cb) {
#pragma unroll
        for (int j = 0; j < MAXP; ++j) {
            const int e = lane + 64 * j;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (e < E) {
                const int pix = e >> 2, part = e & 3;
                const int ly = pix / PW, lx = pix - ly * PW;
                const int gy = iy0 + ly, gx = ix0 + lx;
                if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                    v = *(const f32x4 *)(a.x + (((size_t)n * a.H + gy) * a.W + gx) * a.Cin + cb * 16 + part * 4);
            }
            pre[j] = v;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int j = 0; j < MAXP; ++j) {
            const int e = lane + 64 * j;
            if (e < E) {
                const int pix = e >> 2, part = e & 3;
                bf16x4 pl[NS];
                split4<NS>(pre[j], pl);
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    *(bf16x4 *)(patch + (s * 2 + (part >> 1)) * REGION + pix * 16 + (part & 1) * 8) = pl[s];
            }
        }
    };
    const int nkb = a.ncb * KB_PER_CHUNK;
    // fragments of (K block kbi, cout tile mt, plane s): 1 KB each, this lane's 16 bytes
    const unsigned char *wq = (const unsigned char *)a.wp + (size_t)mb * nkb * MT * NS * 1024 + lane * 16;
    // DEEP: the fragments of a whole K block (MT x NS) are requested one K block ahead (28 MFMAs = ~450 cycles of cover
    // instead of one cout tile = 4 MFMAs); costs MT*NS*4 more registers, so only where they are free
    constexpr bool DEEP = (MT * NS <= 8) || OCC == 1;      // one workgroup per CU: 512 registers per wave, always deep
    bf16x8 a_cur[DEEP ? MT : 1][NS], a_nxt[DEEP ? MT : 1][NS];
#pragma unroll
    for (int m = 0; m < (DEEP ? MT : 1); ++m)
#pragma unroll
        for (int s = 0; s < NS; ++s) a_cur[m][s] = *(const bf16x8 *)(wq + (m * NS + s) * 1024);
    fetch(0);
    stash();
    long wstep = 0;                                   // (kbi * MT + mt): fragment group being multiplied
    const long wsteps = (long)nkb * MT;
    for (int cb = 0; cb < a.ncb; ++cb) {
        const bool more = cb + 1 < a.ncb;
        if (more) fetch(cb + 1);
#pragma unroll 1
        for (int kb = 0; kb < KB_PER_CHUNK; ++kb) {
            int tap = 2 * kb + (g >> 1);
            tap = tap > 8 ? 8 : tap;
            const int ky = tap / 3, kx = tap - ky * 3;
            const unsigned char *bsrc = patch + (g & 1) * REGION + ((ky * PW) + p * S + kx) * 16;
            bf16x8 b[4][NS];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int s = 0; s < NS; ++s) b[nt][s] = *(const bf16x8 *)(bsrc + s * 2 * REGION + nt * S * PW * 16);
            if (DEEP) {
                const long kn = (long)(cb * KB_PER_CHUNK + kb + 1) * MT;       // first fragment group of the next K block
                const long k0 = kn < wsteps ? kn : 0;
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int s = 0; s < NS; ++s) a_nxt[DEEP ? m : 0][s] = *(const bf16x8 *)(wq + ((k0 + m) * NS + s) * 1024);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                ++wstep;
                if (!DEEP) {   // next fragment group, always loaded (the last one re-reads the first: no branch)
                    const long wn = wstep < wsteps ? wstep : 0;
#pragma unroll
                    for (int s = 0; s < NS; ++s) a_nxt[0][s] = *(const bf16x8 *)(wq + (wn * NS + s) * 1024);
                }
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int t = 0; t < n_terms(NS); ++t)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_cur[DEEP ? mt : 0][term_a(NS, t)], b[nt][term_b(NS, t)],
                                                                              acc[mt][nt], 0, 0, 0);
                if (!DEEP) {
#pragma unroll
                    for (int s = 0; s < NS; ++s) a_cur[0][s] = a_nxt[0][s];
                }
            }
            if (DEEP) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int s = 0; s < NS; ++s) a_cur[DEEP ? m : 0][s] = a_nxt[DEEP ? m : 0][s];
            }
        }
        if (more) stash();       // in-order LDS: after this wave's last read of the chunk
    }
    PM_EPILOGUE_NOTRANS(a,
_Pragma("unroll")
    for (int nt = 0; nt < 4; ++nt) {
        const int oy = oy0 + nt, ox = ox0 + p;
        if (oy >= a.Ho || ox >= a.Wo) continue;
        const size_t pbase = (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.Cout;
_Pragma("unroll")
        for (int mt = 0; mt < MT; ++mt) {
            const int co = (mb * MT + mt) * 16 + 4 * g;
            if (co >= a.Cout) continue;
            store_frag<ACT, RES>(a, acc[mt][nt], pbase, co);
        }
    })
}

// cout tiles per M-block / number of M-blocks the kernel is instantiated for
inline bool split_shape(int Cout, int &MT, int &MB) {
    if (Cout % 16) return false;
    const int tiles = Cout / 16;
    if (tiles == 7 || tiles == 4) { MT = tiles; MB = 1; return true; }       // 112 / 64 couts (8 tiles spills)
    return false;
}

inline uint16_t bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);      // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline float bf16_to_f32(uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

template <int MT, int NS, int KBPS>
int launch_split(const SplitArgs &a, int MB, hipStream_t st) {
    constexpr size_t smem = 2 * (size_t)KBPS * MT * NS * 1024 + (size_t)NWAVES * NS * 2 * REGION;
    static_assert(smem <= 160 * 1024, "LDS budget");
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute((const void *)conv3x3_split_kernel<MT, NS, KBPS>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    dim3 grid(a.tiles_x * a.tiles_y, a.N, MB);
    PM_LAUNCH((conv3x3_split_kernel<MT, NS, KBPS>), grid, dim3(THREADS), smem, st, a);
    return pm_launch_status();
}

template <int MT, int NS, int OCC = 2, int S = 1>
int launch_split_wave(SplitArgs a, int MB, hipStream_t st) {
    constexpr size_t smem = (size_t)4 * NS * 2 * region_bytes(S);
    static_assert(smem <= 160 * 1024, "LDS budget");
    if (smem > 64 * 1024) {
        static std::once_flag once;
        std::call_once(once, [] {
            (void)hipFuncSetAttribute((const void *)conv3x3_split_wave_kernel<MT, NS, OCC, S>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        });
    }
    a.tiles_x = (a.Wo + 31) / 32;
    a.tiles_y = (a.Ho + 7) / 8;                     // 8x32-pixel workgroups of four wave tiles
    dim3 grid(a.tiles_x * a.tiles_y, a.N, MB);
    PM_LAUNCH((conv3x3_split_wave_kernel<MT, NS, OCC, S>), grid, dim3(256), smem, st, a);
    return pm_launch_status();
}

template <int MT>
int dispatch_ns(const SplitArgs &a, int MB, int nsplit, hipStream_t st) {
    // Measured (tools/bench_split.py, 8x576x960x112 / 8x1152x1920x64, TFLOP/s-equivalent; exact f32 kernel 130 / 120):
    //   barrier-free, fragments from L1/L2:   x3 196 / 184   x2 251 / 294   x1 485 / 493
    //   fragments shared through LDS:         x3 160 / 169   x2 260 / 286   x1 455 / 382
    // (the two-plane product is L1-bound without sharing and barrier-bound with it).  PMCTF_SPLIT_VARIANT forces one.
    static const int forced = [] { const char *v = getenv("PMCTF_SPLIT_VARIANT"); return v ? atoi(v) : -1; }();
    // round 3: for two planes and 112 couts the barrier-free kernel compiled for ONE workgroup per CU budget (accumulators
    // in AGPRs, the fragments of a whole K block requested a block ahead) beats both: 257 -> 341-378
    // (two planes x 112 couts) and 191 -> 204 (three planes x 64 couts); elsewhere it ties or loses (x3 / 112: 188 vs 205)
    const int variant = forced >= 0 ? forced : (((nsplit == 2 && MT == 7) || (nsplit == 3 && MT == 4)) ? 2 : 1);
    if (variant == 2) {         // one workgroup per CU, every fragment a K block ahead (PMCTF_SPLIT_VARIANT=2)
        switch (nsplit) {
        case 3: return launch_split_wave<MT, 3, 1>(a, MB, st);
        case 2: return launch_split_wave<MT, 2, 1>(a, MB, st);
        case 1: return launch_split_wave<MT, 1, 1>(a, MB, st);
        default: return PMCTF_EINVAL;
        }
    }
    if (variant == 1) {
        switch (nsplit) {
        case 3: return launch_split_wave<MT, 3>(a, MB, st);
        case 2: return launch_split_wave<MT, 2>(a, MB, st);
        case 1: return launch_split_wave<MT, 1>(a, MB, st);
        default: return PMCTF_EINVAL;
        }
    }
    switch (nsplit) {
    case 3: return launch_split<MT, 3, 1>(a, MB, st);
    case 2: return launch_split<MT, 2, 1>(a, MB, st);
    case 1: return launch_split<MT, 1, 5>(a, MB, st);
    default: return PMCTF_EINVAL;
    }
}

}  // namespace

extern "C" int pmctf_conv3x3_split_supported(int Cin, int Cout) {
    int MT, MB;
    return (Cin > 0 && Cin % 16 == 0 && split_shape(Cout, MT, MB)) ? 1 : 0;
}

extern "C" int64_t pmctf_conv3x3_split_packed_size(int Cout, int Cin, int nsplit) {
    int MT, MB;
    if (!split_shape(Cout, MT, MB) || Cin % 16 || nsplit < 1 || nsplit > 3) return -1;
    return (int64_t)MB * (Cin / 16) * KB_PER_CHUNK * MT * nsplit * 512;        // uint16 elements
}

// layout: [mb][cb][kb][mt][plane][lane][j]; lane = (g, p): cout (mb*MT+mt)*16 + p, tap 2kb + (g>>1), channel cb*16 + 8(g&1) + j
extern "C" int pmctf_conv3x3_split_pack_weights(const float *w, const float *bias, int Cout, int Cin, int nsplit,
                                                uint16_t *wp, float *bp) {
    int MT, MB;
    if (!w || !wp || !bp || !split_shape(Cout, MT, MB) || Cin % 16 || nsplit < 1 || nsplit > 3) return PMCTF_EINVAL;
    const int ncb = Cin / 16;
    size_t o = 0;
    for (int mb = 0; mb < MB; ++mb)
        for (int cb = 0; cb < ncb; ++cb)
            for (int kb = 0; kb < KB_PER_CHUNK; ++kb)
                for (int mt = 0; mt < MT; ++mt) {
                    for (int s = 0; s < nsplit; ++s)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 8; ++j) {
                                const int p = lane & 15, g = lane >> 4;
                                const int co = (mb * MT + mt) * 16 + p, tap = 2 * kb + (g >> 1);
                                const int ci = cb * 16 + 8 * (g & 1) + j;
                                float r = (tap < 9 && co < Cout) ? w[((size_t)co * Cin + ci) * 9 + tap] : 0.0f;
                                uint16_t h = 0;
                                for (int q = 0; q <= s; ++q) { h = bf16_rne(r); r = r - bf16_to_f32(h); }
                                wp[o + ((size_t)s * 64 + lane) * 8 + j] = h;
                            }
                    o += (size_t)nsplit * 512;
                }
    for (int i = 0; i < MT * MB * 16; ++i) bp[i] = (bias && i < Cout) ? bias[i] : 0.f;
    return PMCTF_OK;
}

extern "C" int pmctf_conv3x3_split_f32(const float *x, const uint16_t *w_packed, const float *bias_packed,
                                       const float *res1, const float *res2, float *y, int N, int H, int W, int Cin,
                                       int Cout, int nsplit, int act, float slope, void *stream) {
    int MT, MB;
    if (!x || !w_packed || !bias_packed || !y || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cin % 16 ||
        !split_shape(Cout, MT, MB) || N > 65535 || act < 0 || act > 2)      // none / relu / leaky (conv_epilogue.h act_c<-2>)
        return PMCTF_EINVAL;
    SplitArgs a;
    a.x = x; a.wp = w_packed; a.bp = bias_packed; a.res1 = res1; a.res2 = res2; a.y = y;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.ncb = Cin / 16; a.act = act; a.slope = slope;
    a.Ho = H; a.Wo = W; a.pad_h = 1; a.pad_w = 1;
    a.tiles_x = (W + 31) / 32;
    a.tiles_y = (H + 15) / 16;
    hipStream_t st = (hipStream_t)stream;
    switch (MT) {
    case 4: return dispatch_ns<4>(a, MB, nsplit, st);
    case 7: return dispatch_ns<7>(a, MB, nsplit, st);
    default: return PMCTF_EINVAL;
    }
}

// Stride-2 form with explicit geometry (the quarter-resolution context convolutions, pmctf_conv2d_nhwc_geom_f32's
// shape: Ho x Wo outputs, output (oy, ox) reads the 3x3 window whose top-left input pixel is (2 oy - pad_h, 2 ox - pad_w),
// zeros outside).  Same packed weights as the stride-1 form; barrier-free kernel, one workgroup per CU.
extern "C" int pmctf_conv3x3_split_geom_f32(const float *x, const uint16_t *w_packed, const float *bias_packed,
                                            const float *res1, const float *res2, float *y, int N, int H, int W, int Cin,
                                            int Cout, int nsplit, int stride, int pad_h, int pad_w, int Ho, int Wo,
                                            int act, float slope, void *stream) {
    int MT, MB;
    if (!x || !w_packed || !bias_packed || !y || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cin % 16 ||
        !split_shape(Cout, MT, MB) || MT != 7 || N > 65535 || stride != 2 || Ho <= 0 || Wo <= 0 || pad_h < 0 ||
        pad_w < 0 || pad_h > 1 || pad_w > 1 || 2 * (Ho - 1) - pad_h >= H || 2 * (Wo - 1) - pad_w >= W || act < 0 || act > 2)
        return PMCTF_EINVAL;
    SplitArgs a;
    a.x = x; a.wp = w_packed; a.bp = bias_packed; a.res1 = res1; a.res2 = res2; a.y = y;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.ncb = Cin / 16; a.act = act; a.slope = slope;
    a.Ho = Ho; a.Wo = Wo; a.pad_h = pad_h; a.pad_w = pad_w;
    a.tiles_x = a.tiles_y = 0;
    hipStream_t st = (hipStream_t)stream;
    switch (nsplit) {
    case 3: return launch_split_wave<7, 3, 1, 2>(a, MB, st);
    case 2: return launch_split_wave<7, 2, 1, 2>(a, MB, st);
    case 1: return launch_split_wave<7, 1, 1, 2>(a, MB, st);
    default: return PMCTF_EINVAL;
    }
}
