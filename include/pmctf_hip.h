/* pmctf_hip.h — C ABI of libpmctf_hip.so: the gfx950 (MI355X) kernels behind the
 * pMCTF temporal-decomposition encode path.
 *
 * Every entry point replaces one torch functional op (or a fixed group of them)
 * that the reference's hot path calls; the reference file:line is given per
 * function.  Signatures use plain pointers and sizes only.  All pointers are
 * DEVICE pointers unless said otherwise; `stream` is a hipStream_t passed as
 * void*.  Nothing allocates or synchronises; launches are asynchronous on
 * `stream`.  Return value: 0 on success, negative on a rejected shape
 * (PMCTF_EINVAL) or a HIP launch error (PMCTF_ELAUNCH).
 *
 * Layout: feature maps are NHWC ("channels last") float32, dense.  Single-channel
 * planes (frames, subbands, flow components) are therefore identical to the
 * reference's NCHW tensors.  Arithmetic follows the PM-F32 spec in DESIGN.md:
 * IEEE binary32, one rounding per stated operation, convolution sums as an
 * ordered fmaf chain (exactly what v_mfma_f32_16x16x4_f32 computes).
 */
#ifndef PMCTF_HIP_H
#define PMCTF_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PMCTF_OK 0
#define PMCTF_EINVAL (-1)
#define PMCTF_ELAUNCH (-2)

/* activation codes for conv epilogues / elementwise maps */
/* Summation rule of a convolution (the argument `sum_rule`): the ORDER in which the products of one output element are
 * added, which fixes its last bits.  Both are k-ordered fmaf chains over 16-channel blocks of the input, (ky, kx, ci)
 * ascending inside a block:
 *   PMCTF_SUM_CHAIN   acc = bias, then one chain through all blocks.  What ATen's CPU path computes for the 1x1 layers of
 *                     the path whose reduction is one block (oneDNN jit_1x1) and for depthwise layers.
 *   PMCTF_SUM_BLOCKS  every block a chain from zero, S_0 .. S_last; result (..((S_0 + bias) + S_1) + ..) + S_last.  What
 *                     ATen's CPU path (oneDNN direct convolution) computes for every KH*KW > 1 layer, measured bit for bit
 *                     (tools/aten_conv_rules.py).  The drop-in path uses it for every KH*KW > 1 layer with at most 16 input
 *                     channels and for the multi-block layers of the signal path (motion estimation, motion codec, temporal
 *                     and spatial lifting): the layers whose last bits decide symbols (profiles/round4_flip_attribution.md).
 *   B >= 16, B % 16 == 0 ("reduce-B", entry points with an MFMA path only): blocks of B input channels; the FIRST block's
 *                     chain starts at the bias, later blocks at zero, block results added in turn (B >= Cin is the chain).
 *                     What ATen's jit_1x1 kernel computes for the 1x1 layers whose reduction it blocks; B follows from the
 *                     layer's shape (pMCTF/hip/aten_rules.py, e.g. 256 -> 64 on a 576x960 plane: B = 96).
 *   PMCTF_SUM_GEMM    (pmctf_conv2d_smallcin_f32 only) ONE chain from zero over (ci, ky, kx), bias added last: ATen's
 *                     im2col + sgemm path, which it takes instead of oneDNN for a single image of at most 20 480 input
 *                     elements and filters up to 3x3 (Convolution.cpp use_mkldnn).  With one input channel it equals BLOCKS.
 */
#define PMCTF_SUM_CHAIN 0
#define PMCTF_SUM_BLOCKS 1
#define PMCTF_SUM_GEMM 2
/* (pmctf_conv2d_smallcin_f32 only; Cin = 1, 3x3) the same path with ONE output channel is a matrix-vector product over the
 * nine im2col columns p_k = x_k * w_k (k = 3 ky + kx); MKL's kernel sums eight at a time in four chains, the ninth after:
 *   E = fma(p4, fma(p6, bias));  O = fma(p5, round(p7));  A = fma(p0, round(p2)) + fma(p1, round(p3));  y = fma(p8, (E + O) + A)
 * (measured with probe inputs: tools/aten_gemv_probe.py; the 1 -> 1 "lower_level_subband" layer on planes <= 20 480 px) */
#define PMCTF_SUM_GEMV_3X3 3

#define PMCTF_ACT_NONE 0
#define PMCTF_ACT_RELU 1    /* nn.ReLU                 video_net.py:77      */
#define PMCTF_ACT_LEAKY 2   /* nn.LeakyReLU(slope)     video/layers.py:57-60, context_fusion_4step.py:13 */
#define PMCTF_ACT_TANH 3    /* torch.tanh              lifting_1d.py:39,42  */
#define PMCTF_ACT_SIGMOID 4 /* torch.sigmoid           long_context.py:24   */

/* Number of floats pmctf_conv2d_pack_weights() writes for a (Cout,Cin,KH,KW) filter. */
int64_t pmctf_conv2d_packed_size(int Cout, int Cin, int KH, int KW);

/* HOST-side re-layout of an OIHW float filter + bias into the MFMA A-fragment
 * order consumed by pmctf_conv2d_nhwc_f32 (both pointers are host pointers).
 * bias_packed receives pmctf_conv2d_packed_bias_size(Cout) floats (zero padded). */
int pmctf_conv2d_pack_weights(const float *w_oihw, const float *bias, int Cout, int Cin, int KH, int KW,
                              float *w_packed, float *bias_packed);
int64_t pmctf_conv2d_packed_bias_size(int Cout);

/* Launch-shape tuning of pmctf_conv2d_nhwc*_f32 (never changes results, only how the work is cut into workgroups).
 * Names (also read once from the environment as PMCTF_CONV_<NAME>): "WAVE" 0/1 wave-private kernel, "NT" force 1/2/4
 * pixel segments per wave, "MSPLIT_PX" planes with at most this many output pixels run one cout tile per workgroup,
 * "SPLIT" 0/1 cut the partial last round of 8x32 tiles into 4x16 tiles, "BIGPX" smallest plane given 8x32 tiles,
 * "V1"/"V2" force the single-buffer / pipelined kernel, "RES" 0/1/2 resident-patch kernel for the
 * cout-split planes (off / one cout tile / all cout tiles per workgroup), "MSPLIT_NT" 1/2/4 tile rows per wave on the
 * cout-split planes, "C16" 0/1 persistent 16->16 kernel with "C16_WGS" workgroups at most, "NBUF1" 0/1 single patch buffer
 * (three workgroups per CU) for the wave-private kernel on layers of at most 64 couts, "K33" 0/1 the 3x3 stride-1
 * specialisations (wave-private and pipelined kernels, stride 1 and 2: no vector-ALU work in the tap loop), "K77" 0/1 the
 * 7x7 stride-1 pipelined kernel (8x32 and 4x16 tiles), "K33_SMALL" 0/1 the specialised 3x3 pipelined kernel with one cout
 * tile per workgroup on the cout-split planes the wave-private kernel does not take, "K11" 0/1 the flat GEMM kernel for 1x1 layers with at least "K11_MIN_TILES" (7) 16-cout tiles, "WAVE_SMALL" 0/1 the
 * wave-private kernel with one cout tile per workgroup on cout-split 3x3 planes that fill the waves' 4x16 tiles to >= 90 %.
 * Environment only: PMCTF_FEWCOUT_LDS=0 / PMCTF_DWCONV_COLUMN=0 select the older one/two-cout and depthwise kernels. */
int pmctf_conv2d_set_option(const char *name, long value);
/* current value of a knob (-1: unknown name).  The launch plans of the drop-in path record their convolutions with
 * "SPLIT" = 0: luma's and chroma's coders run side by side there, the other stream fills the tail of a launch, and
 * cutting a launch into whole rounds + remainder only adds launches (5.41 -> 5.49 frames/s on the 1080p GOP-16 encode);
 * on a single stream the cut pays (the default). */
long pmctf_conv2d_get_option(const char *name);
/* Measurement aid (no reference counterpart): which kernel(s) the LAST pmctf_conv2d_nhwc[_geom]_f32 call of the calling
 * thread launched — kernel expression, tile parameters and grid, several joined by " + " when the launch was cut
 * (whole rounds of workgroups + remainder).  bench.py names the kernel behind its roofline figure from this. */
int pmctf_conv2d_last_launch(char *buf, int capacity);

/* nn.Conv2d forward (groups=1, zero padding), optionally fused with what follows it
 * in the reference: y = act(conv(x) + bias) [+ res1] [+ res2].
 *   reference: F.conv2d behind every nn.Conv2d on the path, e.g.
 *   pMCTF/layers/video/video_net.py:78-90 (SpyNet 7x7), pMCTF/layers/context_fusion_4step.py:12-20
 *   (ContextResidual), pMCTF/layers/postprocessing.py:9-18,35-44, pMCTF/layers/long_context.py:13-14,
 *   pMCTF/layers/video/layers.py:22-136, pMCTF/layers/context_fusion.py:56-128 (masked filters are
 *   multiplied by their mask at pack time, layers.py:49-51).
 * x: [N,H,W,Cin]  y/res1/res2: [N,Ho,Wo,Cout], Ho=(H+2*pad_h-KH)/stride+1.  Cin % 4 == 0.
 * Sum order per output: bias, then 16-channel chunks (ky,kx raster inside a chunk, channel ascending). */
int pmctf_conv2d_nhwc_f32(const float *x, const float *w_packed, const float *bias_packed,
                          const float *res1, const float *res2, float *y,
                          int N, int H, int W, int Cin, int Cout, int KH, int KW,
                          int stride, int pad_h, int pad_w, int act, float slope, void *stream);

/* The same call with the summation rule (above; pmctf_conv2d_nhwc_f32 is PMCTF_SUM_CHAIN) and the launch-shape options of
 * THIS launch handed over as arguments instead of read from the
 * process-wide knobs (a field < 0, or opts == NULL, takes the knob of that name): "split" = "SPLIT", "msplit_px" =
 * "MSPLIT_PX" above.  The captured launch plans of the drop-in path record their convolutions with their own values
 * (two coders side by side fill each other's launch tails), and do so per launch: nothing process-wide changes while a
 * plan is recorded, so a second model on another host thread keeps its own launch shapes.  Results never depend on them. */
typedef struct pmctf_conv_launch_opts {
    long split;      /* 0/1: cut the partial last round of 8x32 tiles into 4x16 tiles; < 0: process-wide knob */
    long msplit_px;  /* planes with at most this many output pixels run one cout tile per workgroup; < 0: knob */
} pmctf_conv_launch_opts;
int pmctf_conv2d_nhwc_opts_f32(const float *x, const float *w_packed, const float *bias_packed,
                               const float *res1, const float *res2, float *y,
                               int N, int H, int W, int Cin, int Cout, int KH, int KW,
                               int stride, int pad_h, int pad_w, int act, float slope, int sum_rule,
                               const pmctf_conv_launch_opts *opts, void *stream);

/* Explicit-geometry form: output size (Ho,Wo) and top/left padding given by the caller; taps falling outside the
 * input read zero.  Used to evaluate a stride-1 convolution only at one 2x2 parity class of positions
 * (stride 2, pad = 1 - parity): the four-step coder needs the last ContextResidual conv and the 1x1 parameter head of
 * each step only where that step's mask is set (context_fusion_4step.py:127-137,160-189), a quarter of the plane. */
int pmctf_conv2d_nhwc_geom_f32(const float *x, const float *w_packed, const float *bias_packed,
                               const float *res1, const float *res2, float *y,
                               int N, int H, int W, int Cin, int Cout, int KH, int KW,
                               int stride, int pad_top, int pad_left, int Ho, int Wo, int act, float slope,
                               void *stream);

int pmctf_conv2d_nhwc_geom_opts_f32(const float *x, const float *w_packed, const float *bias_packed,
                                    const float *res1, const float *res2, float *y,
                                    int N, int H, int W, int Cin, int Cout, int KH, int KW,
                                    int stride, int pad_top, int pad_left, int Ho, int Wo, int act, float slope,
                                    int sum_rule, const pmctf_conv_launch_opts *opts, void *stream);

/* Same contract for Cin <= 4 (any Cin >= 1), plain OIHW weights on the device: direct convolution on the vector ALU
 * (first layers: PredictUpdate conv1 1->16 lifting_1d.py:38; PostProcess conv1 1->64 postprocessing.py:35; four-step
 * y_spatial_prior_k.0 1->112 and conv1_context 1|2->112 context_fusion_4step.py:48,63,73,83; masked 1->128
 * context_fusion.py:79; MV encoder first conv 2->64 video_net.py:128). */
int pmctf_conv2d_smallcin_f32(const float *x, const float *w_oihw, const float *bias,
                              const float *res1, const float *res2, float *y,
                              int N, int H, int W, int Cin, int Cout, int KH, int KW,
                              int stride, int pad_h, int pad_w, int act, float slope, int sum_rule, void *stream);

/* PredictUpdate conv1 (pMCTF/layers/lifting_1d.py:38,44-45): 3x3, stride 1, pad 1, one input channel, Cout = 16
 * (other Cout: PMCTF_EINVAL).  y = conv(x); if y2 != NULL also y2 = act2(conv(x)) — the block needs both conv1 and
 * tanh(conv1).  x [N,H,W] plane, w OIHW on the device, y/y2 [N,H,W,16]. */
int pmctf_conv3x3_cin1_dual_f32(const float *x, const float *w_oihw, const float *bias, float *y, float *y2,
                                int N, int H, int W, int Cout, int act2, float slope, int sum_rule, void *stream);

/* KxK, stride 1, pad K/2, one or two output channels (PredictUpdate conv4 16->1 lifting_1d.py:41, PostProcess 64->1,
 * SpyNet 16->2 7x7): vector-ALU kernel, plain OIHW weights on the device, same sum order as pmctf_conv2d_nhwc_f32.
 * pmctf_conv2d_fewcout_supported() tells which (Cin, Cout, K) exist; others return PMCTF_EINVAL. */
int pmctf_conv2d_fewcout_supported(int Cin, int Cout, int K);
int pmctf_conv2d_fewcout_f32(const float *x, const float *w_oihw, const float *bias, const float *res1,
                             const float *res2, float *y, int N, int H, int W, int Cin, int Cout, int K,
                             int act, float slope, int sum_rule, void *stream);

/* depthwise KxK conv, stride 1, pad K/2 (pMCTF/layers/video/layers.py:117-118); x,y [N,H,W,C], w [C,1,K,K] */
int pmctf_dwconv2d_nhwc_f32(const float *x, const float *w, const float *bias, float *y,
                            int N, int H, int W, int C, int K, void *stream);

/* flow_warp / torch_warp (pMCTF/layers/video/video_net.py:32-55): bilinear grid_sample,
 * border padding, align_corners=True.  im,out: [N,C,H,W] planar; flow: [flowN,2,H,W] planar
 * (flowN = 1 broadcasts over N, pMCTF_L.py:299-300); lin_x[W], lin_y[H] are the cached
 * linspace(-1,1,.) tables of video_net.py:36-40.  flow_sign = -1 warps with -flow (pMCTF_L.py:307). */
int pmctf_flow_warp_f32(const float *im, const float *flow, const float *lin_x, const float *lin_y,
                        float *out, int N, int C, int H, int W, int flowN, float flow_sign, void *stream);

/* F.avg_pool2d(k=2,s=2) on NC planes (video_net.py:106-108) */
int pmctf_avgpool2_f32(const float *x, float *y, int NC, int H, int W, void *stream);
/* F.interpolate bilinear x2, align_corners=False, result times `scale` (video_net.py:58-63,114) */
int pmctf_bilinear_up2_f32(const float *x, float *y, int NC, int H, int W, float scale, void *stream);
/* F.interpolate bilinear /2, align_corners=False, result divided by `div` (video_net.py:66-71; pMCTF_L.py:317,401) */
int pmctf_bilinear_down2_f32(const float *x, float *y, int NC, int H, int W, float div, void *stream);
/* the same two for factor 2, 4 or 8: bilinearupsacling / bilineardownsacling(x, factor) of the me_downsample paths
 * (video_net.py:58-71; pMCTF_L.py:255-257,274-275,456-458,475-476,516-517) */
int pmctf_bilinear_up_f32(const float *x, float *y, int NC, int H, int W, int factor, float scale, void *stream);
int pmctf_bilinear_down_f32(const float *x, float *y, int NC, int H, int W, int factor, float div, void *stream);

/* ---- elementwise / layout family -------------------------------------------------------------
 * One strided kernel covers the reference's glue tensor ops (torch add/sub/mul/div, slicing, cat,
 * chunk, permute, tile, clamp, round, leaky_relu).  Operands are 4-D views (n,c,h,w) given by element
 * strides so[4]/sa[4]/sb[4] (HOST pointers to int64[4]; 0 = broadcast); b may be NULL.  cfast != 0
 * iterates c fastest (NHWC outputs), else w fastest (planar outputs).  One IEEE rounding per written
 * operation, never fused. */
#define PMCTF_EW_COPY 0              /* a                                                        */
#define PMCTF_EW_ADD 1               /* a + b            e.g. L = ref + inv, pMCTF_L.py:311       */
#define PMCTF_EW_SUB 2               /* a - b            e.g. H = cur - pred, pMCTF_L.py:305      */
#define PMCTF_EW_MUL 3               /* a * b                                                    */
#define PMCTF_EW_DIV 4               /* a / b                                                    */
#define PMCTF_EW_MULS 5              /* a * alpha        e.g. out * quant_step, video_net.py:143  */
#define PMCTF_EW_DIVS 6              /* a / alpha        e.g. subband / QP, pWave.py:199          */
#define PMCTF_EW_ADD_MULS 7          /* a + b*alpha                                              */
#define PMCTF_EW_SUB_MULS 8          /* a - b*alpha                                              */
#define PMCTF_EW_ADD_MULS_MULS 9     /* (a + b*alpha)*beta   predict/update_filter, wavelet_transform_temporal_mctf.py:27-45 */
#define PMCTF_EW_CLAMP_MULS 10       /* clamp(a*alpha, -beta, beta)   quantize_subband, pWave.py:184-189 */
#define PMCTF_EW_ROUND_CLAMP_MULS 11 /* round(clamp(a*alpha, -beta, beta))   pWave.py:408        */
#define PMCTF_EW_ROUND 12            /* round half to even (torch.round)                         */
#define PMCTF_EW_LEAKY 13            /* a > 0 ? a : a*alpha                                      */
#define PMCTF_EW_ADD_MULS2 14        /* a + (b*alpha)*beta   lifting step, lifting_1d.py:108-111  */
#define PMCTF_EW_SUB_MULS2 15        /* a - (b*alpha)*beta   inverse lifting, lifting_1d.py:155-160 */
#define PMCTF_EW_ROUND_CLAMP 16      /* round(clamp(a, alpha, beta))   harness, test_pMCTF_flex.py:299-300 */
#define PMCTF_EW_TANH 17             /* PM-F32 tanh(a)        lifting_1d.py:39                     */
#define PMCTF_EW_LAST 17
int pmctf_ew_f32(int op, float *out, const int64_t *so, const float *a, const int64_t *sa, const float *b,
                 const int64_t *sb, int N, int C, int H, int W, float alpha, float beta, int cfast, void *stream);

/* SpyNet level input torch.cat([im1, warp(im2), flow_up]) (video_net.py:116-119) for Y-only frames
 * (3 identical channels, pMCTF_L.py:453-454): planes im1[HW], warped[HW], flow_up[2][HW] -> NHWC [HW,8]. */
int pmctf_spynet_pack8_f32(const float *im1, const float *warped, const float *flow_up, float *out, int H, int W,
                           void *stream);
/* AUXILIARY reduced-precision profile of the dense 3x3 'same' stride-1 convolutions (conv_split.hip; SURVEY.md §7 step 5,
 * second variant): bf16-input / f32-accumulate MFMA with both operands split into nsplit bf16 planes (3: ~2^-22 relative
 * per product, 2: ~2^-16, 1: plain bf16).  Same interface and epilogue as pmctf_conv2d_nhwc_f32 (same reference call sites:
 * context_fusion_4step.py:12-20, postprocessing.py:9-18,35-44, context_fusion.py:56-128); NOT bit-exact against the
 * oracle — selected only by HipEngine(precision=...), never by the parity path.  Cin % 16 == 0, Cout in {64, 112}. */
int pmctf_conv3x3_split_supported(int Cin, int Cout);
int64_t pmctf_conv3x3_split_packed_size(int Cout, int Cin, int nsplit);          /* uint16 elements */
int pmctf_conv3x3_split_pack_weights(const float *w_oihw, const float *bias, int Cout, int Cin, int nsplit,
                                     uint16_t *w_packed, float *bias_packed);     /* HOST pointers */
int pmctf_conv3x3_split_f32(const float *x, const uint16_t *w_packed, const float *bias_packed, const float *res1,
                            const float *res2, float *y, int N, int H, int W, int Cin, int Cout, int nsplit, int act,
                            float slope, void *stream);
/* the same arithmetic at stride 2 with explicit geometry (the shape of pmctf_conv2d_nhwc_geom_f32: the quarter-resolution
 * context convolutions of context_fusion_4step.py:176-191, evaluated only at one parity class); 112 couts */
int pmctf_conv3x3_split_geom_f32(const float *x, const uint16_t *w_packed, const float *bias_packed, const float *res1,
                                 const float *res2, float *y, int N, int H, int W, int Cin, int Cout, int nsplit,
                                 int stride, int pad_h, int pad_w, int Ho, int Wo, int act, float slope, void *stream);

/* The whole PredictUpdate CNN of a lifting step as ONE launch (pu_fused.hip): 1->16, 16->16 tanh, 16->16 (+c1), 16->1,
 * 3x3 zero-padded, with the arithmetic around it.  x / other / out: single-channel planes (N,1,H,W).
 *   mode 0 — temporal predict / update filter, pMCTF/layers/wavelet_transform_temporal_mctf.py:27-45:
 *            out = (x + PU(x) * 0.1) * c                                     (c = 1/sqrt2 predict, 0.5 update)
 *   mode 1 — one branch of the iWave lifting step, pMCTF/layers/lifting_1d.py:103-145 (forward) / :147-189 (backward):
 *            skip = conv3x1(reflect_pad_H(x)) + lbias;  out = other + sign * (skip + (PU(skip / 256) * 256) * 0.1)
 * PredictUpdate itself: pMCTF/layers/lifting_1d.py:25-49.  w1/b1, w4/b4: OIHW device filters (16,1,3,3)+(16), (1,16,3,3)+(1);
 * w2/w3: pmctf_conv2d_pack_weights() output for the two (16,16,3,3) layers.  Same sums, same order as the separate
 * launches (pmctf_conv3x3_cin1_dual_f32, pmctf_conv2d_nhwc_f32, pmctf_conv2d_fewcout_f32, pmctf_lift_skip3_f32, pmctf_ew_f32). */
int pmctf_predict_update_fused_f32(const float *x, const float *other, float *out, const float *w1, const float *b1,
                                   const float *w2_packed, const float *b2_packed, const float *w3_packed,
                                   const float *b3_packed, const float *w4, const float *b4, int N, int H, int W, int mode,
                                   float c, float sign, float lw0, float lw1, float lw2, float lbias, int sum_rule,
                                   int skip_sum_rule, void *stream);

/* ReflectionPad2d((0,0,1,1)) + 3x1 conv on NC single-channel planes (lifting_1d.py:98,105-106).  sum_rule: one input
 * channel = one block, so PMCTF_SUM_BLOCKS is "three fmaf from zero, bias last" (ATen's oneDNN path), PMCTF_SUM_CHAIN "bias
 * first" — which is what ATen computes when the padded input of ONE plane has at most 20 480 elements and the reference
 * tensor holds one plane (its im2col + gemv path; Convolution.cpp `use_mkldnn`): the caller passes the rule of the
 * REFERENCE's tensor shape, not of a batch it may have stacked. */
int pmctf_lift_skip3_f32(const float *x, float *y, int NC, int H, int W, float w0, float w1, float w2, float bias,
                         int sum_rule, void *stream);
/* nn.Upsample(scale_factor=2, mode="nearest") on NHWC (context_fusion_4step.py:50, long_context.py:50) */
int pmctf_nearest_up2_nhwc_f32(const float *x, float *y, int N, int H, int W, int C, void *stream);
/* nn.PixelShuffle(2) on NHWC [N,H,W,4C] -> [N,2H,2W,C], optional activation (video/layers.py:34-38,96-104) */
int pmctf_pixel_shuffle2_nhwc_f32(const float *x, float *y, int N, int H, int W, int C, int act, float slope,
                                  void *stream);
/* ConvFFN3 gate: leaky(x1,0.1)+leaky(x2,0.01), x = [x1|x2] NHWC [P,2C] -> [P,C] (video/layers.py:163-167) */
int pmctf_ffn3_mix_f32(const float *x, float *y, int64_t P, int C, void *stream);
/* LSTM2D gates (long_context.py:20-33); xh = conv_in(x)+conv_hidden(h), NHWC [P,C]; cell [P,Ccell], Ccell in {1,C} */
int pmctf_lstm_gates_f32(const float *xh, const float *cell, float *cell_out, float *hid_out, int64_t P, int C,
                         int Ccell, void *stream);
/* The same with torch.sigmoid as ATen evaluates it on the reference's contiguous (ref_planes, C, H, W) gate tensor with
 * aten_threads intra-op threads: whole strides of 32 floats of a thread's slice through SLEEF, the rest of the slice
 * through libm's expf (pm_glibc_expf.h).  HW = H * W; xh holds P / HW planes, ref_planes per reference tensor.
 * aten_threads = 0: no tails (= pmctf_lstm_gates_f32). */
int pmctf_lstm_gates_aten_f32(const float *xh, const float *cell, float *cell_out, float *hid_out, int64_t P, int C,
                              int Ccell, int64_t HW, int ref_planes, int aten_threads, void *stream);

/* ---- quantisation + symbol hand-off --------------------------------
 * sym/idx receive one full-size push (int16 symbol, int16 CDF row) in the reference's flattening order
 * (NCHW), exactly what EntropyCoder.encode_with_indexes is given (entropy_models.py:37-40,269-278). */
/* One step of ContextFusionFourStep.forward(write=True): process_with_mask + build_indexes for parity class k
 * (context_fusion_4step.py:127-137,156-189); x, so_far planes [N,1,H,W].
 * params: NHWC [N,H,W,2] (scale, mean) (params_sub = 0) or, for params computed only at the class-k positions,
 * [N,H/2,W/2,2] (params_sub = 1; H and W even). */
int pmctf_fourstep_quant_f32(const float *x, const float *params, float *so_far, int16_t *sym, int16_t *idx, int N,
                             int H, int W, int k, int params_sub, float log_scale_min, float log_scale_step,
                             void *stream);
/* LL subband, one-shot path of pWave.compress (pWave.py:408-418; CompressionModel.process gaussian_model.py:59-63):
 * ll_hat = round(round(round(ll) - mean) + mean), symbol round(round(ll) - mean), CDF row of scale.
 * planes > 0: push in the order of the sequential coder (_compress_subband_ar, pWave.py:531-555): position-major,
 * plane-minor (total = planes * positions); planes = 0: plane-major NCHW order of the one-shot path (pWave.py:418). */
int pmctf_ll_quant_f32(const float *ll, const float *params, float *ll_hat, int16_t *sym, int16_t *idx, int64_t total,
                       int planes, float log_scale_min, float log_scale_step, void *stream);
/* MV hyper latent (pMCTF_L.py:463-464,478): z_hat = round(z) (NHWC [HW][C]); symbols [C][HW] with CDF row = channel
 * (BitEstimator.build_indexes / encode, entropy_models.py:181-195) */
int pmctf_z_symbols_f32(const float *z, float *z_hat, int16_t *sym, int16_t *idx, int HW, int C, void *stream);
/* step t of MVCoderQuad.forward_four_part_prior(write=True) (four_part_prior.py:89-208): y [HW][64], common [HW][192]
 * (quant step | scales | means, LowerBound 0.5 on the step), sp [HW][128] spatial-prior output (t > 0); writes the
 * step's y_hat contribution into so_far [HW][64] and the 16-channel symbol/row planes [16][HW] */
int pmctf_mv_fourpart_step_f32(const float *y, const float *common, const float *sp, float *so_far, int16_t *sym,
                               int16_t *idx, int H, int W, int t, float log_scale_min, float log_scale_step,
                               void *stream);
/* y_hat = so_far * q_dec with q_dec = max(quant step, 0.5) (four_part_prior.py:194-197; video_net.py:14-20) */
int pmctf_mv_dequant_f32(const float *so_far, const float *common, float *y_hat, int64_t HW, void *stream);

/* ---- decoder side ----------------------------------------------------------------
 * LL subband: pWave._decompress_subband_ar + ContextFusionSubband.forward_sequential (pWave.py:557-584,
 * context_fusion.py:140-204) as one persistent workgroup that evaluates the causal masked-conv network per
 * position and decodes the position's symbols from the rANS stream inside the kernel.
 *   w_packed: pmctf_ll_ar_pack_weights() output on the device; stream_words: the rANS payload (after the 1-byte
 *   header) as little-endian uint32 on the device; (x0,pos0) / state_out[0..1]: Rans64 state and next-word index
 *   before / after (state_out[2] != 0: stream exhausted); cdf/sizes/offsets: the Laplace tables on the device;
 *   ll_out [N][H][W]; scratch_zeroed: pmctf_ll_ar_scratch_floats(N,H,W) floats, zero-filled.  N <= 4
 *   planes per stream (Y 1, UV 2, RGB 3).  pack_weights takes the masked weights of maskedConv1 (w_a,b_a), the five
 *   type-B layers residualBlocks.{0,1}.conv{1,2} + maskedConv2 (w_b,b_b) and convs.{0,1,2} (context_fusion.py:79-135). */
int64_t pmctf_ll_ar_packed_size(void);
int pmctf_ll_ar_pack_weights(const float *w_a, const float *b_a, const float *const *w_b, const float *const *b_b,
                             const float *w_p0, const float *b_p0, const float *w_p1, const float *b_p1,
                             const float *w_p2, const float *b_p2, float *out);      /* HOST pointers */
int64_t pmctf_ll_ar_scratch_floats(int N, int H, int W);
int pmctf_ll_ar_decode_f32(const float *w_packed, const uint32_t *stream_words, int64_t n_words, uint64_t x0,
                           int64_t pos0, const int32_t *cdf, const int32_t *sizes, const int32_t *offsets, int cdf_cols,
                           float log_scale_min, float log_scale_step, float *ll_out, float *scratch_zeroed, int N, int H,
                           int W, uint64_t *state_out, void *stream);
/* The same decode under the summation rules the encoder's one-shot LL network ran with (PMCTF_SUM_*): sum_rule_3x3 for
 * the masked 3x3 layers (CHAIN or BLOCKS), sum_rule_head for convs.0 / convs.1 and sum_rule_head_out for convs.2
 * (CHAIN, or the block size B of "reduce-B", a multiple of 16).  pmctf_ll_ar_decode_f32 = all three CHAIN. */
int pmctf_ll_ar_decode_rules_f32(const float *w_packed, const uint32_t *stream_words, int64_t n_words, uint64_t x0,
                                 int64_t pos0, const int32_t *cdf, const int32_t *sizes, const int32_t *offsets,
                                 int cdf_cols, float log_scale_min, float log_scale_step, float *ll_out,
                                 float *scratch_zeroed, int N, int H, int W, uint64_t *state_out, int sum_rule_3x3,
                                 int sum_rule_head, int sum_rule_head_out, void *stream);
/* ContextFusionFourStep.decompress (context_fusion_4step.py:196-249): CDF rows handed to decode_stream for step k
 * (0 off the mask), then x_hat = (q + mean) on the mask.  params as in pmctf_fourstep_quant_f32. */
int pmctf_fourstep_indexes_f32(const float *params, int16_t *idx, int N, int H, int W, int k, int params_sub,
                               float log_scale_min, float log_scale_step, void *stream);
int pmctf_fourstep_dequant_f32(const int16_t *sym, const float *params, float *so_far, int N, int H, int W, int k,
                               int params_sub, void *stream);
/* MVCoderQuad.decompress_four_part_prior (four_part_prior.py:217-280), step t; idx/sym: [16][H*W] */
int pmctf_mv_fourpart_indexes_f32(const float *common, const float *sp, int16_t *idx, int H, int W, int t,
                                  float log_scale_min, float log_scale_step, void *stream);
int pmctf_mv_fourpart_dequant_f32(const int16_t *sym, const float *common, const float *sp, float *so_far, int H, int W,
                                  int t, void *stream);
/* decoded int16 symbols [C][HW] -> float NHWC [HW][C] (mv_z_hat, pMCTF_L.py:504-505) */
int pmctf_sym_to_nhwc_f32(const int16_t *sym, float *out, int HW, int C, void *stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Estimate-mode forward (pMCTF.forward_one_stage, pMCTF_L.py:332-379; pWave.forward_one_channel, pWave.py:243-312):
 * Laplace / factorized bit ESTIMATES (gaussian_model.py:36-53,65-67) and squared-error sums instead of range coding.
 * Per element: sigma = clamp(s, 1e-5, 1e10); cdf(v) = 0.5 - 0.5*sign(v)*(exp(-|v|/sigma) - 1);
 * bits = max(-log2(cdf(y+0.5) - cdf(y-0.5) + 1e-5), 0) in f32; totals are f64 sums.  Every function ADDS to device
 * doubles the caller has zeroed. */

/* step k of ContextFusionFourStep.forward (context_fusion_4step.py:156-186): writes x_hat at the class-k positions
 * of so_far (zeroes the rest when k == 0) and adds their bits to bits_per_plane[n].  Layouts as pmctf_fourstep_quant_f32. */
int pmctf_fourstep_estimate_f32(const float *x, const float *params, float *so_far, int N, int H, int W, int k,
                                int params_sub, double *bits_per_plane, void *stream);
/* LL subband (pWave.py:255-263): bits of ll_hat - mean (not rounded again) under scale; params [N*HW][2]. */
int pmctf_ll_estimate_f32(const float *ll_hat, const float *params, int N, int64_t HW, double *bits_per_plane,
                          void *stream);
/* MV hyper latent (pMCTF_L.py:262,283; entropy_models.py:72-77,114-122): z_hat = round(z) (NHWC [HW][C]) and the bits
 * of the factorized prior; consts [11][C] = softplus(h) of f1..f4, b of f1..f4, tanh(a) of f1..f3. */
int pmctf_z_estimate_f32(const float *z, float *z_hat, const float *consts, int64_t HW, int C, double *bits,
                         void *stream);
/* step t of MVCoderQuad.forward_four_part_prior (four_part_prior.py:89-195); layouts as pmctf_mv_fourpart_step_f32. */
int pmctf_mv_fourpart_estimate_f32(const float *y, const float *common, const float *sp, float *so_far, int H, int W,
                                   int t, double *bits, void *stream);
/* nn.MSELoss numerator (pMCTF_L.py:351,373; pWave.py:309): sum += sum((a[i]-b[i])^2) */
int pmctf_sqdiff_sum_f32(const float *a, const float *b, int64_t n, double *sum, void *stream);

#ifdef __cplusplus
}
#endif
#endif
