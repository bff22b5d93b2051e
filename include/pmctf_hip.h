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

/* Same contract for Cin <= 4 (any Cin >= 1), plain OIHW weights on the device: direct
 * convolution on the vector ALU (first layers: 1->16, 1->64, 1->112, 1->128, 2->112, 2->64 ...). */
int pmctf_conv2d_smallcin_f32(const float *x, const float *w_oihw, const float *bias,
                              const float *res1, const float *res2, float *y,
                              int N, int H, int W, int Cin, int Cout, int KH, int KW,
                              int stride, int pad_h, int pad_w, int act, float slope, void *stream);

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

#ifdef __cplusplus
}
#endif
#endif
