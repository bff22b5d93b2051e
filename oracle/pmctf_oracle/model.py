"""ORACLE — test infrastructure only; never imported by the product path.

Functional CPU restatement of the reference's encode hot path over a plain
state_dict.  Every function cites the reference lines it follows.  Numeric primitives come from a
back-end in kernels.py (TorchK = the ATen ops the reference calls; CdefK = the PM-F32 C
restatement that the HIP kernels match bit for bit).

Pinning: tools/make_golden.py imports the real reference in the build container, runs it on the
build's deterministic weights and writes tests/golden/*.npz; tests/test_oracle_vs_golden.py checks
this restatement against those fixtures.
"""
import math
import os
import struct

import numpy as np
import torch
import torch.nn.functional as F

from . import aten_rules, entropy
from .kernels import CdefK, TorchK


def get_padding_size(height, width, p=64):
    """pMCTF/utils/stream_helper.py:23-32"""
    new_h = (height + p - 1) // p * p
    new_w = (width + p - 1) // p * p
    return 0, new_w - width, 0, new_h - height


def get_rounded_q(q_scale):
    """pMCTF/utils/stream_helper.py:41-45"""
    q_scale = float(np.clip(np.asarray(q_scale, dtype=np.float64).reshape(-1)[0], 0.01, 655.))
    q_index = int(np.round(q_scale * 100))
    return q_index / 100, q_index


def encode_p_bytes(string, mv_y_q_index):
    """stream_helper.py:181-186: >H q_index, >I length, payload"""
    return struct.pack(">H", mv_y_q_index) + struct.pack(">I", len(string)) + string


def encode_image_bytes(height, width, num_channels, bit_stream):
    """stream_helper.py:201-207: >III h,w,c ; >I length ; payload"""
    return struct.pack(">III", height, width, num_channels) + struct.pack(">I", len(bit_stream)) + bit_stream


QP_NUM = 21  # pMCTF_L.py:217-219, pWave.py:227-229


def get_curr_q(q_scale, q_index):
    """pMCTF_L.py:195-209 / pWave.py:209-225 (torch CPU scalar arithmetic on the (2,1,1,1) parameter)"""
    min_q = q_scale[0:1]
    max_q = q_scale[1:2]
    step = (torch.log(max_q) - torch.log(min_q)) / (QP_NUM - 1)
    return torch.exp(torch.log(min_q) + step * q_index)


class Oracle:
    def __init__(self, state_dict, num_me_stages=1, backend="cdef", decomp_levels=4, aten_all=None, aten_threads=8):
        self.K = CdefK() if backend == "cdef" else TorchK()
        if aten_all is None:       # follow the profile the product's models take from the environment (PMCTF_PRECISION)
            aten_all = os.environ.get("PMCTF_PRECISION", "f32") == "f32"
        self.aten_all = aten_all   # PM-F32 back-end: ATen's summation order in every layer (False: the product's "f32-chain")
        if aten_all and backend == "cdef":
            # ... and ATen's split of an elementwise op over its intra-op threads (8 on the machine the fixtures under
            # tests/golden were generated on): torch.sigmoid's scalar tails (oracle/c/pm_glibc_expf.h)
            self.K.aten_threads = aten_threads
        self.num_me_stages = num_me_stages
        self.L = decomp_levels
        sd = {k: v.detach().clone().float() for k, v in state_dict.items()}
        # MaskedConv2d.forward multiplies weight.data by the mask in place (layers/layers.py:49-51)
        for k in list(sd.keys()):
            if k.endswith(".mask"):
                sd[k[:-5] + ".weight"] = sd[k[:-5] + ".weight"] * sd[k]
        self.sd = sd
        self.tables = entropy.GaussianTables()
        self.bit_est = [entropy.BitEstimatorTables(sd, f"mv_bit_est.{i}.") for i in range(num_me_stages)]
        self.trace = None          # list of (symbols, indexes) pushes of the current stream
        self.taps = {}             # named intermediates for parity tests
        self.tap_enabled = False

    # ------------------------------------------------------------------ helpers
    def tap(self, name, t):
        if self.tap_enabled:
            self.taps[name] = t.detach().clone()

    SIGNAL_PATH = ("optic_flow.", "mv_", "temporal_filtering.")

    def sum_rule(self, p, x, w, groups=1, stride=1):
        """Summation rule of the PM-F32 back-end for the convolution with parameter key p (oracle/c/pm_ops.c: 0 = one chain
        from the bias, 1 = per-16-channel-block sums from zero added in turn, bias after the first block, B >= 16 = a 1x1
        layer's reduction in blocks of B channels).  KH*KW > 1 layers of the signal path (motion estimation, motion codec,
        temporal and spatial lifting) and of the post-processing CNN follow rule 1, which is what ATen's CPU path computes
        for them (measured: tools/aten_conv_rules.py); 1x1 layers the chain unless ATen blocks the reduction
        (aten_rules); the entropy-parameter networks rule 0 — or, with aten_all, ATen's order like the signal path.  One
        shape-dependent exception, also ATen's: a lifting step's 3x1 filter whose (reflect-padded) input is ONE plane of
        at most 20 480 elements does not go through oneDNN (Convolution.cpp `use_mkldnn`) but through im2col + gemv,
        which starts from the bias — rule 0."""
        if groups != 1:
            return 0
        if not (self.aten_all or p.startswith(self.SIGNAL_PATH) or ".wavelet_transform." in p or ".dequantModule." in p):
            return 0
        if w.size(2) * w.size(3) == 1:
            if stride == 1 and (w.size(1) >= 112 or self.aten_all):
                return aten_rules.conv1x1_sum_rule(w.size(1), w.size(0), x.size(0), x.size(2), x.size(3))
            return 0
        if w.size(0) == 1 and w.size(1) == 1 and w.size(3) == 1 and x.size(0) == 1 and x.numel() <= 20480:
            return 0
        if self.aten_all and stride == 1:
            return aten_rules.conv_kxk_sum_rule(w.size(1), w.size(0), w.size(2), w.size(3), x.size(0), x.size(2), x.size(3))
        return 1

    def conv(self, p, x, stride=1, padding=0, groups=1):
        w = self.sd[p + ".weight"]
        if self.K.name == "cdef":
            return self.K.conv2d(x, w, self.sd.get(p + ".bias"), stride=stride, padding=padding, groups=groups,
                                 rule=self.sum_rule(p, x, w, groups, stride))
        return self.K.conv2d(x, w, self.sd.get(p + ".bias"), stride=stride, padding=padding, groups=groups)

    # ------------------------------------------------------------------ a6: PredictUpdate, lifting_1d.py:36-49
    def predict_update(self, p, x):
        conv1 = self.conv(p + ".conv1", x, padding=1)
        t = self.K.tanh(conv1)
        t = self.conv(p + ".conv2", t, padding=1)
        t = self.K.tanh(t)
        t = self.conv(p + ".conv3", t, padding=1)
        t = conv1 + t
        return self.conv(p + ".conv4", t, padding=1)

    # wavelet_transform_temporal_mctf.py:27-45
    def predict_filter(self, stage, x):
        tmp = self.predict_update(f"temporal_filtering.{stage}.P_t", x) * 0.1
        x = x + tmp
        return x * torch.tensor(1 / math.sqrt(2))

    def update_filter(self, stage, x):
        tmp = self.predict_update(f"temporal_filtering.{stage}.U_t", x) * 0.1
        x = x + tmp
        return x * torch.tensor(0.5)

    # ------------------------------------------------------------------ a5/a7: pMCTF_L.py:297-330
    def forward_MCTF(self, ref, cur, mv_hat, stage_idx=0):
        me = min(self.num_me_stages - 1, stage_idx)
        if ref.size(0) > mv_hat.size(0):
            mv_hat = mv_hat.tile((ref.size(0), 1, 1, 1))
        pred = self.K.flow_warp(ref, mv_hat)
        pred = self.predict_filter(me, pred)
        H_t = cur - pred
        inv = self.K.flow_warp(H_t, -mv_hat)
        inv = self.update_filter(me, inv)
        L_t = ref + inv
        return L_t, H_t, pred, inv

    def inverse_MCTF(self, L_t, H_t, mv_hat, downscale=False, stage_idx=0):
        me = min(self.num_me_stages - 1, stage_idx)
        if downscale:
            mv_hat = self.K.bilinear_down2(mv_hat) / 2
        if L_t.size(0) > mv_hat.size(0):
            mv_hat = mv_hat.tile((L_t.size(0), 1, 1, 1))
        inv = self.K.flow_warp(H_t, -mv_hat)
        inv = self.update_filter(me, inv)
        ref = L_t - inv
        pred = self.K.flow_warp(ref, mv_hat)
        pred = self.predict_filter(me, pred)
        cur = H_t + pred
        return ref, cur

    # ------------------------------------------------------------------ a3: SpyNet, video_net.py:74-121
    def me_basic(self, level, x):
        p = f"optic_flow.moduleBasic.{level}"
        x = F.relu(self.conv(p + ".conv1", x, padding=3))
        x = F.relu(self.conv(p + ".conv2", x, padding=3))
        x = F.relu(self.conv(p + ".conv3", x, padding=3))
        x = F.relu(self.conv(p + ".conv4", x, padding=3))
        return self.conv(p + ".conv5", x, padding=3)

    def spynet(self, im1, im2, Lv=6):
        im1_list, im2_list = [im1], [im2]
        for level in range(Lv - 1):
            im1_list.append(self.K.avg_pool2(im1_list[level]))
            im2_list.append(self.K.avg_pool2(im2_list[level]))
        shape_fine = im2_list[Lv - 1].size()
        flow = torch.zeros([im1.size(0), 2, shape_fine[2] // 2, shape_fine[3] // 2], dtype=im1.dtype)
        for level in range(Lv):
            flow_up = self.K.bilinear_up2(flow) * 2.0
            idx = Lv - 1 - level
            inp = torch.cat([im1_list[idx], self.K.flow_warp(im2_list[idx], flow_up), flow_up], 1)
            flow = flow_up + self.me_basic(level, inp)
            self.tap(f"spynet.flow{level}", flow)
        return flow

    # ------------------------------------------------------------------ video/layers.py building blocks
    def res_block_stride(self, p, x, stride=2):
        """ResidualBlockWithStride, video/layers.py:46-77"""
        out = self.conv(p + ".conv1", x, stride=stride, padding=1)
        out = F.leaky_relu(out, 0.01)
        out = self.conv(p + ".conv2", out, padding=1)
        out = F.leaky_relu(out, 0.1)
        identity = self.conv(p + ".downsample", x, stride=stride) if stride != 1 else x
        return out + identity

    def res_block_up(self, p, x):
        """ResidualBlockUpsample, video/layers.py:80-105"""
        out = F.pixel_shuffle(self.conv(p + ".subpel_conv.0", x), 2)
        out = F.leaky_relu(out, 0.01)
        out = self.conv(p + ".conv", out, padding=1)
        out = F.leaky_relu(out, 0.1)
        identity = F.pixel_shuffle(self.conv(p + ".upsample.0", x), 2)
        return out + identity

    def depth_conv(self, p, x):
        """DepthConv (stride 1), video/layers.py:108-136"""
        identity = x
        if (p + ".adaptor.weight") in self.sd:
            identity = self.conv(p + ".adaptor", x)
        out = F.leaky_relu(self.conv(p + ".conv1.0", x), 0.01)
        C = out.size(1)
        out = self.conv(p + ".depth_conv", out, padding=1, groups=C)
        out = self.conv(p + ".conv2", out)
        return out + identity

    def conv_ffn(self, p, x):
        """ConvFFN, video/layers.py:139-153"""
        t = F.leaky_relu(self.conv(p + ".conv.0", x), 0.1)
        t = F.leaky_relu(self.conv(p + ".conv.2", t), 0.1)
        return x + t

    def conv_ffn3(self, p, x):
        """ConvFFN3, video/layers.py:155-168"""
        x1, x2 = self.conv(p + ".conv", x).chunk(2, 1)
        out = F.leaky_relu(x1, 0.1) + F.leaky_relu(x2, 0.01)
        return x + self.conv(p + ".conv_out", out)

    def depth_conv_block(self, p, x):
        """DepthConvBlock, video/layers.py:171-181"""
        return self.conv_ffn(p + ".block.1", self.depth_conv(p + ".block.0", x))

    def depth_conv_block4(self, p, x):
        """DepthConvBlock4, video/layers.py:184-193"""
        return self.conv_ffn3(p + ".block.1", self.depth_conv(p + ".block.0", x))

    # ------------------------------------------------------------------ MV codec, video_net.py:124-191
    def mv_enc(self, s, x, context, quant_step):
        p = f"mv_encoder.{s}"
        out = self.res_block_stride(p + ".enc_1.0", x)
        out = self.depth_conv_block(p + ".enc_1.1", out)
        out = out * quant_step
        out = self.res_block_stride(p + ".enc_2", out)
        if context is None:
            out = self.depth_conv_block(p + ".adaptor_0", out)
        else:
            out = self.depth_conv_block(p + ".adaptor_1", torch.cat((out, context), dim=1))
        out = self.res_block_stride(p + ".enc_3.0", out)
        out = self.depth_conv_block(p + ".enc_3.1", out)
        return self.conv(p + ".enc_3.2", out, stride=2, padding=1)

    def mv_dec(self, s, x, quant_step):
        p = f"mv_decoder.{s}"
        f = self.depth_conv_block(p + ".dec_1.0", x)
        f = self.res_block_up(p + ".dec_1.1", f)
        f = self.depth_conv_block(p + ".dec_1.2", f)
        f = self.res_block_up(p + ".dec_1.3", f)
        feature = self.depth_conv_block(p + ".dec_1.4", f)
        out = self.res_block_up(p + ".dec_2", feature)
        out = out * quant_step
        out = self.depth_conv_block(p + ".dec_3.0", out)
        mv = F.pixel_shuffle(self.conv(p + ".dec_3.1.0", out), 2)
        return mv, feature

    def mv_hyper_enc(self, s, x):
        p = f"mv_hyper_prior_encoder.{s}"
        x = self.depth_conv_block4(p + ".0", x)
        x = self.conv(p + ".1", x, stride=2, padding=1)
        x = F.leaky_relu(x, 0.01)
        return self.conv(p + ".3", x, stride=2, padding=1)

    def mv_hyper_dec(self, s, x):
        p = f"mv_hyper_prior_decoder.{s}"
        x = self.res_block_up(p + ".0", x)
        x = self.res_block_up(p + ".1", x)
        return self.depth_conv_block4(p + ".2", x)

    def mv_prior_param_decoder(self, mv_z_hat, dpb, s):
        """pMCTF_L.py:232-241"""
        params = self.mv_hyper_dec(s, mv_z_hat)
        ref_mv_y = dpb["ref_mv_y"]
        if ref_mv_y is None:
            params = self.depth_conv_block(f"mv_y_prior_fusion_adaptor_0.{s}", params)
        else:
            params = self.depth_conv_block(f"mv_y_prior_fusion_adaptor_1.{s}", torch.cat((params, ref_mv_y), dim=1))
        params = self.depth_conv_block(f"mv_y_prior_fusion.{s}.0", params)
        return self.depth_conv_block(f"mv_y_prior_fusion.{s}.1", params)

    @staticmethod
    def masks4(H, W):
        """get_mask_four_parts, four_part_prior.py:51-78 / context_fusion_4step.py:92-119"""
        out = []
        for m in (((1, 0), (0, 0)), ((0, 1), (0, 0)), ((0, 0), (1, 0)), ((0, 0), (0, 1))):
            mm = torch.tensor(m, dtype=torch.float32).repeat((H + 1) // 2, (W + 1) // 2)[:H, :W]
            out.append(mm[None, None])
        return out

    @staticmethod
    def process_with_mask(y, scales, means, mask):
        """four_part_prior.py:40-49 / context_fusion_4step.py:127-137"""
        scales_hat = scales * mask
        means_hat = means * mask
        y_res = (y - means_hat) * mask
        y_q = torch.round(y_res)
        y_hat = y_q + means_hat
        return y_res, y_q, y_hat, scales_hat

    def mv_spatial_prior(self, s, adaptor, params):
        x = self.conv(f"mv_y_spatial_prior_adaptor_{adaptor}.{s}", params)
        for i in range(3):
            x = self.depth_conv_block(f"mv_y_spatial_prior.{s}.{i}", x)
        return x.chunk(8, 1)

    def compress_four_part_prior(self, s, y, common_params, full=False):
        """MVCoderQuad.forward_four_part_prior(write=True), four_part_prior.py:89-208 (enc_dec_quant=True)"""
        quant_step, scales, means = common_params.chunk(3, 1)
        quant_step = torch.max(quant_step, torch.ones_like(quant_step) * 0.5)      # LowerBound, video_net.py:14-20
        q_enc = 1. / quant_step
        q_dec = quant_step
        _, _, H, W = y.size()
        m0, m1, m2, m3 = self.masks4(H, W)
        y = y * q_enc
        y_0, y_1, y_2, y_3 = y.chunk(4, 1)
        s0, s1, s2, s3 = scales.chunk(4, 1)
        u0, u1, u2, u3 = means.chunk(4, 1)
        pm = self.process_with_mask
        _, q00, h00, sh00 = pm(y_0, s0, u0, m0)
        _, q11, h11, sh11 = pm(y_1, s1, u1, m1)
        _, q22, h22, sh22 = pm(y_2, s2, u2, m2)
        _, q33, h33, sh33 = pm(y_3, s3, u3, m3)
        so_far = torch.cat((h00, h11, h22, h33), dim=1)
        s0, s1, s2, s3, u0, u1, u2, u3 = self.mv_spatial_prior(s, 1, torch.cat((so_far, common_params), dim=1))
        _, q03, h03, sh03 = pm(y_0, s0, u0, m3)
        _, q12, h12, sh12 = pm(y_1, s1, u1, m2)
        _, q21, h21, sh21 = pm(y_2, s2, u2, m1)
        _, q30, h30, sh30 = pm(y_3, s3, u3, m0)
        so_far = so_far + torch.cat((h03, h12, h21, h30), dim=1)
        s0, s1, s2, s3, u0, u1, u2, u3 = self.mv_spatial_prior(s, 2, torch.cat((so_far, common_params), dim=1))
        _, q02, h02, sh02 = pm(y_0, s0, u0, m2)
        _, q13, h13, sh13 = pm(y_1, s1, u1, m3)
        _, q20, h20, sh20 = pm(y_2, s2, u2, m0)
        _, q31, h31, sh31 = pm(y_3, s3, u3, m1)
        so_far = so_far + torch.cat((h02, h13, h20, h31), dim=1)
        s0, s1, s2, s3, u0, u1, u2, u3 = self.mv_spatial_prior(s, 3, torch.cat((so_far, common_params), dim=1))
        _, q01, h01, sh01 = pm(y_0, s0, u0, m1)
        _, q10, h10, sh10 = pm(y_1, s1, u1, m0)
        _, q23, h23, sh23 = pm(y_2, s2, u2, m3)
        _, q32, h32, sh32 = pm(y_3, s3, u3, m2)
        # combine_four_parts (:80-87)
        y_hat = torch.cat((h00 + h01 + h02 + h03, h10 + h11 + h12 + h13, h20 + h21 + h22 + h23,
                           h30 + h31 + h32 + h33), dim=1)
        y_hat = y_hat * q_dec
        qw = [q00 + q11 + q22 + q33, q03 + q12 + q21 + q30, q02 + q13 + q20 + q31, q01 + q10 + q23 + q32]
        sw = [sh00 + sh11 + sh22 + sh33, sh03 + sh12 + sh21 + sh30, sh02 + sh13 + sh20 + sh31,
              sh01 + sh10 + sh23 + sh32]
        if full:    # write=False outputs: combine_four_parts of the quantised residuals and of the scales (:80-87,177-192)
            y_q = torch.cat((q00 + q01 + q02 + q03, q10 + q11 + q12 + q13, q20 + q21 + q22 + q23,
                             q30 + q31 + q32 + q33), dim=1)
            scales_hat = torch.cat((sh00 + sh01 + sh02 + sh03, sh10 + sh11 + sh12 + sh13, sh20 + sh21 + sh22 + sh23,
                                    sh30 + sh31 + sh32 + sh33), dim=1)
            return y_q, y_hat, scales_hat
        return qw, sw, y_hat

    # ------------------------------------------------------------------ entropy hand-off (a16)
    def new_stream(self):
        self.trace = []
        self.ec = entropy.EntropyCoder(trace=self.trace)
        self.ec.reset()

    def gaussian_encode(self, x, scales):
        """GaussianEncoder.encode, entropy_models.py:275-278"""
        idx = self.K.build_indexes(self.tables, scales)
        self.ec.encode_with_indexes(x.reshape(-1), idx.reshape(-1), *self.tables.cdf_info())

    def finish_stream(self):
        self.ec.flush()
        return self.ec.get_encoded_stream()

    # ------------------------------------------------------------------ a2: compress_mv, pMCTF_L.py:448-495
    def get_mv_y_q(self, q_index, s):
        enc = get_curr_q(self.sd[f"mv_y_q_scale_enc.{s}"], q_index)
        enc, _ = get_rounded_q(enc.numpy())
        dec = get_curr_q(self.sd[f"mv_y_q_scale_dec.{s}"], q_index)
        dec, _ = get_rounded_q(dec.numpy())
        return enc, dec

    def compress_mv(self, ref_y, cur_y, dpb, stage_idx=0, q_index=0, me_downsample=1):
        s = min(self.num_me_stages - 1, stage_idx)
        q_enc, q_dec = self.get_mv_y_q(q_index, s)
        mv_x = cur_y.tile((1, 3, 1, 1)) / 255
        mv_ref = ref_y.tile((1, 3, 1, 1)) / 255
        if me_downsample > 1:                                   # pMCTF_L.py:456-458
            mv_x = self.K.bilinear_down2(mv_x, me_downsample)
            mv_ref = self.K.bilinear_down2(mv_ref, me_downsample)
        est_mv = self.spynet(mv_x, mv_ref)
        self.tap("est_mv", est_mv)
        mv_y = self.mv_enc(s, est_mv, dpb["mv_feature"], q_enc)
        self.tap("mv_y", mv_y)
        mv_z = self.mv_hyper_enc(s, mv_y)
        mv_z_hat = torch.round(mv_z)
        self.tap("mv_z_hat", mv_z_hat)
        mv_params = self.mv_prior_param_decoder(mv_z_hat, dpb, s)
        self.tap("mv_params", mv_params)
        qw, sw, mv_y_hat = self.compress_four_part_prior(s, mv_y, mv_params)
        mv_hat, mv_feature = self.mv_dec(s, mv_y_hat, q_dec)
        if me_downsample > 1:                                   # :475-476
            mv_hat = self.K.bilinear_up2(mv_hat, me_downsample) * me_downsample
        self.new_stream()
        be = self.bit_est[s]
        idx = be.build_indexes(mv_z_hat.size())
        self.ec.encode_with_indexes(mv_z_hat.reshape(-1), idx.reshape(-1), *be.cdf_info())
        for q, sc in zip(qw, sw):
            self.gaussian_encode(q, sc)
        bit_stream = self.finish_stream()
        return {"bit_stream": bit_stream, "mv_hat": mv_hat, "mv_feature": mv_feature, "mv_y_hat": mv_y_hat,
                "trace": self.trace}

    # ------------------------------------------------------------------ a10: learned lifting DWT
    def lift_skip(self, p, x):
        """reflectionPadSkip + 3x1 conv, lifting_1d.py:98,105-106"""
        xp = F.pad(x, (0, 0, 1, 1), mode="reflect")
        return self.conv(p, xp)

    def lift_branch(self, wt, name_conv, name_pu, x):
        skip = self.lift_skip(f"{wt}.{name_conv}", x)
        t = self.predict_update(f"{wt}.{name_pu}", skip / 256.0) * 256.0
        return skip + t * 0.1

    def forward_lift(self, wt, x):
        """iWave1D.forward_lift, lifting_1d.py:103-145"""
        x_e, x_o = x[:, :, ::2, :], x[:, :, 1::2, :]
        x_o = x_o + self.lift_branch(wt, "conv_P1", "P_1", x_e)
        x_e = x_e + self.lift_branch(wt, "conv_U1", "U_1", x_o)
        x_o = x_o + self.lift_branch(wt, "conv_P2", "P_2", x_e)
        x_e = x_e + self.lift_branch(wt, "conv_U2", "U_2", x_o)
        x_e = x_e * torch.tensor(1.149604398860241)
        x_o = x_o * torch.tensor(0.869864451624781)
        return x_e, x_o

    def backward_lift(self, wt, l, h):
        """iWave1D.backward_lift, lifting_1d.py:147-189"""
        l = l / torch.tensor(1.149604398860241)
        h = h / torch.tensor(0.869864451624781)
        l = l - self.lift_branch(wt, "conv_U2", "U_2", h)
        h = h - self.lift_branch(wt, "conv_P2", "P_2", l)
        l = l - self.lift_branch(wt, "conv_U1", "U_1", h)
        h = h - self.lift_branch(wt, "conv_P1", "P_1", l)
        dims = list(l.size())
        dims[2] *= 2
        x = torch.zeros(dims)
        x[:, :, ::2, :] = l
        x[:, :, 1::2, :] = h
        return x

    def forward_lift_2d(self, coder, x):
        """LiftingScheme2D.forward_lift_2d, wavelet_transform.py:25-42 (lift_v is lift_h)"""
        wt = f"{coder}.wavelet_transform.lift_h"
        P = lambda t: t.permute((0, 1, 3, 2))
        l, h = self.forward_lift(wt, x)
        ll, lh = self.forward_lift(wt, P(l))
        hl, hh = self.forward_lift(wt, P(h))
        return {"ll": P(ll), "lh": P(lh), "hl": P(hl), "hh": P(hh)}

    def backward_lift_2d(self, coder, sb):
        """wavelet_transform.py:44-57"""
        wt = f"{coder}.wavelet_transform.lift_h"
        P = lambda t: t.permute((0, 1, 3, 2))
        l = P(self.backward_lift(wt, P(sb["ll"]), P(sb["lh"])))
        h = P(self.backward_lift(wt, P(sb["hl"]), P(sb["hh"])))
        return self.backward_lift(wt, l, h)

    # ------------------------------------------------------------------ a11: LL entropy parameters, context_fusion.py:100-128
    def context_fusion_ll(self, coder, x):
        p = f"{coder}.context_fusion.{self.L - 1}.ll"
        x = self.conv(p + ".maskedConv1", x, padding=1)
        conv1 = x
        for i in range(2):
            q = f"{p}.residualBlocks.{i}"
            o = self.conv(q + ".conv1", x, padding=1)
            o = F.leaky_relu(o, 0.2)
            o = self.conv(q + ".conv2", o, padding=1)
            x = o + x
        x = x + conv1
        x = self.conv(p + ".maskedConv2", x, padding=1)
        x = F.leaky_relu(x, 0.2)
        x = F.leaky_relu(self.conv(p + ".convs.0", x), 0.2)
        x = F.leaky_relu(self.conv(p + ".convs.1", x), 0.2)
        return self.conv(p + ".convs.2", x)

    # ------------------------------------------------------------------ a12: four-step context fusion
    def context_residual(self, p, x):
        """ContextResidual, context_fusion_4step.py:9-20"""
        o = self.conv(p + ".conv1", x, padding=1)
        o = F.leaky_relu(o, 0.2)
        o = self.conv(p + ".conv2", o, padding=1)
        return o + x

    def fusion_compress(self, p, x, context, prev_subband):
        """ContextFusionFourStep.forward(write=True), context_fusion_4step.py:139-191"""
        if prev_subband is not None:
            prev = F.interpolate(prev_subband, scale_factor=2, mode="nearest")
            prev = self.conv(p + ".lower_level_subband.1", prev, padding=1)
            context = torch.cat((context, prev), dim=1)
        context = self.conv(p + ".conv1_context", context, padding=1)
        context = self.context_residual(p + ".y_hierarchical_prior_enc.0", context)
        context = self.context_residual(p + ".y_hierarchical_prior_enc.1", context)
        hp = self.depth_conv_block(p + ".y_hierarchical_prior_out", context)
        scales, means = hp.chunk(2, dim=1)
        _, _, H, W = x.size()
        masks = self.masks4(H, W)
        qs, ss = [], []
        _, q, x_hat, sh = self.process_with_mask(x, scales, means, masks[0])
        qs.append(q); ss.append(sh)
        so_far = x_hat
        for step in (1, 2, 3):
            t = self.conv(f"{p}.y_spatial_prior_{step}.0", so_far, padding=1)
            t = self.context_residual(f"{p}.y_spatial_prior_{step}.1", t)
            t = t + context
            t = self.context_residual(f"{p}.y_spatial_prior_{step}_out.0", t)
            t = self.context_residual(f"{p}.y_spatial_prior_{step}_out.1", t)
            params = self.conv(f"{p}.y_spatial_prior_{step}_out.2", t)
            scales, means = params.chunk(2, dim=1)
            _, q, x_hat, sh = self.process_with_mask(x, scales, means, masks[step])
            qs.append(q); ss.append(sh)
            so_far = so_far + x_hat
        return qs, ss, so_far

    # ------------------------------------------------------------------ a13: conv-LSTM subband context
    def lstm(self, p, x, hidden, cell):
        """LSTM2D.forward, long_context.py:16-33"""
        x = self.conv(p + ".conv_in", x, padding=1)
        hidden = self.conv(p + ".conv_hidden", hidden, padding=1)
        x_h = x + hidden
        forget_gate = self.K.sigmoid(x_h)
        input_gate = self.K.sigmoid(x_h)
        c_tilde = self.K.tanh(x_h)
        cell = forget_gate * cell + input_gate * c_tilde
        o = self.K.sigmoid(x_h)
        hidden = o * self.K.tanh(cell)
        return hidden, cell

    def ctx_init(self, size):
        """SubbandContext.init_sequential, long_context.py:156-170"""
        N, _, H, W = size
        self.l3 = [torch.zeros(N, 3, H, W), torch.zeros(N, 1, H, W)]
        self.l1 = [torch.zeros(N, 32, H, W), torch.zeros(N, 32, H, W)]
        self.l2 = [torch.zeros(N, 32, H, W), torch.zeros(N, 32, H, W)]

    def ctx_upsample(self, p, x):
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        return self.conv(p + ".conv", x, padding=1)

    def ctx_forward_one_subband(self, coder, subband, name, lvl):
        """SubbandContext.forward_one_subband, long_context.py:199-224"""
        p = f"{coder}.context_prediction"
        h1, c1 = self.lstm(p + ".LSTM1", subband, *self.l1)
        h2, c2 = self.lstm(p + ".LSTM2", h1, *self.l2)
        h3, c3 = self.lstm(p + ".LSTM3", h2, *self.l3)
        self.l1, self.l2, self.l3 = [h1, c1], [h2, c2], [h3, c3]
        if name == "hh" and lvl > 0:
            self.l1 = [self.ctx_upsample(f"{p}.deconv_h1.{lvl - 1}", self.l1[0]),
                       self.ctx_upsample(f"{p}.deconv_c1.{lvl - 1}", self.l1[1])]
            self.l2 = [self.ctx_upsample(f"{p}.deconv_h2.{lvl - 1}", self.l2[0]),
                       self.ctx_upsample(f"{p}.deconv_c2.{lvl - 1}", self.l2[1])]
            self.l3 = [self.ctx_upsample(f"{p}.deconv_h3.{lvl - 1}", self.l3[0]),
                       self.ctx_upsample(f"{p}.deconv_c3.{lvl - 1}", self.l3[1])]
        return self.l3[0]

    # ------------------------------------------------------------------ a14: PostProcess, postprocessing.py:35-44
    def post_process(self, coder, x):
        p = f"{coder}.dequantModule"
        tmp = self.conv(p + ".conv1", x, padding=1)
        conv1 = tmp
        for i in range(6):
            q = f"{p}.resBlocks.{i}"
            o = self.conv(q + ".conv1", tmp, padding=1)
            o = F.leaky_relu(o, 0.2)
            o = self.conv(q + ".conv2", o, padding=1)
            tmp = o + tmp
        tmp = self.conv(p + ".conv2", tmp, padding=1) + conv1
        tmp = self.conv(p + ".conv3", tmp, padding=1)
        return x + tmp

    # ------------------------------------------------------------------ a9: pWave.compress (skip_decoding=True), pWave.py:381-463
    def pwave_compress(self, coder, x, sideinfo, q_index, qp_scale=None, skip_decoding=True):
        """skip_decoding=True: LL parameters from the one-shot masked network, symbols pushed plane by plane
        (pWave.py:413-418).  skip_decoding=False: the reference codes LL position by position
        (_compress_subband_ar, pWave.py:531-555), i.e. both planes of a position before the next position; the
        parameters are the same causal function of the same symbols, so only the push ORDER differs (N > 1)."""
        _, num_channels, height, width = sideinfo
        q_scale = get_curr_q(self.sd[f"{coder}.QP"], q_index)
        q_scale_ll = get_curr_q(self.sd[f"{coder}.QP_ll"], q_index)
        if qp_scale is not None:
            q_scale = q_scale * qp_scale
            q_scale_ll = q_scale_ll * qp_scale
        clip = 8192.
        # encode(): 4-level DWT (:139-148)
        y = {}
        ll = x
        for lvl in range(self.L):
            y[lvl] = self.forward_lift_2d(coder, ll)
            ll = y[lvl]["ll"]
        self.tap(f"{coder}.ll", ll)
        subbands_hat = {lvl: {} for lvl in range(self.L)}
        ll = (ll * q_scale_ll).clamp(-clip, clip).round()
        self.new_stream()
        params = self.context_fusion_ll(coder, ll)
        scales, means = params.chunk(2, dim=1)
        y_q = torch.round(ll)                   # CompressionModel.process, gaussian_model.py:59-63
        ll_res = y_q - means
        ll_hat = (ll_res.round() + means).round()
        if skip_decoding:
            self.gaussian_encode(ll_res.round(), scales)
        else:
            self.gaussian_encode(ll_res.round().permute(2, 3, 0, 1).contiguous(), scales.permute(2, 3, 0, 1).contiguous())
        subbands_hat[self.L - 1]["ll"] = ll_hat
        self.ctx_init(list(ll.size()))
        context = self.ctx_forward_one_subband(coder, ll_hat, "ll", self.L - 1)
        for lvl in range(self.L - 1, -1, -1):
            for sidx, sb in enumerate(["lh", "hl", "hh"]):
                ctx = context.chunk(3, dim=1)[sidx]
                prev = subbands_hat[lvl + 1][sb] if lvl < self.L - 1 else None
                s_curr = (y[lvl][sb] * q_scale).clamp(-clip, clip)
                qs, ss, s_hat = self.fusion_compress(f"{coder}.context_fusion.{lvl}.{sb}", s_curr, ctx, prev)
                subbands_hat[lvl][sb] = s_hat
                self.tap(f"{coder}.s_hat.{lvl}.{sb}", s_hat)
                for q, sc in zip(qs, ss):
                    self.gaussian_encode(q, sc)
                context = self.ctx_forward_one_subband(coder, s_hat, sb, lvl)
        # dequantize (:191-202) + decode (:150-157)
        rec = {lvl: {} for lvl in range(self.L)}
        for lvl in range(self.L - 1, -1, -1):
            for sb in (["ll", "lh", "hl", "hh"] if lvl == self.L - 1 else ["lh", "hl", "hh"]):
                rec[lvl][sb] = subbands_hat[lvl][sb] / (q_scale_ll if sb == "ll" else q_scale)
        out = None
        for lvl in range(self.L - 1, -1, -1):
            out = self.backward_lift_2d(coder, rec[lvl])
            if lvl > 0:
                rec[lvl - 1]["ll"] = out
        self.tap(f"{coder}.idwt", out)
        x_hat = self.post_process(coder, out / 256.0) * 256.0
        bit_stream = self.finish_stream()
        data = encode_image_bytes(height, width, num_channels, bit_stream)
        return x_hat, data, self.trace

    # ------------------------------------------------------------------ a8: compress_one_stage, pMCTF_L.py:398-420
    def compress_one_stage(self, ref, cur, code_lt, mv_hat, ischroma, sideinfo, stage_idx=0, q_index=0,
                           skip_decoding=True):
        if ischroma:
            mv_hat = self.K.bilinear_down2(mv_hat) / 2
        L_t, H_t, pred, inv = self.forward_MCTF(ref, cur, mv_hat, stage_idx)
        qp_scale = get_curr_q(self.sd[f"hp_q_scale.{stage_idx}"], q_index)
        H_hat, h_bytes, h_trace = self.pwave_compress("hp_coder", H_t, sideinfo, q_index, qp_scale, skip_decoding)
        out = {"L_t": L_t, "H_t": H_t, "H_t_hat": H_hat, "H_bytes": h_bytes, "H_trace": h_trace,
               "L_t_hat": None, "L_bytes": None, "L_trace": None}
        if code_lt:
            L_hat, l_bytes, l_trace = self.pwave_compress("lp_coder", L_t, sideinfo, q_index, None, skip_decoding)
            out.update({"L_t_hat": L_hat, "L_bytes": l_bytes, "L_trace": l_trace})
        return out

    # ------------------------------------------------------------------ a1: encode_one_stage write branch, pMCTF_L.py:553-637
    def encode_one_stage(self, ref_frame, cur_frame, code_lt, dpb, output_path=None, pic_width=None, pic_height=None,
                         psize=128, skip_decoding=True, stage_idx=0, q_index=0, me_downsample=1):
        ref_y, ref_c = ref_frame
        cur_y, cur_c = cur_frame
        mv = self.compress_mv(ref_y, cur_y, dpb, stage_idx=stage_idx, q_index=q_index, me_downsample=me_downsample)
        files = {"mv": encode_p_bytes(mv["bit_stream"], 0)}
        luma = self.compress_one_stage(ref_y, cur_y, code_lt, mv["mv_hat"], False, [1, 1, pic_height, pic_width],
                                       stage_idx, q_index, skip_decoding)
        files["H"] = luma["H_bytes"]
        chroma = self.compress_one_stage(ref_c, cur_c, code_lt, mv["mv_hat"], True,
                                         [1, 2, pic_height // 2, pic_width // 2], stage_idx, q_index, skip_decoding)
        files["Hc"] = chroma["H_bytes"]
        if code_lt:
            files["L"] = luma["L_bytes"]
            files["Lc"] = chroma["L_bytes"]
        if output_path is not None:
            base = os.path.basename(output_path)
            names = {"mv": output_path.replace(".bin", "_mv.bin"), "H": output_path,
                     "Hc": output_path.replace(".bin", "_C_main.bin"),
                     "L": output_path.replace(base, "0_main.bin"), "Lc": output_path.replace(base, "0_C_main.bin")}
            for k, data in files.items():
                with open(names[k], "wb") as f:
                    f.write(data)
        bits = {k: len(v) * 8.0 for k, v in files.items()}
        mv_hat_out, mv_feature_out = mv["mv_hat"], mv["mv_feature"]
        if not skip_decoding:        # pMCTF_L.py:594-612: return what the decoder reconstructs from the files
            dec = self.decode_one_stage(files, code_lt, dpb, pic_height, pic_width, psize, stage_idx, q_index,
                                        me_downsample)
            mv_hat_out, mv_feature_out = dec["mv_hat"], dec["mv_feature"]
            luma = dict(luma, H_t_hat=dec["H_t"], L_t_hat=dec.get("L_t"))
            chroma = dict(chroma, H_t_hat=dec["H_tc"], L_t_hat=dec.get("L_tc"))
        return {
            "L_t": luma["L_t_hat"] if code_lt else luma["L_t"],
            "H_t": luma["H_t_hat"],
            "L_tc": chroma["L_t_hat"] if code_lt else chroma["L_t"],
            "H_tc": chroma["H_t_hat"],
            "bit_H": bits["H"] + bits["Hc"],
            "bit_L": bits["L"] + bits["Lc"] if code_lt else None,
            "bit_Lc": bits["Lc"] if code_lt else None,
            "bit_Hc": bits["Hc"],
            "bit_ME": bits["mv"],
            "mv_hat": mv_hat_out,
            "dpb": {"mv_feature": mv_feature_out, "ref_mv_y": mv["mv_y_hat"]},
            "decoding_time": 0, "encoding_time": 0,        # pMCTF_L.py:588,611 (wall-clock, not part of any comparison)
            "files": files,
            "enc": {"H_t": luma.get("H_t_enc"), "mv_hat": mv["mv_hat"]},
            "traces": {"mv": mv["trace"], "H": luma["H_trace"], "Hc": chroma["H_trace"],
                       "L": luma["L_trace"], "Lc": chroma["L_trace"]},
        }


# ======================================================================================================
# Decoder restatement: decompress_mv, pWave.decompress, sequential LL AR decode,
# four-step / four-part decompress.  Mixed into Oracle below.
# ======================================================================================================
def get_downsampled_shape(height, width, p):
    """stream_helper.py:35-38"""
    new_h = (height + p - 1) // p * p
    new_w = (width + p - 1) // p * p
    return int(new_h / p + 0.5), int(new_w / p + 0.5)


def decode_p_bytes(data):
    """stream_helper.py:189-198"""
    (q,) = struct.unpack(">H", data[:2])
    (n,) = struct.unpack(">I", data[2:6])
    return q, data[6:6 + n]


def decode_image_bytes(data):
    """stream_helper.py:210-220"""
    h, w, c = struct.unpack(">III", data[:12])
    (n,) = struct.unpack(">I", data[12:16])
    return h, w, c, data[16:16 + n]


class _DecoderMixin:
    def gaussian_decode(self, scales):
        """GaussianEncoder.decode_stream, entropy_models.py:280-285"""
        idx = self.K.build_indexes(self.tables, scales)
        val = self.ec.decode_stream(idx.reshape(-1), *self.tables.cdf_info())
        return val.reshape(scales.shape)

    # ---- LL: per-position causal evaluation, context_fusion.py:33-43,140-204 and pWave.py:557-584
    def ll_sequential_decode(self, coder, N, H, W):
        p = f"{coder}.context_fusion.{self.L - 1}.ll"
        sd = self.sd
        nf = 128
        cur = torch.zeros(N, 1, H + 2, W + 2)
        bufs = {k: torch.zeros(N, nf, H + 2, W + 2) for k in ("r0c1", "r0c2", "r1c1", "r1c2", "m2")}
        w1, b1 = sd[p + ".maskedConv1.weight"], sd[p + ".maskedConv1.bias"]

        def conv(x, name):
            # PM-F32: the summation rule of the layer as the ENCODER's one-shot network ran it (on the whole N x H x W plane)
            wt = sd[name + ".weight"]
            if self.K.name == "cdef":
                rule = self.sum_rule(name, torch.empty((N, wt.size(1), H, W), device="meta"), wt)
                return self.K.conv2d(x, wt, sd[name + ".bias"], rule=rule)
            return self.K.conv2d(x, wt, sd[name + ".bias"])

        for h in range(H):
            for w in range(W):
                crop = cur[:, :, h:h + 3, w:w + 3]
                tmp = conv(crop, p + ".maskedConv1")
                conv1 = tmp
                for i in range(2):
                    q = f"{p}.residualBlocks.{i}"
                    x = tmp
                    bufs[f"r{i}c1"][:, :, h + 1:h + 2, w + 1:w + 2] = x
                    t = conv(bufs[f"r{i}c1"][:, :, h:h + 3, w:w + 3], q + ".conv1")
                    t = F.leaky_relu(t, 0.2)
                    bufs[f"r{i}c2"][:, :, h + 1:h + 2, w + 1:w + 2] = t
                    t = conv(bufs[f"r{i}c2"][:, :, h:h + 3, w:w + 3], q + ".conv2")
                    tmp = t + x
                tmp = tmp + conv1
                bufs["m2"][:, :, h + 1:h + 2, w + 1:w + 2] = tmp
                t = conv(bufs["m2"][:, :, h:h + 3, w:w + 3], p + ".maskedConv2")
                t = F.leaky_relu(t, 0.2)
                t = F.leaky_relu(conv(t, p + ".convs.0"), 0.2)
                t = F.leaky_relu(conv(t, p + ".convs.1"), 0.2)
                params = conv(t, p + ".convs.2")
                scale, mean = params.chunk(2, dim=1)
                rec = self.gaussian_decode(scale) + mean
                cur[:, :, h + 1, w + 1] = rec.round()[:, :, 0, 0]
        return cur[:, :, 1:-1, 1:-1].contiguous()

    def fusion_decompress(self, p, context, prev_subband):
        """ContextFusionFourStep.decompress, context_fusion_4step.py:196-249"""
        if prev_subband is not None:
            prev = F.interpolate(prev_subband, scale_factor=2, mode="nearest")
            prev = self.conv(p + ".lower_level_subband.1", prev, padding=1)
            context = torch.cat((context, prev), dim=1)
        context = self.conv(p + ".conv1_context", context, padding=1)
        context = self.context_residual(p + ".y_hierarchical_prior_enc.0", context)
        context = self.context_residual(p + ".y_hierarchical_prior_enc.1", context)
        hp = self.depth_conv_block(p + ".y_hierarchical_prior_out", context)
        scales, means = hp.chunk(2, dim=1)
        _, _, H, W = scales.size()
        masks = self.masks4(H, W)
        q = self.gaussian_decode(scales * masks[0])
        so_far = (q + means) * masks[0]
        for step in (1, 2, 3):
            t = self.conv(f"{p}.y_spatial_prior_{step}.0", so_far, padding=1)
            t = self.context_residual(f"{p}.y_spatial_prior_{step}.1", t)
            t = t + context
            t = self.context_residual(f"{p}.y_spatial_prior_{step}_out.0", t)
            t = self.context_residual(f"{p}.y_spatial_prior_{step}_out.1", t)
            params = self.conv(f"{p}.y_spatial_prior_{step}_out.2", t)
            scales, means = params.chunk(2, dim=1)
            q = self.gaussian_decode(scales * masks[step])
            so_far = so_far + (q + means) * masks[step]
        return so_far

    def pwave_decompress(self, coder, data, padding, q_index, qp_scale=None):
        """pWave.decompress, pWave.py:467-529"""
        q_scale = get_curr_q(self.sd[f"{coder}.QP"], q_index)
        q_scale_ll = get_curr_q(self.sd[f"{coder}.QP_ll"], q_index)
        if qp_scale is not None:
            q_scale = q_scale * qp_scale
            q_scale_ll = q_scale_ll * qp_scale
        height, width, num_channel, bit_stream = decode_image_bytes(data)
        self.ec = entropy.EntropyCoder()
        self.ec.set_stream(bit_stream)
        new_h = (height + padding - 1) // padding * padding
        new_w = (width + padding - 1) // padding * padding
        sh, sw = new_h // (2 ** self.L), new_w // (2 ** self.L)
        ll_rec = self.ll_sequential_decode(coder, num_channel, sh, sw)
        ret = {lvl: {} for lvl in range(self.L)}
        ret[self.L - 1]["ll"] = ll_rec
        self.ctx_init(list(ll_rec.size()))
        context = self.ctx_forward_one_subband(coder, ll_rec, "ll", self.L - 1)
        for lvl in range(self.L - 1, -1, -1):
            for sidx, sb in enumerate(["lh", "hl", "hh"]):
                ctx = context.chunk(3, dim=1)[sidx]
                prev = ret[lvl + 1][sb] if lvl < self.L - 1 else None
                s_hat = self.fusion_decompress(f"{coder}.context_fusion.{lvl}.{sb}", ctx, prev)
                ret[lvl][sb] = s_hat
                context = self.ctx_forward_one_subband(coder, s_hat, sb, lvl)
        rec = {lvl: {} for lvl in range(self.L)}
        for lvl in range(self.L - 1, -1, -1):
            for sb in (["ll", "lh", "hl", "hh"] if lvl == self.L - 1 else ["lh", "hl", "hh"]):
                rec[lvl][sb] = ret[lvl][sb] / (q_scale_ll if sb == "ll" else q_scale)
        out = None
        for lvl in range(self.L - 1, -1, -1):
            out = self.backward_lift_2d(coder, rec[lvl])
            if lvl > 0:
                rec[lvl - 1]["ll"] = out
        return self.post_process(coder, out / 256.0) * 256.0

    def decompress_four_part_prior(self, s, common_params):
        """MVCoderQuad.decompress_four_part_prior, four_part_prior.py:217-280"""
        quant_step, scales, means = common_params.chunk(3, 1)
        quant_step = torch.max(quant_step, torch.ones_like(quant_step) * 0.5)
        _, _, H, W = means.size()
        m = self.masks4(H, W)
        sc = scales.chunk(4, 1)
        mu = means.chunk(4, 1)
        perms = ((0, 1, 2, 3), (3, 2, 1, 0), (2, 3, 0, 1), (1, 0, 3, 2))
        so_far = None
        for t in range(4):
            if t > 0:
                out = self.mv_spatial_prior(s, t, torch.cat((so_far, common_params), dim=1))
                sc, mu = out[:4], out[4:]
            pm = perms[t]
            scales_r = sc[0] * m[pm[0]] + sc[1] * m[pm[1]] + sc[2] * m[pm[2]] + sc[3] * m[pm[3]]
            y_q_r = self.gaussian_decode(scales_r)
            cur = torch.cat([(y_q_r + mu[g]) * m[pm[g]] for g in range(4)], dim=1)
            so_far = cur if so_far is None else so_far + cur
        return so_far * quant_step

    def decompress_mv(self, string, height, width, dpb, stage_idx=0, q_index=0, me_downsample=1):
        """pMCTF.decompress_mv, pMCTF_L.py:497-523 (height/width: size of the plane motion was estimated on)"""
        s = min(self.num_me_stages - 1, stage_idx)
        _, q_dec = self.get_mv_y_q(q_index, s)
        self.ec = entropy.EntropyCoder()
        self.ec.set_stream(string)
        zh, zw = get_downsampled_shape(int(height), int(width), 64)
        be = self.bit_est[s]
        idx = be.build_indexes((1, 64, zh, zw))
        z_hat = self.ec.decode_stream(idx.reshape(-1), *be.cdf_info()).reshape(idx.shape).float()
        mv_params = self.mv_prior_param_decoder(z_hat, dpb, s)
        mv_y_hat = self.decompress_four_part_prior(s, mv_params)
        mv_hat, mv_feature = self.mv_dec(s, mv_y_hat, q_dec)
        if me_downsample > 1:
            mv_hat = self.K.bilinear_up2(mv_hat, me_downsample) * me_downsample
        return {"mv_hat": mv_hat, "mv_feature": mv_feature, "mv_y_hat": mv_y_hat}

    def decode_one_stage(self, files, code_lt, dpb, pic_height, pic_width, psize=128, stage_idx=0, q_index=0,
                         me_downsample=1):
        """decode branch of encode_one_stage (pMCTF_L.py:594-612): files = the dict written by the encoder.
        (With me_downsample > 1 the reference's own branch passes the full-resolution size and no factor to
        decompress_mv, :597-602, which cannot decode; here the motion stream is decoded at the size it was coded at.)"""
        _, string = decode_p_bytes(files["mv"])
        ph = (pic_height + psize - 1) // psize * psize
        pw = (pic_width + psize - 1) // psize * psize
        mv = self.decompress_mv(string, ph // me_downsample, pw // me_downsample, dpb, stage_idx, q_index, me_downsample)
        qp_scale = get_curr_q(self.sd[f"hp_q_scale.{stage_idx}"], q_index)
        out = {"mv_hat": mv["mv_hat"], "mv_feature": mv["mv_feature"], "mv_y_hat": mv["mv_y_hat"],
               "H_t": self.pwave_decompress("hp_coder", files["H"], psize, q_index, qp_scale),
               "H_tc": self.pwave_decompress("hp_coder", files["Hc"], psize // 2, q_index, qp_scale)}
        if code_lt:
            out["L_t"] = self.pwave_decompress("lp_coder", files["L"], psize, q_index)
            out["L_tc"] = self.pwave_decompress("lp_coder", files["Lc"], psize // 2, q_index)
        return out


for _n, _f in list(vars(_DecoderMixin).items()):
    if callable(_f) and not _n.startswith("__"):
        setattr(Oracle, _n, _f)


class _EstimateMixin:
    """Estimate-mode forward: pMCTF.forward_one_stage / pWave.forward_one_channel with Laplace bit
    estimates instead of range coding (pMCTF_L.py:244-295,332-379; pWave.py:231-312; gaussian_model.py:36-67).
    Scalars come back as Python floats: with the ATen back-end they are the reference's own f32 reductions, with the
    PM-F32 back-end f64 sums of per-element f32 values (the product's definition)."""

    def bitparm_params(self, s):
        out = []
        for i in (1, 2, 3, 4):
            pre = f"mv_bit_est.{s}.f{i}."
            a = self.sd.get(pre + "a")
            out.append((F.softplus(self.sd[pre + "h"]), self.sd[pre + "b"], None if a is None else torch.tanh(a)))
        return out

    def mse(self, a, b):
        if self.K.name == "torch":
            return float(F.mse_loss(a, b, reduction="mean"))
        d = a - b
        return self.K.total(d * d) / d.numel()

    def pwave_forward(self, coder, x, q_index=None, qp_scale=None):
        """pWave.forward -> forward_one_channel (pWave.py:231-312)"""
        if q_index is None:
            q_scale, q_scale_ll = self.sd[f"{coder}.QP"][-1], self.sd[f"{coder}.QP_ll"][-1]
        else:
            q_scale = get_curr_q(self.sd[f"{coder}.QP"], q_index)
            q_scale_ll = get_curr_q(self.sd[f"{coder}.QP_ll"], q_index)
            if qp_scale is not None:
                q_scale = q_scale * qp_scale
                q_scale_ll = q_scale_ll * qp_scale
        clip = 8192.
        y = {}
        ll = x
        for lvl in range(self.L):
            y[lvl] = self.forward_lift_2d(coder, ll)
            ll = y[lvl]["ll"]
        subbands_hat = {lvl: {} for lvl in range(self.L)}
        ll = (ll * q_scale_ll).clamp(-clip, clip)
        ll_hat = torch.round(ll)
        params = self.context_fusion_ll(coder, ll_hat)
        scales, means = params.chunk(2, dim=1)
        bits_ll = self.K.laplace_bits(ll_hat - means, scales)
        subbands_hat[self.L - 1]["ll"] = ll_hat
        N = x.size(0)
        if self.K.name == "torch":
            bits_total = torch.sum(bits_ll, dim=(1, 2, 3))
        else:
            bits_total = bits_ll.double().sum(dim=(1, 2, 3))
        self.ctx_init(list(ll.size()))
        context = self.ctx_forward_one_subband(coder, ll_hat, "ll", self.L - 1)
        for lvl in range(self.L - 1, -1, -1):
            for sidx, sb in enumerate(["lh", "hl", "hh"]):
                ctx = context.chunk(3, dim=1)[sidx]
                prev = subbands_hat[lvl + 1][sb] if lvl < self.L - 1 else None
                s_curr = (y[lvl][sb] * q_scale).clamp(-clip, clip)
                qs, ss, s_hat = self.fusion_compress(f"{coder}.context_fusion.{lvl}.{sb}", s_curr, ctx, prev)
                subbands_hat[lvl][sb] = s_hat
                s_q = qs[0] + qs[1] + qs[2] + qs[3]
                sc = ss[0] + ss[1] + ss[2] + ss[3]
                bits = self.K.laplace_bits(s_q, sc)
                if self.K.name == "torch":
                    bits_total += torch.sum(bits, dim=(1, 2, 3))
                else:
                    bits_total = bits_total + bits.double().sum(dim=(1, 2, 3))
                context = self.ctx_forward_one_subband(coder, s_hat, sb, lvl)
        rec = {lvl: {} for lvl in range(self.L)}
        for lvl in range(self.L - 1, -1, -1):
            for sb in (["ll", "lh", "hl", "hh"] if lvl == self.L - 1 else ["lh", "hl", "hh"]):
                rec[lvl][sb] = subbands_hat[lvl][sb] / (q_scale_ll if sb == "ll" else q_scale)
        out = None
        for lvl in range(self.L - 1, -1, -1):
            out = self.backward_lift_2d(coder, rec[lvl])
            if lvl > 0:
                rec[lvl - 1]["ll"] = out
        x_hat = self.post_process(coder, out / 256.0) * 256.0
        tot = bits_total.sum()
        return {"x_hat": x_hat, "subbands": subbands_hat, "bits_per_plane": [float(b) for b in bits_total],
                "bpp_total": float(tot / (x_hat.size(2) * x_hat.size(3) * x_hat.size(0))),
                "bits_total": float(tot / x_hat.size(0)), "mse": self.mse(x, x_hat)}

    def forward_four_part_prior(self, s, y, common_params):
        """MVCoderQuad.forward_four_part_prior(write=False): (y_q, y_hat, scales_hat), four_part_prior.py:89-195"""
        return self.compress_four_part_prior(s, y, common_params, full=True)

    def compute_and_code_motion(self, ref_frame, cur_frame, q_index, dpb, stage_idx=0, me_downsample=1):
        """pMCTF_L.py:244-292 at inference"""
        s = min(self.num_me_stages - 1, stage_idx)
        q_enc = get_curr_q(self.sd[f"mv_y_q_scale_enc.{s}"], q_index)          # not rounded here (inference=False)
        q_dec = get_curr_q(self.sd[f"mv_y_q_scale_dec.{s}"], q_index)
        mv_cur = cur_frame[0, :, :, :].tile((1, 3, 1, 1)) / 255
        mv_ref = ref_frame[0, :, :, :].tile((1, 3, 1, 1)) / 255
        if me_downsample > 1:
            mv_cur = self.K.bilinear_down2(mv_cur, me_downsample)
            mv_ref = self.K.bilinear_down2(mv_ref, me_downsample)
        est_mv = self.spynet(mv_cur, mv_ref)
        mv_y = self.mv_enc(s, est_mv, dpb["mv_feature"], q_enc)
        mv_z = self.mv_hyper_enc(s, mv_y)
        mv_z_hat = torch.round(mv_z)
        mv_params = self.mv_prior_param_decoder(mv_z_hat, dpb, s)
        mv_y_q, mv_y_hat, mv_scales_hat = self.forward_four_part_prior(s, mv_y, mv_params)
        mv_hat, mv_feature = self.mv_dec(s, mv_y_hat, q_dec)
        if me_downsample > 1:
            mv_hat = self.K.bilinear_up2(mv_hat, me_downsample) * me_downsample
        bits_y = self.K.laplace_bits(mv_y_q, mv_scales_hat)
        bits_z = self.K.z_bits(mv_z_hat, self.bitparm_params(s))
        pixel_num = ref_frame.size(2) * ref_frame.size(3)
        if self.K.name == "torch":
            bpp_y = float(torch.sum(torch.sum(bits_y, dim=(1, 2, 3)) / pixel_num))
            bpp_z = float(torch.sum(torch.sum(bits_z, dim=(1, 2, 3)) / pixel_num))
        else:
            bpp_y = self.K.total(bits_y) / pixel_num
            bpp_z = self.K.total(bits_z) / pixel_num
        return mv_hat, {"mv_feature": mv_feature, "mv_y_hat": mv_y_hat}, bpp_y, bpp_z

    def forward_one_stage(self, ref_frame, cur_frame, q_index, code_lt, dpb, mv_hat=None, stage_idx=0, me_downsample=1):
        """pMCTF_L.py:332-379"""
        if mv_hat is not None:
            bpp_y = bpp_z = None
            ref_mv = {"mv_feature": None, "mv_y_hat": None}
            mv_hat = self.K.bilinear_down2(mv_hat) / 2
        else:
            mv_hat, ref_mv, bpp_y, bpp_z = self.compute_and_code_motion(ref_frame, cur_frame, q_index, dpb, stage_idx,
                                                                        me_downsample)
        L_t, H_t, pred, inv = self.forward_MCTF(ref_frame, cur_frame, mv_hat, stage_idx)
        qp_scale = get_curr_q(self.sd[f"hp_q_scale.{stage_idx}"], q_index)
        res_H = self.pwave_forward("hp_coder", H_t, q_index, qp_scale)
        pix = ref_frame.size(2) * ref_frame.size(3)
        f32 = (lambda v: float(np.float32(v))) if self.K.name == "torch" else (lambda v: v)
        ret = {"bpp_mv_y": bpp_y, "bpp_mv_z": bpp_z,
               "bpp_me": None if bpp_z is None else f32(np.float32(bpp_z) + np.float32(bpp_y)) if self.K.name == "torch"
               else bpp_z + bpp_y,
               "me_mse": self.mse(pred, cur_frame), "bpp_H": res_H["bpp_total"], "bit_H": res_H["bits_total"],
               "mse_H": res_H["mse"], "mv_hat": mv_hat,
               "dpb": {"mv_feature": ref_mv["mv_feature"], "ref_mv_y": ref_mv["mv_y_hat"]}, "H_t": res_H["x_hat"]}
        if self.K.name == "torch":
            t = torch.tensor
            bpp = t(res_H["bpp_total"]) if bpp_z is None else t(res_H["bpp_total"]) + t(bpp_z) + t(bpp_y)
            ret["bpp"] = float(bpp)
            ret["bit_ME"] = None if bpp_z is None else float((t(bpp_y) + t(bpp_z)) * pix)
            ret["bit"] = float(bpp * pix)
        else:
            ret["bpp"] = res_H["bpp_total"] if bpp_z is None else res_H["bpp_total"] + bpp_z + bpp_y
            ret["bit_ME"] = None if bpp_z is None else (bpp_y + bpp_z) * pix
            ret["bit"] = ret["bpp"] * pix
        if code_lt:
            res_L = self.pwave_forward("lp_coder", L_t, q_index)
            ret.update({"bpp_L": res_L["bpp_total"], "bit_L": res_L["bits_total"], "mse_L": res_L["mse"],
                        "me_mse_inv": self.mse(inv, ref_frame), "L_t": res_L["x_hat"]})
        else:
            ret["L_t"] = L_t
        return ret


    def estimate_one_stage(self, ref_frame, cur_frame, code_lt, dpb, stage_idx=0, q_index=0, me_downsample=1):
        """the estimate-only branch of encode_one_stage (pMCTF_L.py:530-551): forward_one_stage for luma, then chroma
        with the luma motion.  The reference reads result["mv_feature"] / result["ref_mv_y"] here, keys that
        forward_one_stage does not return (KeyError upstream); the context it was meant to hand on is
        forward_one_stage's "dpb"."""
        ry = self.forward_one_stage(ref_frame[0], cur_frame[0], q_index, code_lt, dpb, stage_idx=stage_idx,
                                    me_downsample=me_downsample)
        rc = self.forward_one_stage(ref_frame[1], cur_frame[1], q_index, code_lt, dpb, mv_hat=ry["mv_hat"],
                                    stage_idx=stage_idx, me_downsample=me_downsample)
        return {"L_t": ry["L_t"], "H_t": ry["H_t"], "L_tc": rc["L_t"], "H_tc": rc["H_t"],
                "bit_L": ry["bit_L"] + rc["bit_L"] if code_lt else None, "bit_H": ry["bit_H"] + rc["bit_H"],
                "bit_Lc": rc["bit_L"] if code_lt else None, "bit_Hc": rc["bit_H"], "bit_ME": ry["bit_ME"],
                "mv_hat": ry["mv_hat"], "dpb": ry["dpb"], "decoding_time": 0, "encoding_time": 0}


for _n, _f in list(vars(_EstimateMixin).items()):
    if callable(_f) and not _n.startswith("__"):
        setattr(Oracle, _n, _f)
