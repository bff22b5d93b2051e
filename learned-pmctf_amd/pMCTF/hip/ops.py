"""Tensor-level wrappers over the C ABI (include/pmctf_hip.h).

torch is used here only for device memory (torch.empty), the current HIP stream and
host-side weight staging; every numeric operation is a call into libpmctf_hip.so.
Feature maps are NHWC float32 tensors of shape (N, H, W, C); single-channel planes
(N, 1, H, W) alias the same memory.
"""
import contextlib
import ctypes as C
import os
import threading

import numpy as np
import torch

from . import lib as _lib

ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3, 4
SUM_CHAIN, SUM_BLOCKS, SUM_GEMM, SUM_GEMV_3X3 = 0, 1, 2, 3        # PMCTF_SUM_* of include/pmctf_hip.h: the summation rule of a convolution

# Optional live timing of one conv signature with HIP events on the launch stream (bench.py: roofline of the
# dominant kernel).  CONV_PROBE = {"match": fn(conv, x, stride) -> bool, "events": [(start, end, flops)]}
CONV_PROBE = None
# planes below this keep the exact f32 kernels under the reduced-precision profile (too few tiles)
SPLIT_MIN_PX = int(os.environ.get("PMCTF_SPLIT_MIN_PX", "30000"))


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class ConvLaunchOpts(C.Structure):
    """pmctf_conv_launch_opts (include/pmctf_hip.h): launch-shape options handed over WITH a launch; < 0 = process knob"""
    _fields_ = [("split", C.c_long), ("msplit_px", C.c_long)]


_tls = threading.local()


@contextlib.contextmanager
def launch_opts(opts):
    """Convolutions launched by the calling thread inside the block carry `opts` (a ConvLaunchOpts, or None) as their
    per-launch argument.  Nothing process-wide is touched: other host threads keep their own launch shapes."""
    prev = getattr(_tls, "opts", None)
    _tls.opts = opts
    try:
        yield
    finally:
        _tls.opts = prev


def _opts():
    o = getattr(_tls, "opts", None)
    return None if o is None else C.byref(o)


def _p(t):
    if t is None:
        return None
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), "expect dense float32 device tensor"
    return C.c_void_p(t.data_ptr())


def _dev(x):
    if not x.is_cuda:
        raise RuntimeError("pMCTF HIP ops need device tensors: there is no CPU fallback on the product path")
    return x.device


# filters larger than this go to the matrix-core kernel even with one or two couts (see Conv2d.__init__)
FEWCOUT_MAX_K = int(os.environ.get("PMCTF_FEWCOUT_MAX_K", "7"))


class Conv2d:
    """A packed nn.Conv2d (groups=1): weights re-laid out once for the MFMA kernel
    (Cin % 4 == 0) or kept OIHW for the small-Cin vector kernel."""

    def __init__(self, weight, bias, stride=1, padding=(0, 0), device="cuda", split=0, rule=SUM_CHAIN):
        """rule: the layer's summation rule (SUM_CHAIN / SUM_BLOCKS / a reduce-block size B >= 16, include/pmctf_hip.h),
        or a function (N, H, W) -> rule for layers whose rule follows from the reference tensor's shape (1x1 layers,
        pMCTF.hip.aten_rules) — part of the layer's arithmetic, chosen by whoever owns the layer (HipEngine.sum_rule),
        never by the launch shape.
        split = 1, 2 or 3: ALSO pack bf16-split weights for the auxiliary reduced-precision kernel (conv_split.hip) and
        use it on planes of at least SPLIT_MIN_PX output pixels when the shape is supported; 0 (default): exact f32 only."""
        w = weight.detach().to("cpu", torch.float32).contiguous()
        b = None if bias is None else bias.detach().to("cpu", torch.float32).contiguous()
        self.Cout, self.Cin, self.KH, self.KW = w.shape
        self.stride = int(stride)
        self.rule = rule if callable(rule) else int(rule)
        self.pad = (int(padding[0]), int(padding[1])) if isinstance(padding, (tuple, list)) else (int(padding),) * 2
        self.small = self.Cin <= 4
        L = _lib.hip()
        # one or two couts: vector-ALU kernel on plain OIHW weights (a matrix-core tile would be 15/16 empty)
        self.few = (not self.small and self.stride == 1 and self.KH == self.KW and self.pad == (self.KH // 2,) * 2
                    and bool(L.pmctf_conv2d_fewcout_supported(self.Cin, self.Cout, self.KH))
                    and self.KH <= FEWCOUT_MAX_K)
        if self.small or self.few:
            self.w = w.to(device)
            self.b = None if b is None else b.to(device)
        else:
            if self.Cin % 4:
                raise ValueError("MFMA conv path needs Cin % 4 == 0")
            n = L.pmctf_conv2d_packed_size(self.Cout, self.Cin, self.KH, self.KW)
            nb = L.pmctf_conv2d_packed_bias_size(self.Cout)
            wp = np.empty(n, np.float32)
            bp = np.empty(nb, np.float32)
            wn = w.numpy()
            bn = None if b is None else b.numpy()
            _lib.check(L.pmctf_conv2d_pack_weights(wn.ctypes.data, None if bn is None else bn.ctypes.data,
                                                   self.Cout, self.Cin, self.KH, self.KW,
                                                   wp.ctypes.data, bp.ctypes.data), "pack_weights")
            self.w = torch.from_numpy(wp).to(device)
            self.b = torch.from_numpy(bp).to(device)
        self.split = 0
        if (split and not self.small and not self.few and self.stride == 1 and self.KH == 3 and self.KW == 3
                and self.pad == (1, 1) and L.pmctf_conv3x3_split_supported(self.Cin, self.Cout)):
            n16 = L.pmctf_conv3x3_split_packed_size(self.Cout, self.Cin, int(split))
            wp16 = np.empty(n16, np.uint16)
            bp2 = np.empty(L.pmctf_conv2d_packed_bias_size(self.Cout), np.float32)
            wn = w.numpy()
            bn = None if b is None else b.numpy()
            _lib.check(L.pmctf_conv3x3_split_pack_weights(wn.ctypes.data, None if bn is None else bn.ctypes.data,
                                                          self.Cout, self.Cin, int(split), wp16.ctypes.data,
                                                          bp2.ctypes.data), "split pack_weights")
            self.w16 = torch.from_numpy(wp16.view(np.int16)).to(device)
            self.split = int(split)

    def out_shape(self, x):
        N, H, W, Cin = x.shape
        Ho = (H + 2 * self.pad[0] - self.KH) // self.stride + 1
        Wo = (W + 2 * self.pad[1] - self.KW) // self.stride + 1
        return (N, Ho, Wo, self.Cout)

    def __call__(self, x, act=ACT_NONE, slope=0.0, res1=None, res2=None, out=None, rule_hw=None):
        """rule_hw: (H, W) of the plane the REFERENCE evaluates this layer on, when x holds only some of its positions
        (a shape-dependent summation rule follows the reference's call, not this launch's)"""
        _dev(x)
        N, H, W, Cin = x.shape
        if Cin != self.Cin:
            raise ValueError(f"conv expects {self.Cin} input channels, got {Cin}")
        shp = self.out_shape(x)
        y = out if out is not None else torch.empty(shp, dtype=torch.float32, device=x.device)
        assert tuple(y.shape) == shp
        for r in (res1, res2):
            assert r is None or tuple(r.shape) == shp
        L = _lib.hip()
        rule = self.rule(N, *(rule_hw or (H, W))) if callable(self.rule) else self.rule
        if self.few:
            _lib.check(L.pmctf_conv2d_fewcout_f32(_p(x), _p(self.w), _p(self.b), _p(res1), _p(res2), _p(y), N, H, W, Cin,
                                                  self.Cout, self.KH, int(act), float(slope), rule, _stream()),
                       "conv2d_fewcout")
            return y
        probe = CONV_PROBE if (CONV_PROBE is not None and not torch.cuda.is_current_stream_capturing()
                               and CONV_PROBE["match"](self, x, self.stride)) else None
        if probe is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        use_split = self.split and N * H * W >= SPLIT_MIN_PX and int(act) <= ACT_LEAKY
        if use_split:                                       # auxiliary reduced-precision profile (never the parity path)
            _lib.check(L.pmctf_conv3x3_split_f32(_p(x), C.c_void_p(self.w16.data_ptr()), _p(self.b), _p(res1), _p(res2),
                                                 _p(y), N, H, W, Cin, self.Cout, self.split, int(act), float(slope),
                                                 _stream()), "conv3x3_split")
        elif self.small:
            _lib.check(L.pmctf_conv2d_smallcin_f32(_p(x), _p(self.w), _p(self.b), _p(res1), _p(res2), _p(y), N, H, W, Cin,
                                                   self.Cout, self.KH, self.KW, self.stride, self.pad[0], self.pad[1],
                                                   int(act), float(slope), rule, _stream()), "conv2d_smallcin")
        else:
            _lib.check(L.pmctf_conv2d_nhwc_opts_f32(_p(x), _p(self.w), _p(self.b), _p(res1), _p(res2), _p(y), N, H, W, Cin,
                                                    self.Cout, self.KH, self.KW, self.stride, self.pad[0], self.pad[1],
                                                    int(act), float(slope), rule, _opts(), _stream()), "conv2d")
        if probe is not None:
            e1.record()
            probe["events"].append((e0, e1, 2.0 * shp[0] * shp[1] * shp[2] * self.Cout * Cin * self.KH * self.KW))
            if "kernels" in probe and not use_split:
                buf = C.create_string_buffer(512)
                L.pmctf_conv2d_last_launch(buf, 512)
                probe["kernels"][buf.value.decode()] = probe["kernels"].get(buf.value.decode(), 0) + 1
        return y


def conv3x3_cin1_dual(conv, x, act2):
    """conv: a 1->16 3x3 'same' Conv2d; x (N,H,W,1).  Returns (conv(x), act2(conv(x))) from one launch."""
    assert conv.small and conv.Cin == 1 and conv.Cout == 16 and conv.KH == 3 and conv.stride == 1 and conv.pad == (1, 1)
    N, H, W, _ = x.shape
    y = torch.empty((N, H, W, 16), dtype=torch.float32, device=x.device)
    y2 = torch.empty_like(y)
    _lib.check(_lib.hip().pmctf_conv3x3_cin1_dual_f32(_p(x), _p(conv.w), _p(conv.b), _p(y), _p(y2), N, H, W, 16, int(act2),
                                                      0.0, conv.rule, _stream()), "conv3x3_cin1_dual")
    return y, y2


class DepthwiseConv2d:
    def __init__(self, weight, bias, device="cuda"):
        self.C, _, self.K, _ = weight.shape
        self.w = weight.detach().to(device, torch.float32).contiguous()
        self.b = None if bias is None else bias.detach().to(device, torch.float32).contiguous()

    def __call__(self, x):
        N, H, W, Cc = x.shape
        assert Cc == self.C
        y = torch.empty_like(x)
        _lib.check(_lib.hip().pmctf_dwconv2d_nhwc_f32(_p(x), _p(self.w), _p(self.b), _p(y), N, H, W, Cc, self.K,
                                                      _stream()), "dwconv2d")
        return y


def flow_warp(im, flow, lin_x, lin_y, sign=1.0):
    """im (N,C,H,W) planar, flow (1|N,2,H,W) planar."""
    N, Cc, H, W = im.shape
    out = torch.empty_like(im)
    _lib.check(_lib.hip().pmctf_flow_warp_f32(_p(im), _p(flow), _p(lin_x), _p(lin_y), _p(out), N, Cc, H, W,
                                              flow.shape[0], float(sign), _stream()), "flow_warp")
    return out


def avgpool2(x):
    N, Cc, H, W = x.shape
    y = torch.empty((N, Cc, H // 2, W // 2), dtype=torch.float32, device=x.device)
    _lib.check(_lib.hip().pmctf_avgpool2_f32(_p(x), _p(y), N * Cc, H, W, _stream()), "avgpool2")
    return y


def bilinear_up2(x, scale=1.0, factor=2):
    """F.interpolate(bilinear, align_corners=False) to factor x the size (2, 4 or 8), result times `scale`"""
    N, Cc, H, W = x.shape
    y = torch.empty((N, Cc, factor * H, factor * W), dtype=torch.float32, device=x.device)
    _lib.check(_lib.hip().pmctf_bilinear_up_f32(_p(x), _p(y), N * Cc, H, W, int(factor), float(scale), _stream()),
               "bilinear_up")
    return y


def bilinear_down2(x, div=1.0, factor=2):
    """F.interpolate(bilinear, align_corners=False) to 1/factor of the size (2, 4 or 8), result divided by `div`"""
    N, Cc, H, W = x.shape
    y = torch.empty((N, Cc, H // factor, W // factor), dtype=torch.float32, device=x.device)
    _lib.check(_lib.hip().pmctf_bilinear_down_f32(_p(x), _p(y), N * Cc, H, W, int(factor), float(div), _stream()),
               "bilinear_down")
    return y


# ------------------------------------------------------------------------------------------------
# elementwise / layout family (pmctf_ew_f32): operands are logical (N,C,H,W) views with any strides
EW_COPY, EW_ADD, EW_SUB, EW_MUL, EW_DIV, EW_MULS, EW_DIVS, EW_ADD_MULS, EW_SUB_MULS, EW_ADD_MULS_MULS, \
    EW_CLAMP_MULS, EW_ROUND_CLAMP_MULS, EW_ROUND, EW_LEAKY, EW_ADD_MULS2, EW_SUB_MULS2, EW_ROUND_CLAMP, EW_TANH = range(18)

_I64x4 = C.c_int64 * 4


def _raw(t):
    assert t.is_cuda and t.dtype == torch.float32
    return C.c_void_p(t.data_ptr())


def as_nchw(t_nhwc):
    """NHWC storage (N,H,W,C) -> logical NCHW view (no copy)."""
    return t_nhwc.permute(0, 3, 1, 2)


def as_nhwc(t_nchw):
    """logical NCHW view whose storage is channels-last dense -> (N,H,W,C) dense tensor (no copy)."""
    v = t_nchw.permute(0, 2, 3, 1)
    if not v.is_contiguous():
        raise ValueError("tensor is not channels-last dense")
    return v


def empty_planar(n, c, h, w, device):
    return torch.empty((n, c, h, w), dtype=torch.float32, device=device)


def empty_nhwc(n, h, w, c, device):
    return torch.empty((n, h, w, c), dtype=torch.float32, device=device)


def ew(op, a, b=None, alpha=0.0, beta=0.0, out=None):
    """out = op(a, b) over logical (N,C,H,W) views (b may be broadcast with stride 0 via expand)."""
    N, Cc, H, W = a.shape
    if out is None:
        if Cc > 1 and a.stride(1) == 1:
            out = as_nchw(empty_nhwc(N, H, W, Cc, a.device))
        else:
            out = empty_planar(N, Cc, H, W, a.device)
    assert tuple(out.shape) == (N, Cc, H, W), (tuple(out.shape), (N, Cc, H, W))
    if b is not None and tuple(b.shape) != (N, Cc, H, W):
        b = b.expand(N, Cc, H, W)
    so, sa = _I64x4(*out.stride()), _I64x4(*a.stride())
    sb = _I64x4(*b.stride()) if b is not None else None
    cfast = 1 if (Cc > 1 and out.stride(1) == 1) else 0
    _lib.check(_lib.hip().pmctf_ew_f32(int(op), _raw(out), so, _raw(a), sa, None if b is None else _raw(b), sb,
                                       N, Cc, H, W, float(alpha), float(beta), cfast, _stream()), "ew")
    return out


def spynet_pack8(im1, warped, flow_up):
    _, _, H, W = im1.shape
    out = empty_nhwc(1, H, W, 8, im1.device)
    _lib.check(_lib.hip().pmctf_spynet_pack8_f32(_p(im1), _p(warped), _p(flow_up), _p(out), H, W, _stream()),
               "spynet_pack8")
    return out


def lift_skip3(x, w3, bias, rule=SUM_CHAIN):
    N, Cc, H, W = x.shape
    y = torch.empty_like(x)
    _lib.check(_lib.hip().pmctf_lift_skip3_f32(_p(x), _p(y), N * Cc, H, W, float(w3[0]), float(w3[1]), float(w3[2]),
                                               float(bias), int(rule), _stream()), "lift_skip3")
    return y


def predict_update_fused(x, other, pu, mode, c=1.0, sign=1.0, lift=(0.0, 0.0, 0.0, 0.0), skip_rule=SUM_CHAIN):
    """pu: (conv1, conv2, conv3, conv4) Conv2d objects of one PredictUpdate block.  mode 0: (x + PU(x)*0.1)*c;
    mode 1: other + sign * (skip + PU(skip/256)*256*0.1) with skip = reflect 3x1 conv of x (lift = w0, w1, w2, bias;
    skip_rule: summation rule of that 3x1 filter).  The four layers carry their own rule (one for the block)."""
    c1, c2, c3, c4 = pu
    N, Cc, H, W = x.shape
    assert Cc == 1 and c1.small and c4.few and not c2.small and not c2.few
    assert c1.rule == c2.rule == c3.rule == c4.rule
    out = torch.empty_like(x)
    _lib.check(_lib.hip().pmctf_predict_update_fused_f32(
        _p(x), _p(other), _p(out), _p(c1.w), _p(c1.b), _p(c2.w), _p(c2.b), _p(c3.w), _p(c3.b), _p(c4.w), _p(c4.b),
        N, H, W, int(mode), float(c), float(sign), float(lift[0]), float(lift[1]), float(lift[2]), float(lift[3]),
        c1.rule, int(skip_rule), _stream()), "predict_update_fused")
    return out


def nearest_up2(x):
    N, H, W, Cc = x.shape
    y = empty_nhwc(N, 2 * H, 2 * W, Cc, x.device)
    _lib.check(_lib.hip().pmctf_nearest_up2_nhwc_f32(_p(x), _p(y), N, H, W, Cc, _stream()), "nearest_up2")
    return y


def pixel_shuffle2(x, act=ACT_NONE, slope=0.0):
    N, H, W, C4 = x.shape
    assert C4 % 4 == 0
    y = empty_nhwc(N, 2 * H, 2 * W, C4 // 4, x.device)
    _lib.check(_lib.hip().pmctf_pixel_shuffle2_nhwc_f32(_p(x), _p(y), N, H, W, C4 // 4, int(act), float(slope),
                                                        _stream()), "pixel_shuffle2")
    return y


def ffn3_mix(x):
    N, H, W, C2 = x.shape
    y = empty_nhwc(N, H, W, C2 // 2, x.device)
    _lib.check(_lib.hip().pmctf_ffn3_mix_f32(_p(x), _p(y), N * H * W, C2 // 2, _stream()), "ffn3_mix")
    return y


def lstm_gates(xh, cell, ref_planes=None, aten_threads=0):
    """aten_threads > 0: torch.sigmoid as ATen splits the reference's (ref_planes, C, H, W) gate tensor over that many
    threads (include/pmctf_hip.h pmctf_lstm_gates_aten_f32); ref_planes defaults to the batch of xh"""
    N, H, W, Cc = xh.shape
    cell_out, hid_out = torch.empty_like(xh), torch.empty_like(xh)
    _lib.check(_lib.hip().pmctf_lstm_gates_aten_f32(_p(xh), _p(cell), _p(cell_out), _p(hid_out), N * H * W, Cc,
                                                    cell.shape[3], H * W, int(ref_planes or N), int(aten_threads),
                                                    _stream()), "lstm_gates")
    return hid_out, cell_out


def _p16(t, off):
    assert t.is_cuda and t.dtype == torch.int16 and t.is_contiguous()
    return C.c_void_p(t.data_ptr() + 2 * off)


def fourstep_quant(x, params, so_far, sym, idx, off, k, lmin, lstep):
    """params: (N,H,W,2), or (N,H/2,W/2,2) when computed only at the class-k positions"""
    N, _, H, W = x.shape
    sub = 0 if params.shape[1] == H else 1
    assert params.shape[1] * (1 + sub) == H and params.shape[2] * (1 + sub) == W
    _lib.check(_lib.hip().pmctf_fourstep_quant_f32(_p(x), _p(params), _p(so_far), _p16(sym, off), _p16(idx, off),
                                                   N, H, W, k, sub, float(lmin), float(lstep), _stream()),
               "fourstep_quant")


def conv_at_class(conv, x, cls, act=ACT_NONE, slope=0.0, res1=None, res2=None):
    """Evaluate the stride-1 'same' 3x3 conv `conv` only at the positions (2i+py, 2j+px) of parity class cls = 2*py+px:
    the same sums as the full conv at those positions (stride 2, top/left pad 1-py / 1-px, zeros outside)."""
    assert not conv.small and conv.stride == 1 and conv.KH == 3 and conv.KW == 3 and conv.pad == (1, 1)
    N, H, W, Cin = x.shape
    assert H % 2 == 0 and W % 2 == 0 and Cin == conv.Cin
    py, px = cls >> 1, cls & 1
    y = torch.empty((N, H // 2, W // 2, conv.Cout), dtype=torch.float32, device=x.device)
    probe = CONV_PROBE if (CONV_PROBE is not None and not torch.cuda.is_current_stream_capturing()
                           and CONV_PROBE["match"](conv, x, 2)) else None
    if probe is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if conv.split and conv.Cout == 112 and y.shape[0] * y.shape[1] * y.shape[2] >= SPLIT_MIN_PX and int(act) <= ACT_LEAKY:
        _lib.check(_lib.hip().pmctf_conv3x3_split_geom_f32(_p(x), C.c_void_p(conv.w16.data_ptr()), _p(conv.b), _p(res1),
                                                           _p(res2), _p(y), N, H, W, Cin, conv.Cout, conv.split, 2, 1 - py,
                                                           1 - px, H // 2, W // 2, int(act), float(slope), _stream()),
                   "conv3x3_split_geom")
    else:
        _lib.check(_lib.hip().pmctf_conv2d_nhwc_geom_opts_f32(_p(x), _p(conv.w), _p(conv.b), _p(res1), _p(res2), _p(y), N, H,
                                                              W, Cin, conv.Cout, 3, 3, 2, 1 - py, 1 - px, H // 2, W // 2,
                                                              int(act), float(slope), conv.rule, _opts(), _stream()),
                   "conv2d_geom")
    if probe is not None:
        e1.record()
        probe["events"].append((e0, e1, 2.0 * y.numel() * Cin * 9))
    return y


def ll_quant(ll, params, sym, idx, off, lmin, lstep, ar_order=False):
    """ar_order: symbols in the sequential coder's order (position-major over the N planes)"""
    ll_hat = torch.empty_like(ll)
    _lib.check(_lib.hip().pmctf_ll_quant_f32(_p(ll), _p(params), _p(ll_hat), _p16(sym, off), _p16(idx, off),
                                             ll.numel(), ll.shape[0] if ar_order else 0, float(lmin), float(lstep),
                                             _stream()), "ll_quant")
    return ll_hat


def _pi16(t):
    assert t.is_cuda and t.dtype == torch.int16 and t.is_contiguous()
    return C.c_void_p(t.data_ptr())


def fourstep_indexes(params, N, H, W, k, lmin, lstep):
    idx = torch.empty(N * H * W, dtype=torch.int16, device=params.device)
    sub = 0 if params.shape[1] == H else 1
    _lib.check(_lib.hip().pmctf_fourstep_indexes_f32(_p(params), _pi16(idx), N, H, W, k, sub, float(lmin), float(lstep),
                                                     _stream()), "fourstep_indexes")
    return idx


def fourstep_dequant(sym, params, so_far, k):
    N, _, H, W = so_far.shape
    sub = 0 if params.shape[1] == H else 1
    _lib.check(_lib.hip().pmctf_fourstep_dequant_f32(_pi16(sym), _p(params), _p(so_far), N, H, W, k, sub, _stream()),
               "fourstep_dequant")


def mv_fourpart_indexes(common, sp, H, W, t, lmin, lstep):
    idx = torch.empty(16 * H * W, dtype=torch.int16, device=common.device)
    _lib.check(_lib.hip().pmctf_mv_fourpart_indexes_f32(_p(common), _p(sp), _pi16(idx), H, W, t, float(lmin),
                                                        float(lstep), _stream()), "mv_fourpart_indexes")
    return idx


def mv_fourpart_dequant(sym, common, sp, so_far, t):
    _, H, W, _ = so_far.shape
    _lib.check(_lib.hip().pmctf_mv_fourpart_dequant_f32(_pi16(sym), _p(common), _p(sp), _p(so_far), H, W, t, _stream()),
               "mv_fourpart_dequant")


def sym_to_nhwc(sym, H, W, Cc):
    out = torch.empty((1, H, W, Cc), dtype=torch.float32, device=sym.device)
    _lib.check(_lib.hip().pmctf_sym_to_nhwc_f32(_pi16(sym), _p(out), H * W, Cc, _stream()), "sym_to_nhwc")
    return out


def z_symbols(z, sym, idx, off):
    N, H, W, Cc = z.shape
    z_hat = torch.empty_like(z)
    _lib.check(_lib.hip().pmctf_z_symbols_f32(_p(z), _p(z_hat), _p16(sym, off), _p16(idx, off), H * W, Cc, _stream()),
               "z_symbols")
    return z_hat


def mv_fourpart_step(y, common, sp, so_far, sym, idx, off, t, lmin, lstep):
    N, H, W, Cc = y.shape
    assert N == 1 and Cc == 64
    _lib.check(_lib.hip().pmctf_mv_fourpart_step_f32(_p(y), _p(common), _p(sp), _p(so_far), _p16(sym, off),
                                                     _p16(idx, off), H, W, t, float(lmin), float(lstep), _stream()),
               "mv_fourpart_step")


def mv_dequant(so_far, common):
    y_hat = torch.empty_like(so_far)
    _lib.check(_lib.hip().pmctf_mv_dequant_f32(_p(so_far), _p(common), _p(y_hat), so_far.shape[1] * so_far.shape[2],
                                               _stream()), "mv_dequant")
    return y_hat


# ------------------------------------------------------------------------------------------------
# estimate mode (bit estimates / squared errors accumulate into device float64 tensors)
def _pd(t):
    assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
    return C.c_void_p(t.data_ptr())


def fourstep_estimate(x, params, so_far, k, bits):
    N, _, H, W = x.shape
    sub = 0 if params.shape[1] == H else 1
    assert params.shape[1] * (1 + sub) == H and params.shape[2] * (1 + sub) == W and bits.numel() == N
    _lib.check(_lib.hip().pmctf_fourstep_estimate_f32(_p(x), _p(params), _p(so_far), N, H, W, k, sub, _pd(bits),
                                                      _stream()), "fourstep_estimate")


def ll_estimate(ll_hat, params, bits):
    N, _, H, W = ll_hat.shape
    assert bits.numel() == N
    _lib.check(_lib.hip().pmctf_ll_estimate_f32(_p(ll_hat), _p(params), N, H * W, _pd(bits), _stream()), "ll_estimate")


def z_estimate(z, consts, bits):
    N, H, W, Cc = z.shape
    assert N == 1 and tuple(consts.shape) == (11, Cc)
    z_hat = torch.empty_like(z)
    _lib.check(_lib.hip().pmctf_z_estimate_f32(_p(z), _p(z_hat), _p(consts), H * W, Cc, _pd(bits), _stream()), "z_estimate")
    return z_hat


def mv_fourpart_estimate(y, common, sp, so_far, t, bits):
    N, H, W, Cc = y.shape
    assert N == 1 and Cc == 64
    _lib.check(_lib.hip().pmctf_mv_fourpart_estimate_f32(_p(y), _p(common), _p(sp), _p(so_far), H, W, t, _pd(bits),
                                                         _stream()), "mv_fourpart_estimate")


def sqdiff_sum(a, b, acc):
    assert a.shape == b.shape and a.is_contiguous() and b.is_contiguous()
    _lib.check(_lib.hip().pmctf_sqdiff_sum_f32(_p(a), _p(b), a.numel(), _pd(acc), _stream()), "sqdiff_sum")
