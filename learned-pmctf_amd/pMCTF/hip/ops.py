"""Tensor-level wrappers over the C ABI (include/pmctf_hip.h).

torch is used here only for device memory (torch.empty), the current HIP stream and
host-side weight staging; every numeric operation is a call into libpmctf_hip.so.
Feature maps are NHWC float32 tensors of shape (N, H, W, C); single-channel planes
(N, 1, H, W) alias the same memory.
"""
import ctypes as C

import numpy as np
import torch

from . import lib as _lib

ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3, 4


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    if t is None:
        return None
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), "expect dense float32 device tensor"
    return C.c_void_p(t.data_ptr())


def _dev(x):
    if not x.is_cuda:
        raise RuntimeError("pMCTF HIP ops need device tensors: there is no CPU fallback on the product path")
    return x.device


class Conv2d:
    """A packed nn.Conv2d (groups=1): weights re-laid out once for the MFMA kernel
    (Cin % 4 == 0) or kept OIHW for the small-Cin vector kernel."""

    def __init__(self, weight, bias, stride=1, padding=(0, 0), device="cuda"):
        w = weight.detach().to("cpu", torch.float32).contiguous()
        b = None if bias is None else bias.detach().to("cpu", torch.float32).contiguous()
        self.Cout, self.Cin, self.KH, self.KW = w.shape
        self.stride = int(stride)
        self.pad = (int(padding[0]), int(padding[1])) if isinstance(padding, (tuple, list)) else (int(padding),) * 2
        self.small = self.Cin <= 4
        L = _lib.hip()
        if self.small:
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

    def out_shape(self, x):
        N, H, W, Cin = x.shape
        Ho = (H + 2 * self.pad[0] - self.KH) // self.stride + 1
        Wo = (W + 2 * self.pad[1] - self.KW) // self.stride + 1
        return (N, Ho, Wo, self.Cout)

    def __call__(self, x, act=ACT_NONE, slope=0.0, res1=None, res2=None, out=None):
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
        fn = L.pmctf_conv2d_smallcin_f32 if self.small else L.pmctf_conv2d_nhwc_f32
        _lib.check(fn(_p(x), _p(self.w), _p(self.b), _p(res1), _p(res2), _p(y), N, H, W, Cin, self.Cout,
                      self.KH, self.KW, self.stride, self.pad[0], self.pad[1], int(act), float(slope), _stream()),
                   "conv2d")
        return y


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


def bilinear_up2(x, scale=1.0):
    N, Cc, H, W = x.shape
    y = torch.empty((N, Cc, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
    _lib.check(_lib.hip().pmctf_bilinear_up2_f32(_p(x), _p(y), N * Cc, H, W, float(scale), _stream()), "bilinear_up2")
    return y


def bilinear_down2(x, div=1.0):
    N, Cc, H, W = x.shape
    y = torch.empty((N, Cc, H // 2, W // 2), dtype=torch.float32, device=x.device)
    _lib.check(_lib.hip().pmctf_bilinear_down2_f32(_p(x), _p(y), N * Cc, H, W, float(div), _stream()),
               "bilinear_down2")
    return y
