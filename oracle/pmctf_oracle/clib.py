"""ORACLE — test infrastructure only; never imported by the product path.

ctypes loader for oracle/c/build/libpm_oracle.so (built by oracle/Makefile, or
by __graft_entry__.build()).  Thin numpy wrappers, NCHW float32.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE = os.path.dirname(_HERE)
_SO = os.path.join(_ORACLE, "c", "build", "libpm_oracle.so")
_lib = None

f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
i16p = np.ctypeslib.ndpointer(np.int16, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")


def build():
    subprocess.check_call(["make", "-s", "-C", _ORACLE, "c/build/libpm_oracle.so"])


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        build()
    L = C.CDLL(_SO)
    L.pm_conv2d.argtypes = [f32p, f32p, C.c_void_p, f32p] + [C.c_int] * 10
    L.pm_conv2d_rule.argtypes = [f32p, f32p, C.c_void_p, f32p] + [C.c_int] * 11
    L.pm_dwconv2d.argtypes = [f32p, f32p, C.c_void_p, f32p] + [C.c_int] * 5
    for n in ("pm_tanh_arr", "pm_sigmoid_arr", "pm_log_arr", "pm_exp_arr"):
        getattr(L, n).argtypes = [f32p, f32p, C.c_long]
    L.pm_sigmoid_aten_arr.argtypes = [f32p, f32p, C.c_long, C.c_int]
    L.pm_flow_warp.argtypes = [f32p, f32p, f32p, f32p, f32p] + [C.c_int] * 5
    for n in ("pm_avgpool2", "pm_bilinear_up2", "pm_bilinear_down2"):
        getattr(L, n).argtypes = [f32p, f32p, C.c_int, C.c_int, C.c_int]
    for n in ("pm_bilinear_up", "pm_bilinear_down"):
        getattr(L, n).argtypes = [f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.pm_rans_enc_new.restype = C.c_void_p
    L.pm_rans_enc_free.argtypes = [C.c_void_p]
    L.pm_rans_enc_reset.argtypes = [C.c_void_p]
    L.pm_rans_enc_encode_with_indexes.argtypes = [C.c_void_p, i16p, i16p, C.c_long, i32p, C.c_int, i32p, i32p]
    L.pm_rans_enc_flush.argtypes = [C.c_void_p]
    L.pm_rans_enc_stream_size.argtypes = [C.c_void_p]
    L.pm_rans_enc_stream_size.restype = C.c_long
    L.pm_rans_enc_get_stream.argtypes = [C.c_void_p, u8p]
    L.pm_rans_dec_new.restype = C.c_void_p
    L.pm_rans_dec_free.argtypes = [C.c_void_p]
    L.pm_rans_dec_set_stream.argtypes = [C.c_void_p, u8p, C.c_long]
    L.pm_rans_dec_set_stream.restype = C.c_int
    L.pm_rans_dec_decode_stream.argtypes = [C.c_void_p, i16p, C.c_long, i32p, C.c_int, i32p, i32p, i16p]
    L.pm_rans_menc_new.restype = C.c_void_p
    L.pm_rans_menc_new.argtypes = [C.c_int]
    L.pm_rans_menc_free.argtypes = [C.c_void_p]
    L.pm_rans_menc_reset.argtypes = [C.c_void_p]
    L.pm_rans_menc_encode_with_indexes.argtypes = [C.c_void_p, i16p, i16p, C.c_long, i32p, C.c_int, i32p, i32p]
    L.pm_rans_menc_flush.argtypes = [C.c_void_p]
    L.pm_rans_menc_stream_size.argtypes = [C.c_void_p]
    L.pm_rans_menc_stream_size.restype = C.c_long
    L.pm_rans_menc_get_stream.argtypes = [C.c_void_p, u8p]
    L.pm_rans_mdec_new.restype = C.c_void_p
    L.pm_rans_mdec_new.argtypes = [C.c_int]
    L.pm_rans_mdec_free.argtypes = [C.c_void_p]
    L.pm_rans_mdec_set_stream.argtypes = [C.c_void_p, u8p, C.c_long]
    L.pm_rans_mdec_set_stream.restype = C.c_int
    L.pm_rans_mdec_decode_stream.argtypes = [C.c_void_p, i16p, C.c_long, i32p, C.c_int, i32p, i32p, i16p]
    L.pm_pmf_to_quantized_cdf.argtypes = [f32p, C.c_int, C.c_int, u32p]
    L.pm_pmf_to_quantized_cdf.restype = C.c_int
    _lib = L
    return L


def _c(a, dt=np.float32):
    return np.ascontiguousarray(a, dtype=dt)


def conv2d(x, w, b, stride=1, pad=(0, 0), rule=0):
    x, w = _c(x), _c(w)
    N, Cin, H, W = x.shape
    Cout, Cin2, KH, KW = w.shape
    assert Cin == Cin2
    ph, pw = pad
    Ho = (H + 2 * ph - KH) // stride + 1
    Wo = (W + 2 * pw - KW) // stride + 1
    y = np.empty((N, Cout, Ho, Wo), np.float32)
    bp = None
    if b is not None:
        b = _c(b)
        bp = b.ctypes.data
    lib().pm_conv2d_rule(x, w, bp, y, N, Cin, H, W, Cout, KH, KW, stride, ph, pw, rule)
    return y


def dwconv2d(x, w, b):
    x, w = _c(x), _c(w)
    N, Cc, H, W = x.shape
    K = w.shape[-1]
    y = np.empty_like(x)
    bp = None
    if b is not None:
        b = _c(b)
        bp = b.ctypes.data
    lib().pm_dwconv2d(x, w, bp, y, N, Cc, H, W, K)
    return y


def _map(name, x):
    x = _c(x)
    y = np.empty_like(x)
    getattr(lib(), name)(x.reshape(-1), y.reshape(-1), x.size)
    return y


def tanh(x):
    return _map("pm_tanh_arr", x)


def sigmoid(x, aten_threads=0):
    """aten_threads > 0: x is one contiguous tensor as ATen evaluates it with that many intra-op threads (the scalar tail
    of every thread's slice goes through libm's expf: oracle/c/pm_glibc_expf.h)"""
    if not aten_threads:
        return _map("pm_sigmoid_arr", x)
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty_like(x)
    lib().pm_sigmoid_aten_arr(x.reshape(-1), y.reshape(-1), x.size, int(aten_threads))
    return y


def log(x):
    return _map("pm_log_arr", x)


def exp(x):
    return _map("pm_exp_arr", x)


def flow_warp(im, flow, lin_x, lin_y):
    im, flow = _c(im), _c(flow)
    N, Cc, H, W = im.shape
    out = np.empty_like(im)
    lib().pm_flow_warp(im, flow, _c(lin_x), _c(lin_y), out, N, Cc, H, W, flow.shape[0])
    return out


def avgpool2(x):
    x = _c(x)
    N, Cc, H, W = x.shape
    y = np.empty((N, Cc, H // 2, W // 2), np.float32)
    lib().pm_avgpool2(x, y, N * Cc, H, W)
    return y


def bilinear_up(x, f=2):
    x = _c(x)
    N, Cc, H, W = x.shape
    assert f in (2, 4, 8)
    y = np.empty((N, Cc, f * H, f * W), np.float32)
    lib().pm_bilinear_up(x, y, N * Cc, H, W, f)
    return y


def bilinear_down(x, f=2):
    x = _c(x)
    N, Cc, H, W = x.shape
    assert f in (2, 4, 8) and H >= f and W >= f
    y = np.empty((N, Cc, H // f, W // f), np.float32)
    lib().pm_bilinear_down(x, y, N * Cc, H, W, f)
    return y


def bilinear_up2(x):
    return bilinear_up(x, 2)


def bilinear_down2(x):
    return bilinear_down(x, 2)


def pmf_to_quantized_cdf(pmf, precision=16):
    pmf = _c(pmf)
    cdf = np.zeros(pmf.size + 1, np.uint32)
    rc = lib().pm_pmf_to_quantized_cdf(pmf, pmf.size, precision, cdf)
    if rc != 0:
        raise RuntimeError("pmf_to_quantized_cdf: no frequency to steal")
    return cdf


class RansEncoder:
    """Single-stream counterpart of MLCodec_rans.RansEncoder (py_rans.cpp:227-235)."""

    def __init__(self):
        self._h = lib().pm_rans_enc_new()

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.pm_rans_enc_free(self._h)
            self._h = None

    def reset(self):
        lib().pm_rans_enc_reset(self._h)

    def encode_with_indexes(self, symbols, indexes, cdfs, cdf_sizes, offsets):
        symbols, indexes = _c(symbols, np.int16).reshape(-1), _c(indexes, np.int16).reshape(-1)
        cdfs = _c(cdfs, np.int32)
        lib().pm_rans_enc_encode_with_indexes(self._h, symbols, indexes, symbols.size, cdfs, cdfs.shape[1],
                                              _c(cdf_sizes, np.int32), _c(offsets, np.int32))

    def flush(self):
        lib().pm_rans_enc_flush(self._h)

    def get_encoded_stream(self):
        n = lib().pm_rans_enc_stream_size(self._h)
        out = np.empty(n, np.uint8)
        lib().pm_rans_enc_get_stream(self._h, out)
        return out


class RansDecoder:
    def __init__(self):
        self._h = lib().pm_rans_dec_new()

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.pm_rans_dec_free(self._h)
            self._h = None

    def set_stream(self, stream):
        stream = _c(np.frombuffer(bytes(stream), dtype=np.uint8), np.uint8)
        if lib().pm_rans_dec_set_stream(self._h, stream, stream.size) != 0:
            raise ValueError("bad stream")

    def decode_stream(self, indexes, cdfs, cdf_sizes, offsets):
        indexes = _c(indexes, np.int16).reshape(-1)
        cdfs = _c(cdfs, np.int32)
        out = np.empty(indexes.size, np.int16)
        lib().pm_rans_dec_decode_stream(self._h, indexes, indexes.size, cdfs, cdfs.shape[1],
                                        _c(cdf_sizes, np.int32), _c(offsets, np.int32), out)
        return out


class RansEncoderParts:
    """MLCodec_rans.RansEncoder(False, parts) — the N-part container of py_rans.cpp:29-119 (stream_part > 1)."""

    def __init__(self, parts):
        self._h = lib().pm_rans_menc_new(parts)
        if not self._h:
            raise ValueError("parts must be 1..16")

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.pm_rans_menc_free(self._h)
            self._h = None

    def reset(self):
        lib().pm_rans_menc_reset(self._h)

    def encode_with_indexes(self, symbols, indexes, cdfs, cdf_sizes, offsets):
        symbols, indexes = _c(symbols, np.int16).reshape(-1), _c(indexes, np.int16).reshape(-1)
        cdfs = _c(cdfs, np.int32)
        lib().pm_rans_menc_encode_with_indexes(self._h, symbols, indexes, symbols.size, cdfs, cdfs.shape[1],
                                               _c(cdf_sizes, np.int32), _c(offsets, np.int32))

    def flush(self):
        lib().pm_rans_menc_flush(self._h)

    def get_encoded_stream(self):
        n = lib().pm_rans_menc_stream_size(self._h)
        out = np.empty(n, np.uint8)
        lib().pm_rans_menc_get_stream(self._h, out)
        return out


class RansDecoderParts:
    """MLCodec_rans.RansDecoder(parts), py_rans.cpp:127-224."""

    def __init__(self, parts):
        self._h = lib().pm_rans_mdec_new(parts)
        if not self._h:
            raise ValueError("parts must be 1..16")

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.pm_rans_mdec_free(self._h)
            self._h = None

    def set_stream(self, stream):
        stream = _c(np.frombuffer(bytes(stream), dtype=np.uint8), np.uint8)
        if lib().pm_rans_mdec_set_stream(self._h, stream, stream.size) != 0:
            raise ValueError("bad stream")

    def decode_stream(self, indexes, cdfs, cdf_sizes, offsets):
        indexes = _c(indexes, np.int16).reshape(-1)
        cdfs = _c(cdfs, np.int32)
        out = np.empty(indexes.size, np.int16)
        lib().pm_rans_mdec_decode_stream(self._h, indexes, indexes.size, cdfs, cdfs.shape[1],
                                         _c(cdf_sizes, np.int32), _c(offsets, np.int32), out)
        return out
