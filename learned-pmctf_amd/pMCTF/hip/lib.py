"""ctypes binding of libpmctf_hip.so (C ABI: include/pmctf_hip.h).

The product path has no CPU fallback: if the library is missing or a call is made
without a GPU, an exception is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(os.path.dirname(_HERE))          # learned-pmctf_amd/
LIB_DIR = os.path.join(PKG_ROOT, "lib")
HIP_SO = os.path.join(LIB_DIR, "libpmctf_hip.so")
RANS_SO = os.path.join(LIB_DIR, "libpmctf_rans.so")

_hip = None

vp, ci, cf, i64 = C.c_void_p, C.c_int, C.c_float, C.c_int64

_SIGS = {
    "pmctf_conv2d_packed_size": (i64, [ci] * 4),
    "pmctf_conv2d_packed_bias_size": (i64, [ci]),
    "pmctf_conv2d_pack_weights": (ci, [vp, vp, ci, ci, ci, ci, vp, vp]),
    "pmctf_conv2d_nhwc_f32": (ci, [vp] * 6 + [ci] * 11 + [cf, vp]),
    "pmctf_conv2d_smallcin_f32": (ci, [vp] * 6 + [ci] * 11 + [cf, vp]),
    "pmctf_dwconv2d_nhwc_f32": (ci, [vp] * 4 + [ci] * 5 + [vp]),
    "pmctf_flow_warp_f32": (ci, [vp] * 5 + [ci] * 5 + [cf, vp]),
    "pmctf_avgpool2_f32": (ci, [vp, vp, ci, ci, ci, vp]),
    "pmctf_bilinear_up2_f32": (ci, [vp, vp, ci, ci, ci, cf, vp]),
    "pmctf_bilinear_down2_f32": (ci, [vp, vp, ci, ci, ci, cf, vp]),
}


class NativeLibraryError(RuntimeError):
    pass


def hip():
    """Load libpmctf_hip.so (raises if it has not been built: there is no fallback path)."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_SO):
            raise NativeLibraryError(
                f"{HIP_SO} not found: build it with `make -C {PKG_ROOT}` (or __graft_entry__.build())")
        L = C.CDLL(HIP_SO)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)          # AttributeError if the symbol is missing
            fn.restype, fn.argtypes = res, args
        _hip = L
    return _hip


def exported_symbols():
    return list(_SIGS.keys())


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with status {rc} "
                           f"({'invalid shape/argument' if rc == -1 else 'HIP launch error'})")
