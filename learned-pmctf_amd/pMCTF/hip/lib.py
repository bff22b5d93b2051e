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
    "pmctf_conv2d_set_option": (ci, [C.c_char_p, C.c_long]),
    "pmctf_conv2d_last_launch": (ci, [C.c_char_p, ci]),
    "pmctf_conv2d_get_option": (C.c_long, [C.c_char_p]),
    "pmctf_conv2d_nhwc_f32": (ci, [vp] * 6 + [ci] * 11 + [cf, vp]),
    "pmctf_conv2d_nhwc_geom_f32": (ci, [vp] * 6 + [ci] * 13 + [cf, vp]),
    "pmctf_conv2d_nhwc_opts_f32": (ci, [vp] * 6 + [ci] * 11 + [cf, ci, vp, vp]),
    "pmctf_conv2d_nhwc_geom_opts_f32": (ci, [vp] * 6 + [ci] * 13 + [cf, ci, vp, vp]),
    "pmctf_conv2d_smallcin_f32": (ci, [vp] * 6 + [ci] * 11 + [cf, ci, vp]),
    "pmctf_conv3x3_cin1_dual_f32": (ci, [vp] * 5 + [ci] * 5 + [cf, ci, vp]),
    "pmctf_conv2d_fewcout_supported": (ci, [ci] * 3),
    "pmctf_conv2d_fewcout_f32": (ci, [vp] * 6 + [ci] * 7 + [cf, ci, vp]),
    "pmctf_fourstep_estimate_f32": (ci, [vp] * 3 + [ci] * 5 + [vp, vp]),
    "pmctf_ll_estimate_f32": (ci, [vp, vp, ci, i64, vp, vp]),
    "pmctf_z_estimate_f32": (ci, [vp, vp, vp, i64, ci, vp, vp]),
    "pmctf_mv_fourpart_estimate_f32": (ci, [vp] * 4 + [ci] * 3 + [vp, vp]),
    "pmctf_sqdiff_sum_f32": (ci, [vp, vp, i64, vp, vp]),
    "pmctf_dwconv2d_nhwc_f32": (ci, [vp] * 4 + [ci] * 5 + [vp]),
    "pmctf_flow_warp_f32": (ci, [vp] * 5 + [ci] * 5 + [cf, vp]),
    "pmctf_avgpool2_f32": (ci, [vp, vp, ci, ci, ci, vp]),
    "pmctf_bilinear_up2_f32": (ci, [vp, vp, ci, ci, ci, cf, vp]),
    "pmctf_bilinear_down2_f32": (ci, [vp, vp, ci, ci, ci, cf, vp]),
    "pmctf_bilinear_up_f32": (ci, [vp, vp, ci, ci, ci, ci, cf, vp]),
    "pmctf_bilinear_down_f32": (ci, [vp, vp, ci, ci, ci, ci, cf, vp]),
    "pmctf_ew_f32": (ci, [ci, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, cf, cf, ci, vp]),
    "pmctf_spynet_pack8_f32": (ci, [vp, vp, vp, vp, ci, ci, vp]),
    "pmctf_conv3x3_split_supported": (ci, [ci, ci]),
    "pmctf_conv3x3_split_packed_size": (i64, [ci, ci, ci]),
    "pmctf_conv3x3_split_pack_weights": (ci, [vp, vp, ci, ci, ci, vp, vp]),
    "pmctf_conv3x3_split_f32": (ci, [vp] * 6 + [ci] * 7 + [cf, vp]),
    "pmctf_conv3x3_split_geom_f32": (ci, [vp] * 6 + [ci] * 12 + [cf, vp]),
    "pmctf_predict_update_fused_f32": (ci, [vp] * 11 + [ci] * 4 + [cf] * 6 + [ci, ci, vp]),
    "pmctf_lift_skip3_f32": (ci, [vp, vp, ci, ci, ci, cf, cf, cf, cf, ci, vp]),
    "pmctf_nearest_up2_nhwc_f32": (ci, [vp, vp, ci, ci, ci, ci, vp]),
    "pmctf_pixel_shuffle2_nhwc_f32": (ci, [vp, vp, ci, ci, ci, ci, ci, cf, vp]),
    "pmctf_ffn3_mix_f32": (ci, [vp, vp, i64, ci, vp]),
    "pmctf_lstm_gates_f32": (ci, [vp, vp, vp, vp, i64, ci, ci, vp]),
    "pmctf_lstm_gates_aten_f32": (ci, [vp, vp, vp, vp, i64, ci, ci, i64, ci, ci, vp]),
    "pmctf_fourstep_quant_f32": (ci, [vp] * 5 + [ci, ci, ci, ci, ci, cf, cf, vp]),
    "pmctf_ll_quant_f32": (ci, [vp] * 5 + [i64, ci, cf, cf, vp]),
    "pmctf_ll_ar_packed_size": (i64, []),
    "pmctf_ll_ar_pack_weights": (ci, [vp] * 11),
    "pmctf_ll_ar_scratch_floats": (i64, [ci, ci, ci]),
    "pmctf_ll_ar_decode_f32": (ci, [vp, vp, i64, C.c_uint64, i64, vp, vp, vp, ci, cf, cf, vp, vp, ci, ci, ci, vp, vp]),
    "pmctf_ll_ar_decode_rules_f32": (ci, [vp, vp, i64, C.c_uint64, i64, vp, vp, vp, ci, cf, cf, vp, vp, ci, ci, ci, vp, ci, ci,
                                          ci, vp]),
    "pmctf_fourstep_indexes_f32": (ci, [vp, vp, ci, ci, ci, ci, ci, cf, cf, vp]),
    "pmctf_fourstep_dequant_f32": (ci, [vp, vp, vp, ci, ci, ci, ci, ci, vp]),
    "pmctf_mv_fourpart_indexes_f32": (ci, [vp, vp, vp, ci, ci, ci, cf, cf, vp]),
    "pmctf_mv_fourpart_dequant_f32": (ci, [vp, vp, vp, vp, ci, ci, ci, vp]),
    "pmctf_sym_to_nhwc_f32": (ci, [vp, vp, ci, ci, vp]),
    "pmctf_z_symbols_f32": (ci, [vp] * 4 + [ci, ci, vp]),
    "pmctf_mv_fourpart_step_f32": (ci, [vp] * 6 + [ci, ci, ci, cf, cf, vp]),
    "pmctf_mv_dequant_f32": (ci, [vp, vp, vp, i64, vp]),
}


_RANS_SIGS = {
    "pmctf_rans_encoder_create": (vp, [ci, ci]),
    "pmctf_rans_encoder_destroy": (None, [vp]),
    "pmctf_rans_encoder_reset": (ci, [vp]),
    "pmctf_rans_encoder_encode_with_indexes": (ci, [vp, vp, vp, i64, vp, ci, ci, vp, vp]),
    "pmctf_rans_encoder_set_borrow": (ci, [vp, ci]),
    "pmctf_rans_encoder_flush": (ci, [vp]),
    "pmctf_rans_encoder_stream_size": (i64, [vp]),
    "pmctf_rans_encoder_get_encoded_stream": (ci, [vp, vp, i64]),
    "pmctf_rans_encoder_write_file": (i64, [vp, vp, i64, C.c_char_p]),
    "pmctf_rans_decoder_create": (vp, [ci]),
    "pmctf_rans_decoder_destroy": (None, [vp]),
    "pmctf_rans_decoder_set_stream": (ci, [vp, vp, i64]),
    "pmctf_rans_decoder_decode_stream": (ci, [vp, vp, i64, vp, ci, ci, vp, vp, vp]),
    "pmctf_pmf_to_quantized_cdf": (ci, [vp, ci, ci, vp]),
    "pmctf_rans_decoder_get_state": (ci, [vp, vp, vp]),
    "pmctf_rans_decoder_set_state": (ci, [vp, C.c_uint64, i64]),
}
_rans = None


class NativeLibraryError(RuntimeError):
    pass


def rans():
    """Load libpmctf_rans.so (host range coder, C ABI include/pmctf_rans.h)."""
    global _rans
    if _rans is None:
        if not os.path.exists(RANS_SO):
            raise NativeLibraryError(
                f"{RANS_SO} not found: build it with `make -C {PKG_ROOT}` (or __graft_entry__.build())")
        L = C.CDLL(RANS_SO)
        for name, (res, args) in _RANS_SIGS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _rans = L
    return _rans


def rans_exported_symbols():
    return list(_RANS_SIGS.keys())


def hip():
    """Load libpmctf_hip.so (raises if it has not been built: there is no fallback path)."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_SO):
            raise NativeLibraryError(
                f"{HIP_SO} not found: build it with `make -C {PKG_ROOT}` (or __graft_entry__.build())")
        # torch bundles its own HIP runtime (same SONAME libamdhip64.so.7): it must be the one already loaded
        # when libpmctf_hip.so is mapped, so that kernels launch on the runtime/streams torch manages.
        import torch  # noqa: F401
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
