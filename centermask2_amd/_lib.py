"""ctypes binding of libcmk_hip.so (include/cmk.h).  No fallback: a missing library is an error."""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CMK_LIB") or os.path.join(_HERE, "libcmk_hip.so")      # CMK_LIB: an alternative build (A/B tooling, tools/ab)
_lib = None


class CmkError(RuntimeError):
    pass


class ConvDesc(Structure):
    _fields_ = [
        ("x", c_void_p), ("x_cs", c_int), ("x_co", c_int),
        ("w", c_void_p),
        ("scale", c_void_p), ("shift", c_void_p),
        ("res", c_void_p), ("res_cs", c_int), ("res_co", c_int),
        ("res_mode", c_int), ("Hr", c_int), ("Wr", c_int),
        ("y", c_void_p), ("y_cs", c_int), ("y_co", c_int),
        ("N", c_int), ("H", c_int), ("W", c_int), ("Cin", c_int), ("Cout", c_int),
        ("ksize", c_int), ("stride", c_int), ("relu_upto", c_int), ("in_relu", c_int),
        ("tune_wm", c_int), ("tune_sc", c_int), ("tune_wn", c_int),
        ("w_wino", c_void_p),
        ("in_scale", c_void_p), ("in_shift", c_void_p),
        ("splitk", c_int), ("splitk_ws", c_void_p),
        ("gn_ws", c_void_p), ("gn_groups", c_int),
        ("w_wino6", c_void_p),
        ("pool_ws", c_void_p),
        ("w_split", c_void_p),
        ("w_splith", c_void_p), ("w_splith_scale", c_float),
    ]


class FcosLevel(Structure):
    _fields_ = [("logits", c_void_p), ("regctr", c_void_p), ("H", c_int), ("W", c_int), ("stride", c_int)]


# name -> (restype, argtypes); every symbol include/cmk.h declares
SIGNATURES = {
    "cmk_version": (c_int, []),
    "cmk_arch": (c_char_p, []),
    "cmk_last_error": (c_char_p, []),
    "cmk_conv2d_nhwc": (c_int, [POINTER(ConvDesc), c_void_p]),
    "cmk_conv2d_nhwc_multi": (c_int, [POINTER(ConvDesc), c_int, c_void_p]),
    "cmk_conv_pool_rows": (c_int, [POINTER(ConvDesc)]),
    "cmk_ese_gate_pooled": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "cmk_conv_packed_floats": (c_int64, [c_int, c_int, c_int]),
    "cmk_conv_cout_pad": (c_int, [c_int]),
    "cmk_wino_packed_floats": (c_int64, [c_int, c_int]),
    "cmk_wino6_packed_floats": (c_int64, [c_int, c_int]),
    "cmk_split_packed_halves": (c_int64, [c_int, c_int]),
    "cmk_splith_packed_halves": (c_int64, [c_int, c_int]),
    "cmk_conv_gn_records": (c_int, [c_int, c_int, c_int]),
    "cmk_dwconv3x3_nhwc": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "cmk_stem_conv_nchw3": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "cmk_maxpool3x3s2_ceil_nhwc": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "cmk_maxpool1x1s2_nhwc": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "cmk_ese_gate": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "cmk_ese_scale": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "cmk_groupnorm_relu_nhwc": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "cmk_groupnorm_nhwc": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "cmk_upsample2x_add_nhwc": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "cmk_groupnorm_affine": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "cmk_groupnorm_affine_multi": (c_int, [POINTER(c_void_p), POINTER(c_int), c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float,
                                           POINTER(c_void_p), POINTER(c_void_p), c_void_p]),
    "cmk_groupnorm_affine_tiles": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int), c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_float,
                                           POINTER(c_void_p), POINTER(c_void_p), c_void_p]),
    "cmk_conv_gn_tiles": (c_int, [c_int, c_int]),
    "cmk_fcos_select": (c_int, [POINTER(FcosLevel), c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "cmk_fcos_select_ws_len": (c_int64, [POINTER(FcosLevel), c_int, c_int, c_int]),
    "cmk_nms_topk": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_int,
                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "cmk_roi_align_ratio": (c_int, [POINTER(c_void_p), POINTER(c_int), POINTER(c_int), POINTER(c_float), c_int, c_int, c_int,
                                    c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "cmk_roi_align_pool": (c_int, [POINTER(c_void_p), POINTER(c_int), POINTER(c_int), POINTER(c_float), c_int, c_int, c_int,
                                   c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int,
                                   c_void_p, c_int, c_void_p, c_void_p]),
    "cmk_spatial_attention": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "cmk_mask_predict": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                 c_void_p, c_void_p, c_void_p]),
    "cmk_mask_pool_concat": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "cmk_preprocess_chw": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, POINTER(c_float), POINTER(c_float), c_void_p]),
    "cmk_paste_masks": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "cmk_pack_records": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "cmk_mask_iou_score": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
}


def load():
    """Load the library once; raise CmkError (never fall back) if it is absent or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CmkError(
            "libcmk_hip.so not found at {}: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or centermask2_amd/csrc/build.sh). There is no CPU fallback.".format(LIB_PATH))
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise CmkError("libcmk_hip.so lacks symbol {} declared in include/cmk.h".format(name)) from e
        fn.restype = res
        fn.argtypes = args
    arch = lib.cmk_arch().decode()
    if arch != "gfx950":
        raise CmkError("libcmk_hip.so was built for {}, need gfx950".format(arch))
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise CmkError("{} failed ({}): {}".format(what or "cmk call", rc, load().cmk_last_error().decode()))
