"""ctypes binding of libxfeat_hip.so (declared in include/xfeat_hip.h).

There is exactly one backend.  If the shared library is missing or cannot be loaded this module
raises -- there is no CPU / PyTorch fallback (the product path must fail loudly without the
HIP extension).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("XFH_LIB_PATH") or os.path.join(_HERE, "libxfeat_hip.so")   # override: A/B builds only

XFH_OK = 0
LAYOUT_NCHW, LAYOUT_NHWC = 0, 1
LG_NO_PRUNING = 1 << 30
SAMPLE_MODES = {'nearest': 0, 'bilinear': 1, 'bicubic': 2}
PROF_NONE, PROF_CONV_MFMA, PROF_MATCH, PROF_BLOCK1, PROF_HEADS, PROF_CONV_64_64_S1, PROF_CONV_24_24, PROF_CONV_LAYER0 = 0, 1, 2, 3, 4, 5, 6, 100
PROF_ALL = 1000

# name -> (restype, argtypes); mirrors include/xfeat_hip.h one to one
_p = C.c_void_p
_i = C.c_int
_f = C.c_float
_sz = C.c_size_t
SIGNATURES = {
    "xfh_version": (_i, []),
    "xfh_last_error": (C.c_char_p, []),
    "xfh_num_weight_arrays": (_i, []),
    "xfh_weight_array_floats": (_sz, [_i]),
    "xfh_create": (_i, [C.POINTER(_p), _i, _i, C.POINTER(_p)]),
    "xfh_destroy": (None, [_p]),
    "xfh_resize_bilinear": (_i, [_p, _i, _i, _i, _p, _i, _i, _f, _f, _p]),
    "xfh_backbone_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "xfh_backbone": (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "xfh_backbone_u8": (_i, [_p, _p, _i, _f, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "xfh_backbone_resized": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _f, _f, _i, _i, _f, _f, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "xfh_conv_layer": (_i, [_p, _i, _p, _i, _i, _i, _p, _i, _p]),
    "xfh_detect_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "xfh_detect_sparse": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _f, _i, _i, _f, _f, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "xfh_dense_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "xfh_extract_dense": (_i, [_p, _p, _p, _i, _i, _i, _i, _f, _f, _f, _p, _p, _p, _p, _sz, _p]),
    "xfh_match_workspace_bytes": (_sz, [_i, _i, _i]),
    "xfh_match_mnn": (_i, [_p, _p, _sz, _p, _sz, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p, _p, _p, _p, _sz, _p]),
    "xfh_refine_workspace_bytes": (_sz, [_i, _i]),
    "xfh_refine_matches": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _f, _p, _p, _p, _sz, _p]),
    "xfh_kpts_heatmap": (_i, [_p, _i, _i, _i, _p, _p]),
    "xfh_nms": (_i, [_p, _p, _i, _i, _i, _f, _i, _i, _p, _p, _p, _sz, _p]),
    "xfh_sample_sparse": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p]),
    "xfh_fine_matcher": (_i, [_p, _p, _i, _p, _p, _sz, _p]),
    "xfh_homography_workspace_bytes": (_sz, [_i, _i]),
    "xfh_find_homography": (_i, [_p, _p, _p, _i, _i, _i, C.c_double, _i, C.c_double, C.c_uint64, _p, _p, _p, _p, _sz, _p]),
    "xfh_find_homography_matches": (_i, [_p, _p, _i, _p, _p, _p, _i, _i, C.c_double, _i, C.c_double, C.c_uint64, _p, _p, _p, _p, _sz, _p]),
    "xfh_homography_tables": (_i, [C.c_double, _p, _p, _p]),
    "xfh_lg_num_weight_arrays": (_i, []),
    "xfh_lg_weight_array_floats": (_sz, [_i]),
    "xfh_lg_create": (_i, [C.POINTER(_p), _i, _i, C.POINTER(_p)]),
    "xfh_lg_destroy": (None, [_p]),
    "xfh_lg_workspace_bytes": (_sz, [_i, _i]),
    "xfh_lg_match": (_i, [_p, _p, _p, _i, _f, _f, _p, _p, _i, _f, _f, _f, _i, _p, _p, _p, _p, _sz, _p]),
    "xfh_lg_profile": (_i, [_p, _i]),
    "xfh_lg_profile_read": (_i, [_p, C.POINTER(_i), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "xfh_lg_match_pairs": (_i, [_p, _p, _p, _p, _i, _i, _f, _f, _f, _i, _p, _p, _p, _p, _sz, _p]),
    "xfh_profile_select": (_i, [_p, _i]),
    "xfh_debug_trace": (_i, [_p, _p]),
    "xfh_debug_match_occupancy": (_i, []),
    "xfh_debug_cold_start": (_i, [_i]),
    "xfh_debug_block1": (_i, [_p, _p, _p, _i, _i, _i, _p, _p]),
    "xfh_set_option": (_i, [_p, C.c_char_p, _i]),
    "xfh_get_option": (_i, [_p, C.c_char_p, C.POINTER(_i)]),
    "xfh_set_status_buffer": (_i, [_p, _p]),
    "xfh_profile_read_spans": (_i, [_p, C.POINTER(_i), C.POINTER(C.c_double), _i, C.POINTER(_i)]),
    "xfh_profile_read": (_i, [_p, C.POINTER(_i), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}

_lib = None


class XFeatHipError(RuntimeError):
    pass


def load():
    """Load the shared library (once) and attach the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise XFeatHipError(
            f"{LIB_PATH} not found: build it with `python -m accelerated_features_amd.build` "
            "(hipcc, --offload-arch=gfx950).  There is no fallback backend.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != XFH_OK:
        msg = load().xfh_last_error().decode("utf-8", "replace")
        raise XFeatHipError(f"{what} failed with code {rc}: {msg}")
