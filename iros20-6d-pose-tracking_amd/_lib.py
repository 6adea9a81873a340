"""ctypes binding of libse3tracknet.so (C ABI: include/se3tracknet.h).

The HIP library IS the product path: there is no CPU / PyTorch fallback.  If the shared object
is missing this module raises ImportError telling how to build it; a compute call on a box
without a gfx950 GPU fails in se3tn_create with SE3TN_E_DEVICE / a hipError.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SE3TN_LIB: developer override used for within-session A/B of kernel variants
LIB_PATH = os.environ.get("SE3TN_LIB") or os.path.join(_HERE, "libse3tracknet.so")

NCHW, NHWC = 0, 1
PREC_F32, PREC_F16X3 = 0, 1
WINOGRAD_TILE_AUTO, WINOGRAD_TILE6_MIN_BATCH = 46, 14   # as include/se3tracknet.h: F(4x4) below 14 pairs, F(6x6) from there
WINOGRAD_TILE_6_4, WINOGRAD_HEADS_TILE6_MAX_ROT = 64, 0.2     # F(6x6) 256-channel block + F(4x4) heads; AUTO's rot_normalizer bound for F(6x6) heads
TRUNK_WINOGRAD_DEFAULT_MIN_BATCH, TRUNK_WINOGRAD_DEFAULT_MIN_FILL = 8, 55   # as include/se3tracknet.h (tests/test_host_abi.py)
OFFSET_RULE_NUMPY1, OFFSET_RULE_NUMPY2 = 0, 1   # as include/se3tracknet.h: rounding of OffsetDepth's float64-scalar subtraction
BLUR_NONE, BLUR_BILATERAL, BLUR_GAUSSIAN = 0, 1, 2
RES = 176


class Se3tnError(RuntimeError):
    pass


class Crop(C.Structure):
    """se3tn_crop (include/se3tracknet.h)."""
    _fields_ = [("rgb", C.c_void_p), ("depth", C.c_void_p), ("H", C.c_int32), ("W", C.c_int32),
                ("left", C.c_int32), ("top", C.c_int32), ("right", C.c_int32), ("bottom", C.c_int32),
                ("z_offset_mm", C.c_double), ("stats", C.c_int32), ("_pad", C.c_int32)]


# the same record as a numpy dtype, for building many descriptors without a Python loop
import numpy as _np
CROP_DTYPE = _np.dtype([("rgb", "<u8"), ("depth", "<u8"), ("H", "<i4"), ("W", "<i4"), ("left", "<i4"), ("top", "<i4"),
                        ("right", "<i4"), ("bottom", "<i4"), ("z_offset_mm", "<f8"), ("stats", "<i4"), ("_pad", "<i4")])
assert CROP_DTYPE.itemsize == C.sizeof(Crop)


_SIGS = {
    "se3tn_version": (C.c_char_p, []),
    "se3tn_last_error": (C.c_char_p, []),
    "se3tn_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "se3tn_destroy": (None, [C.c_void_p]),
    "se3tn_max_batch": (C.c_int, [C.c_void_p]),
    "se3tn_reserve": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "se3tn_split_weights_bytes": (C.c_size_t, [C.c_void_p]),
    "se3tn_split_weights_device": (C.c_void_p, [C.c_void_p]),
    "se3tn_split_weights_host": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    "se3tn_set_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int]),
    "se3tn_pack_weights": (C.c_int, [C.c_void_p]),
    "se3tn_packed_bytes": (C.c_size_t, [C.c_void_p]),
    "se3tn_packed_host": (C.c_void_p, [C.c_void_p]),
    "se3tn_upload_weights": (C.c_int, [C.c_void_p, C.c_void_p]),
    "se3tn_bind_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "se3tn_set_normalization": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "se3tn_set_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "se3tn_set_offset_rule": (C.c_int, [C.c_void_p, C.c_int]),
    "se3tn_get_offset_rule": (C.c_int, [C.c_void_p]),
    "se3tn_set_small_kernels": (C.c_int, [C.c_void_p, C.c_int]),
    "se3tn_get_small_kernels": (C.c_int, [C.c_void_p]),
    "se3tn_set_raster_rule": (C.c_int, [C.c_void_p, C.c_int]),
    "se3tn_get_raster_rule": (C.c_int, [C.c_void_p]),
    "se3tn_set_winograd": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "se3tn_get_winograd": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "se3tn_set_trunk_winograd": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "se3tn_get_trunk_winograd": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "se3tn_overflow": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "se3tn_keep_intermediates": (C.c_int, [C.c_void_p, C.c_int]),
    "se3tn_set_normalizers": (C.c_int, [C.c_void_p, C.c_double, C.c_double]),
    "se3tn_preprocess": (C.c_int, [C.c_void_p, C.POINTER(Crop), C.c_int, C.c_void_p, C.c_void_p]),
    "se3tn_crop_raw": (C.c_int, [C.c_void_p, C.POINTER(Crop), C.c_void_p, C.c_void_p, C.c_void_p]),
    "se3tn_input_buffer": (C.c_void_p, [C.c_void_p, C.c_int]),
    "se3tn_infer": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                              C.c_void_p, C.c_void_p, C.c_void_p]),
    "se3tn_enable_graphs": (C.c_int, [C.c_void_p, C.c_int]),
    "se3tn_get_feature": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "se3tn_logits": (C.c_void_p, [C.c_void_p]),
    "se3tn_mesh_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                    C.POINTER(C.c_void_p)]),
    "se3tn_mesh_destroy": (None, [C.c_void_p]),
    "se3tn_mesh_set_texture": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "se3tn_render_frame": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_void_p]),
    "se3tn_on_track": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.c_void_p, C.c_void_p,
                                 C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                 C.POINTER(C.c_int32), C.c_void_p]),
    "se3tn_on_track_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_double, C.c_void_p, C.c_void_p,
                                       C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "se3tn_render": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                               C.POINTER(C.c_int32), C.c_void_p, C.c_void_p, C.c_void_p]),
    "se3tn_fill_depth": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_void_p,
                                   C.c_void_p, C.c_void_p]),
    "se3tn_compute_bbox": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double,
                                     C.POINTER(C.c_int32)]),
    "se3tn_pose_update_host": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                         C.c_double, C.c_double, C.POINTER(C.c_double)]),
    "se3tn_debug_buffer": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int32)]),
    "se3tn_memcpy_d2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "se3tn_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "se3tn_profile_read": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int),
                                     C.POINTER(C.c_float)]),
    "se3tn_profile_launches": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_float)]),
}

_lib = None


def load():
    """Load the HIP library (once).  torch is imported first so that the process keeps a single
    HIP runtime: torch's bundled libamdhip64.so.7 has the same SONAME as /opt/rocm's and the
    dynamic loader reuses the already-loaded one for our NEEDED entry."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise ImportError(
            "HIP extension %s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C %s/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback." % (LIB_PATH, _HERE))
    import torch  # noqa: F401  (loads the HIP runtime this library must share)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError = header/library drift: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def exported_symbols():
    return sorted(_SIGS)


def check(rc, what=""):
    if rc != 0:
        msg = load().se3tn_last_error().decode()
        raise Se3tnError("%s failed (rc=%d): %s" % (what or "se3tn call", rc, msg))
