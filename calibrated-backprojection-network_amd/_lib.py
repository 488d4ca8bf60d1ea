"""ctypes binding of libkbnet_hip.so (the C ABI declared in include/kbnet_hip.h).

`import torch` happens before `ctypes.CDLL` on purpose: torch bundles its own
libamdhip64 with the same SONAME as /opt/rocm's, and the extension must bind to the
HIP runtime that is already loaded so streams and device pointers are shared.
There is NO fallback: a missing library is an error, never a silent CPU path.
"""

from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede CDLL, see module docstring)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KBN_LIB_PATH") or os.path.join(HERE, "libkbnet_hip.so")   # KBN_LIB_PATH: an alternative build (same-box A/B of compile-time variants, tools/ab_lib.sh)

KBN_OK = 0
KBN_ERR_INVALID_ARGUMENT, KBN_ERR_UNSUPPORTED, KBN_ERR_WORKSPACE, KBN_ERR_LAUNCH = -1, -2, -3, -4
KBN_SRC_TENSOR, KBN_SRC_COORDS, KBN_SRC_XYZ, KBN_SRC_PAIR = 0, 1, 2, 3
KBN_ACT_ELU, KBN_ACT_SIGMOID = 1, 2
KBN_RESIZE_NONE, KBN_RESIZE_NEAREST = 0, 1
KBN_MAX_SRC = 3
ABI_VERSION = 7


class KbnError(RuntimeError):
    pass


class ConvSrc(C.Structure):
    """Mirror of `kbn_conv_src` (include/kbnet_hip.h)."""
    _fields_ = [
        ("kind", C.c_int),
        ("channels", C.c_int),
        ("data", C.c_void_p),
        ("batch_stride", C.c_longlong),
        ("src_height", C.c_int),
        ("src_width", C.c_int),
        ("aux_channels", C.c_int),
        ("proj_weight", C.c_void_p),
        ("coordinates", C.c_void_p),
        ("coordinates_batch_stride", C.c_longlong),
        ("kinv", C.c_void_p),
        ("absmax", C.c_void_p),
        ("scale", C.c_void_p),
    ]


_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_longlong, C.c_float

# name -> (restype, argtypes); must list every symbol include/kbnet_hip.h declares
SIGNATURES = {
    "kbn_version": (_I, []),
    "kbn_status_string": (C.c_char_p, [_I]),
    "kbn_reload_env": (None, []),
    "kbn_knob": (_I, [C.c_char_p]),
    "kbn_set_autotune": (_I, [_I]),
    "kbn_get_autotune": (_I, []),
    "kbn_s2d_forward": (_I, [_P, C.POINTER(_P), _P, _P, _I, _I, _I, _I, C.POINTER(_I), _I,
                             C.POINTER(_I), _I, _I, _I, _F, _P]),
    "kbn_s2d_pyramid": (_I, [_P, _L, _P, _I, _I, _I, C.POINTER(_I), _I, C.POINTER(_I), _I, _P]),
    "kbn_intrinsics_inverse": (_I, [_P, _P, _I, _F, _F, _P]),
    "kbn_camera_coordinates": (_I, [_P, _P, _I, _I, _I, _P]),
    "kbn_conv2d_packed_weight_bytes": (C.c_size_t, [_I, _I, _I, _I]),
    "kbn_conv2d_pack_weight": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "kbn_conv2d_forward": (_I, [C.POINTER(ConvSrc), _I, _P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _I,
                                _F, _P, _P]),
    "kbn_upconv2x_packed_weight_bytes": (C.c_size_t, [_I, _I]),
    "kbn_upconv2x_pack_weight": (_I, [_P, _P, _I, _I, _P]),
    "kbn_upconv2x_forward": (_I, [_P, _L, _P, _P, _L, _I, _I, _I, _I, _I, _I, _F, _P, _P]),
    "kbn_activation_forward": (_I, [_P, _L, _I, _L, _I, _P, _P]),
    "kbn_scale_planes_forward": (_I, [_P, _L, _P, _L, _P, _L, _I, _I, _I, _I, _P]),
    "kbn_deconv2x_pack_weight": (_I, [_P, _P, _I, _I, _P]),
    "kbn_deconv2x_forward": (_I, [_P, _L, _P, _P, _L, _I, _I, _I, _I, _I, _I, _F, _P, _P]),
    "kbn_upconv2x_query": (_I, [_I, _I, _I, _I, _I, C.POINTER(_I)]),
    "kbn_conv2d_query": (_I, [_I, _I, _I, _I, _I, _I, _I, _I, C.POINTER(_I)]),
    "kbn_kb_block_forward": (_I, [_P, _L, _P, _L, _P, _P, _P, _L, _P, _P, _P, _P, _P, _L, _P, _L, _P,
                                  _L, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P, _P, _P]),
    "kbn_kb1_front_packed_weight_bytes": (C.c_size_t, [_I, _I, _I]),
    "kbn_kb1_front_pack_weight": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "kbn_kb1_front_query": (_I, [_I, _I, _I, _I, _I, _F]),
    "kbn_kb1_depth_front_query": (_I, [_I, _I, _I, _I, _I, _F]),
    "kbn_kb1_front_forward": (_I, [_P, _L, _P, _P, _L, _P, _L, _P, _L, _I, _I, _I, _I, _I, _I, _F, _F, _P, _P, _P]),
    "kbn_kb1_front_next_packed_weight_bytes": (C.c_size_t, [_I, _I, _I]),
    "kbn_kb1_front_next_pack_weight": (_I, [_P, _P, _I, _I, _I, _P]),
    "kbn_kb1_front_next_query": (_I, [_I, _I, _I, _I, _I, _I, _F]),
    "kbn_kb1_front_next_forward": (_I, [_P, _L, _P, _P, _L, _P, _L, _P, _L, _I, _I, _I, _I, _I, _I, _F, _F, _P, _P,
                                        _P, _P, _L, _P, _L, _I, _F, _P, _P]),
    "kbn_kb1_depth_front_packed_weight_bytes": (C.c_size_t, [_I, _I, _I]),
    "kbn_kb1_depth_front_pack_weight": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "kbn_kb1_depth_front_forward": (_I, [_P, _L, _P, _P, _P, _L, _P, _L, _I, _I, _I, _I, _I, _I, _F, _F, _I, _F, _P, _P]),
    "kbn_s2d_depth_front_packed_weight_bytes": (C.c_size_t, [_I]),
    "kbn_s2d_depth_front_pack_weight": (_I, [_P, _P, _P, _P, _P, _I, _P]),
    "kbn_s2d_depth_front_query": (_I, [_I, C.POINTER(_I), _I, C.POINTER(_I), _I, _I, _I, _I, _I, _I, _I, _F, _F]),
    "kbn_s2d_depth_front_forward": (_I, [_P, _L, _P, _P, _P, _P, _L, _P, _L, _I, _I, C.POINTER(_I), _I, C.POINTER(_I), _I, _I, _I, _I, _I,
                                        _I, _I, _F, _F, _F, _I, _F, _P, _P]),
    "kbn_preprocess_forward": (_I, [_P, _P, _P, _P, _P, _P, C.c_size_t, _I, _I, _I, _I, _I, _F, _I, _P]),
    "kbn_eval_accumulate": (_I, [_P, _P, _P, _P, _I, _I, _I, _F, _F, _P]),
    "kbn_depth_head_forward": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _P]),
    "kbn_conv_head_forward": (_I, [_P, _L, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _F, _P]),
    "kbn_conv_tail_packed_weight_bytes": (C.c_size_t, [_I]),
    "kbn_conv_tail_pack_weight": (_I, [_P, _P, _I, _P]),
    "kbn_conv_tail_forward": (_I, [_P, _L, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _F, _P]),
    "kbn_conv_tail_forward_pair": (_I, [_P, _L, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _F, _P]),
    "kbn_absmax_frames": (_I, [_P, _L, _I, _L, _P, _P]),
    "kbn_conv3x3_split_packed_weight_bytes": (C.c_size_t, [_I, _I, _I]),
    "kbn_conv3x3_split_pack_weight": (_I, [_P, _P, _I, _I, _I, _P]),
    "kbn_conv3x3_split_forward": (_I, [C.POINTER(ConvSrc), _I, _P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P, _L, _P, _P]),
    "kbn_conv3x3_split_forward_ksplit": (_I, [C.POINTER(ConvSrc), _I, _P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _F, _P, _I, _P, _P]),
    "kbn_conv1x1s2_split_packed_weight_bytes": (C.c_size_t, [_I, _I, _I]),
    "kbn_conv1x1s2_split_pack_weight": (_I, [_P, _P, _I, _I, _I, _P]),
    "kbn_conv1x1s2_split_forward": (_I, [C.POINTER(ConvSrc), _I, _P, _P, _L, _P, _L, _I, _I, _I, _I, _I, _I, _F, _P, _P]),
    "kbn_kb_xyz_s2_forward": (_I, [_P, _L, _I, _I, _I, _P, _P, _I, _F, _P, _L, _I, _P]),
    "kbn_png_info": (_I, [_P, C.c_size_t, _P, _P, _P, _P]),
    "kbn_png_decode": (_I, [_P, C.c_size_t, _P, C.c_size_t]),
    "kbn_png_decode_batch": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "kbn_depth_to_u16_forward": (_I, [_P, _P, _L, _P]),
    "kbn_png_encode_gray16_bound": (C.c_size_t, [_I, _I]),
    "kbn_png_encode_gray16": (_I, [_P, _I, _I, _P, C.c_size_t, _P, _I]),
    "kbn_unpack_frames_forward": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
}

_lib = None


def load():
    """Loads the shared library once; raises KbnError when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KbnError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`"
            " (needs hipcc).  There is no CPU fallback for the product path.")
    lib = C.CDLL(LIB_PATH)
    rebuild = "rebuild it with `python -c 'import __graft_entry__ as g; g.build()'`"
    try:
        lib.kbn_version.restype = _I
        lib.kbn_version.argtypes = []
        have = lib.kbn_version()
    except AttributeError:
        raise KbnError(f"{LIB_PATH} exports no kbn_version: not a KBNet HIP library; {rebuild}") from None
    if have != ABI_VERSION:   # checked BEFORE the symbols are bound: a stale library names the mismatch, not a missing symbol
        raise KbnError(f"libkbnet_hip.so ABI {have} != binding ABI {ABI_VERSION}: stale library; {rebuild}")
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise KbnError(f"libkbnet_hip.so (ABI {have}) does not export {name}: header / library mismatch; {rebuild}") from None
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int, what: str):
    if status != KBN_OK:
        msg = load().kbn_status_string(status).decode()
        raise KbnError(f"{what} failed: {msg} (status {status})")
