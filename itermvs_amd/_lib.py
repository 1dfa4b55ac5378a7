"""ctypes binding of libitermvs_hip.so (the C ABI declared in include/itermvs_hip.h).

This is the binding a maintainer of the reference would add next to ``models/module.py`` to
call the HIP path.  Loading fails LOUDLY: there is no CPU or PyTorch
fallback anywhere in the product path.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

MAX_SRC = 16
MAX_HYP = 8
GROUPS = 8
F32, F16, BF16 = 0, 1, 2      # itermvs_dtype: storage type of feature maps
ABI_VERSION = 16

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libitermvs_hip.so")

c_float_p = C.POINTER(C.c_float)


class FMap(C.Structure):
    """itermvs_fmap"""
    _fields_ = [("data", C.c_void_p), ("sb", C.c_int64), ("sc", C.c_int64), ("sy", C.c_int64),
                ("sx", C.c_int64), ("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("dtype", C.c_int32)]


class LevelSrc(C.Structure):
    """itermvs_level_src"""
    _fields_ = [("view", C.c_void_p * MAX_SRC), ("sb", C.c_int64), ("sc", C.c_int64), ("sy", C.c_int64),
                ("sx", C.c_int64), ("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("dtype", C.c_int32)]


class CorrIterParams(C.Structure):
    """itermvs_corr_iter_params"""
    _fields_ = [("B", C.c_int32), ("S", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("N", C.c_int32 * 3), ("impl", C.c_int32),
                ("src", LevelSrc * 3),
                ("ref_q", C.c_void_p), ("proj", C.c_void_p), ("view_w", C.c_void_p),
                ("view_w_sb", C.c_int64), ("view_w_ss", C.c_int64), ("view_w_sp", C.c_int64),
                ("depth", C.c_void_p * 3), ("norm_depth", C.c_void_p), ("norm_depth_sb", C.c_int64),
                ("offsets", (C.c_float * MAX_HYP) * 3),
                ("inv_depth_min", C.c_void_p), ("inv_depth_max", C.c_void_p),
                ("out", C.c_void_p * 3)]


class CorrInitParams(C.Structure):
    """itermvs_corr_init_params"""
    _fields_ = [("B", C.c_int32), ("S", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("N", C.c_int32),
                ("out_layout", C.c_int32),
                ("src", LevelSrc), ("ref", FMap),
                ("proj", C.c_void_p), ("depth", C.c_void_p),
                ("inv_depth_min", C.c_void_p), ("inv_depth_max", C.c_void_p), ("out", C.c_void_p)]


class TapParams(C.Structure):
    """itermvs_tap_params"""
    _fields_ = [("B", C.c_int32), ("S", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("N", C.c_int32),
                ("H1", C.c_int32), ("W1", C.c_int32), ("init", C.c_int32),
                ("proj", C.c_void_p), ("depth", C.c_void_p), ("norm_depth", C.c_void_p), ("norm_depth_sb", C.c_int64),
                ("offsets", C.c_float * MAX_HYP),
                ("inv_depth_min", C.c_void_p), ("inv_depth_max", C.c_void_p), ("out", C.c_void_p), ("coords", C.c_void_p)]


class ConvParams(C.Structure):
    """itermvs_conv_params"""
    _fields_ = [("inp", C.c_void_p), ("out", C.c_void_p), ("out2", C.c_void_p), ("add", C.c_void_p),
                ("aux1", C.c_void_p), ("aux2", C.c_void_p),
                ("in_sn", C.c_int64), ("out_sn", C.c_int64), ("add_sn", C.c_int64), ("aux1_sn", C.c_int64),
                ("aux2_sn", C.c_int64),
                ("weight", C.c_void_p * 3), ("bias", C.c_void_p * 3), ("seg_end", C.c_int32 * 3), ("n_seg", C.c_int32),
                ("N", C.c_int32), ("Cin", C.c_int32), ("Hin", C.c_int32), ("Win", C.c_int32), ("Cout", C.c_int32),
                ("ksize", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("dilation", C.c_int32),
                ("transposed", C.c_int32), ("act", C.c_int32), ("weight_format", C.c_int32), ("add_mode", C.c_int32), ("out_layout", C.c_int32),
                ("split_cout", C.c_int32), ("act_b", C.c_int32), ("in_layout", C.c_int32), ("out_b", C.c_void_p), ("out_b_sn", C.c_int64)]


# name -> (restype, argtypes); every symbol include/itermvs_hip.h declares
PROTOTYPES = {
    "itermvs_version": (C.c_int, []),
    "itermvs_error_string": (C.c_char_p, [C.c_int]),
    "itermvs_compose_proj": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "itermvs_warp": (C.c_int, [C.POINTER(FMap), C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                               C.c_void_p, C.c_void_p, C.c_void_p]),
    "itermvs_warp_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int32] * 7 + [C.c_void_p, C.c_void_p]),
    "itermvs_ref_quarter": (C.c_int, [C.POINTER(FMap)] * 3 + [C.c_int32, C.c_void_p, C.c_void_p]),
    "itermvs_ref_quarter_compose": (C.c_int, [C.POINTER(FMap)] * 3 + [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "itermvs_view_aggregate_up": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                            C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "itermvs_final_upsample": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                         C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                         C.c_void_p]),
    "itermvs_copy_multi": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int32, C.c_void_p]),
    "itermvs_box_probe": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "itermvs_clock_stamp": (C.c_int, [C.c_void_p, C.c_void_p]),
    "itermvs_box_chase": (C.c_int, [C.c_void_p, C.c_uint32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "itermvs_corr_iter": (C.c_int, [C.POINTER(CorrIterParams), C.c_void_p]),
    "itermvs_corr_init": (C.c_int, [C.POINTER(CorrInitParams), C.c_void_p]),
    "itermvs_tap_indices": (C.c_int, [C.POINTER(TapParams), C.c_void_p]),
    "itermvs_corr_iter_backward": (C.c_int, [C.POINTER(CorrIterParams), C.POINTER(C.c_void_p * 3),
                                             C.POINTER(C.POINTER(C.c_void_p) * 3), C.c_void_p, C.c_void_p]),
    "itermvs_corr_init_backward": (C.c_int, [C.POINTER(CorrInitParams), C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p]),
    "itermvs_view_aggregate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                         C.c_void_p, C.c_void_p]),
    "itermvs_pvw_tail": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                   C.c_void_p]),
    "itermvs_softmax_max": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "itermvs_head_fused": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                     C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "itermvs_conv3x3_conv1x1": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                          C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "itermvs_head_fused_conf": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                          C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_int64, C.c_void_p]),
    "itermvs_head_regress": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "itermvs_prob_regress": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p,
                                       C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "itermvs_gru_rh": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                 C.c_int32, C.c_void_p]),
    "itermvs_gru_out": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32,
                                  C.c_int32, C.c_void_p]),
    "itermvs_pack_scores": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_int32, C.c_int32,
                                      C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "itermvs_convex_upsample": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p,
                                          C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    "itermvs_bilinear_up2": (C.c_int, [C.c_void_p] + [C.c_int32] * 6 + [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "itermvs_bilinear_up": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                      C.c_void_p]),
    "itermvs_conv2d": (C.c_int, [C.POINTER(ConvParams), C.c_void_p]),
    "itermvs_corrnet": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32,
                                  C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "itermvs_corrnet_bf16x3": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32,
                                  C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "itermvs_res_chain16": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                      C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p, C.c_int64, C.c_void_p]),
    "itermvs_lateral_conv3x3": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int32,
                                          C.c_void_p, C.c_void_p]),
    "itermvs_gru_conv": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "itermvs_stem": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_int64, C.c_int32, C.c_void_p]),
    "itermvs_stem_compose": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "itermvs_image_pyramid": (C.c_int, [C.c_void_p] + [C.c_int32] * 5 + [C.c_void_p] * 5),
    "itermvs_bn_workspace_floats": (C.c_int, [C.c_int32, C.c_int32, C.c_int32]),
    "itermvs_bn_train_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                           C.c_float, C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p]),
    "itermvs_bn_train_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p]),
    "itermvs_profile_enable": (C.c_int, [C.c_int32]),
    "itermvs_profile_set_mask": (C.c_int, [C.c_int32]),
    "itermvs_fuse_depth": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_double,
                                     C.c_float, C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    "itermvs_profile_graph_count": (C.c_int, []),
    "itermvs_profile_graph_read": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "itermvs_profile_collect": (C.c_int, [C.POINTER(C.c_int32), c_float_p, C.c_int32]),
}


class HipLibraryError(RuntimeError):
    pass


_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """dlopen the in-tree library and type every entry point.  Raises HipLibraryError if the
    library is missing, cannot be loaded, lacks a declared symbol or has another ABI version."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C itermvs_amd/csrc`).  The IterMVS engine has no CPU fallback.")
    # PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so); the library links /opt/rocm's.  Both answer to the
    # same soname, so whichever is loaded FIRST serves the whole process.  Streams and device memory come from torch: its
    # runtime must be the resident one, or the first launch on a torch stream fails (seen as ITERMVS_ERR_LAUNCH when this
    # library was dlopen-ed before `import torch`, e.g. build() followed by smoke() in one process).
    import torch  # noqa: F401
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the box
        raise HipLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    got = lib.itermvs_version()
    if got != ABI_VERSION:
        raise HipLibraryError(f"ABI mismatch: library reports {got}, binding expects {ABI_VERSION}")
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    """Map a non-zero C return to RuntimeError (SURVEY.md 8(b) error convention)."""
    if status != 0:
        msg = load().itermvs_error_string(status).decode()
        raise RuntimeError(f"{what} failed: {msg} (status {status})")
