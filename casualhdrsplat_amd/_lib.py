"""ctypes binding of libhdrsplat.so -- the C ABI declared in include/hdrsplat.h.

The shared library is built in-tree by `make -C casualhdrsplat_amd/csrc` (hipcc, gfx950) and
is the ONLY compute path of this package: there is no CPU or PyTorch fallback.  If the library
is missing or a symbol cannot be resolved, loading fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HS_LIB_PATH", os.path.join(_HERE, "libhdrsplat.so"))

HS_OK, HS_EINVAL, HS_EHIP, HS_EOVERFLOW = 0, -1, -2, -3
HS_STAGE_PREPROCESS, HS_STAGE_BIN, HS_STAGE_RENDER, HS_STAGE_ALL, HS_STAGE_OFFSETS = 1, 2, 4, 7, 8
HS_STAGE_PREPROCESS_ONLY = 16
HS_FLAG_HDR, HS_FLAG_BLUR_HDR, HS_FLAG_DEBUG, HS_FLAG_ANTIALIAS = 1, 2, 4, 8
HS_FLAG_RADIANCE_EXP, HS_FLAG_RADIANCE_SOFTPLUS = 16, 32
HS_BWD_RENDER, HS_BWD_PREPROCESS, HS_BWD_CRF, HS_BWD_ALL = 1, 2, 4, 7
HS_BWD_SEGSUM, HS_BWD_PROJECT = 8, 16
HS_TILE = 16

_fp = C.c_void_p  # device pointers travel as plain addresses


class hs_dims(C.Structure):
    _fields_ = [("P", C.c_int32), ("M", C.c_int32), ("sh_degree", C.c_int32), ("W", C.c_int32), ("H", C.c_int32),
                ("n_poses", C.c_int32), ("capacity", C.c_int64), ("crf_K", C.c_int32), ("reserved", C.c_int32)]


class hs_sizes(C.Structure):
    _fields_ = [("geom_bytes", C.c_int64), ("binning_bytes", C.c_int64), ("image_bytes", C.c_int64),
                ("bwd_bytes", C.c_int64)]


class hs_counters(C.Structure):
    _fields_ = [("num_rendered", C.c_uint32), ("overflow", C.c_uint32), ("reserved", C.c_uint32 * 6)]


class hs_fwd_args(C.Structure):
    _fields_ = [
        ("dims", hs_dims),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("scale_modifier", C.c_float),
        ("flags", C.c_int32), ("stages", C.c_int32), ("crf_K", C.c_int32),
        ("crf_umin", C.c_float), ("crf_umax", C.c_float),
        ("bg", _fp), ("viewmatrices", _fp), ("projmatrices", _fp), ("camposes", _fp),
        ("means3D", _fp), ("opacities", _fp), ("shs", _fp), ("colors_precomp", _fp), ("scales", _fp),
        ("rotations", _fp), ("cov3D_precomp", _fp), ("exposure", _fp), ("crf_table", _fp),
        ("geom", _fp), ("binning", _fp), ("image", _fp),
        ("out_color", _fp), ("out_hdr", _fp), ("radii", _fp), ("out_invdepth", _fp), ("counters_host", _fp),
    ]


class hs_bwd_args(C.Structure):
    _fields_ = [
        ("dims", hs_dims),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("scale_modifier", C.c_float),
        ("flags", C.c_int32), ("stages", C.c_int32), ("crf_K", C.c_int32), ("crf_umin", C.c_float),
        ("crf_umax", C.c_float),
        ("bg", _fp), ("viewmatrices", _fp), ("projmatrices", _fp), ("camposes", _fp),
        ("means3D", _fp), ("opacities", _fp), ("shs", _fp), ("colors_precomp", _fp), ("scales", _fp),
        ("rotations", _fp), ("cov3D_precomp", _fp), ("exposure", _fp), ("crf_table", _fp),
        ("geom", _fp), ("binning", _fp), ("image", _fp), ("bwd", _fp),
        ("dL_dout_color", _fp), ("dL_dout_hdr", _fp), ("dL_dout_alpha", _fp),
        ("dL_dmeans3D", _fp), ("dL_dmeans2D", _fp), ("dL_dopacities", _fp), ("dL_dshs", _fp),
        ("dL_dcolors_precomp", _fp), ("dL_dscales", _fp), ("dL_drotations", _fp), ("dL_dcov3D_precomp", _fp),
        ("dL_dexposure", _fp), ("dL_dcrf_table", _fp),
        ("dL_dviewmatrices", _fp), ("dL_dprojmatrices", _fp), ("dL_dcamposes", _fp),
        ("dL_dview_colors", _fp), ("dL_dout_invdepth", _fp),
        ("densify_grad_accum", _fp), ("densify_denom", _fp), ("densify_max_radii", _fp),
        ("g_begin", C.c_int32), ("g_end", C.c_int32),
    ]


class hs_layout(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "counters", "rec", "depth", "radii", "tiles_touched", "offsets", "cov3D", "clamped", "scan_spine", "binfo",
        "keys_sorted", "point_list", "pairs_tmp", "ranges", "sort_tmp", "depth_pairs", "inst_sorted", "offs_sorted", "pair_sort_tmp",
        "pair_flags", "pair_act",
        "final_T", "n_contrib", "pose_hdr", "tile_work", "tile_order",
        "pair_grads", "crf_partials", "inst_grads", "pose_partials",
        "tile_matrix", "hier_ws", "depth_ws")]


EXPORTS = ("hs_version", "hs_last_error", "hs_plan", "hs_forward", "hs_backward", "hs_mark_visible",
           "hs_sh_backward_views", "hs_sort_tmp_bytes", "hs_sort_pairs", "hs_render_stats", "hs_sort_tickets", "hs_spline_poses", "hs_depth_sort")
HS_RENDER_STATS = 24

_lib = None


def load() -> C.CDLL:
    """Load libhdrsplat.so; raises (never falls back) when it is absent or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension has not been built "
            "(run `make -C casualhdrsplat_amd/csrc` or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "casualhdrsplat_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise RuntimeError(f"{LIB_PATH} does not export {name}")
    lib.hs_version.restype = C.c_int
    lib.hs_last_error.restype = C.c_char_p
    lib.hs_plan.argtypes = [C.POINTER(hs_dims), C.POINTER(hs_sizes), C.POINTER(hs_layout)]
    lib.hs_plan.restype = C.c_int
    lib.hs_forward.argtypes = [C.POINTER(hs_fwd_args), C.c_void_p]
    lib.hs_forward.restype = C.c_int
    lib.hs_backward.argtypes = [C.POINTER(hs_bwd_args), C.c_void_p]
    lib.hs_backward.restype = C.c_int
    lib.hs_mark_visible.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.hs_mark_visible.restype = C.c_int
    lib.hs_sh_backward_views.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p]
    lib.hs_sh_backward_views.restype = C.c_int
    lib.hs_render_stats.argtypes = [C.POINTER(hs_fwd_args), C.POINTER(hs_bwd_args), C.c_void_p, C.c_void_p, C.c_void_p]
    lib.hs_render_stats.restype = C.c_int
    lib.hs_sort_tmp_bytes.argtypes = [C.c_int64]
    lib.hs_sort_tmp_bytes.restype = C.c_int64
    lib.hs_sort_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                  C.c_void_p, C.c_void_p]
    lib.hs_sort_pairs.restype = C.c_int
    lib.hs_spline_poses.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p]
    lib.hs_spline_poses.restype = C.c_int
    lib.hs_sort_tickets.argtypes = [C.c_int]
    lib.hs_sort_tickets.restype = C.c_int
    lib.hs_depth_sort.argtypes = [C.c_int]
    lib.hs_depth_sort.restype = C.c_int
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != HS_OK:
        msg = load().hs_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def plan(P: int, M: int, sh_degree: int, W: int, H: int, n_poses: int, capacity: int, crf_K: int = 0):
    d = hs_dims(P, M, sh_degree, W, H, n_poses, capacity, crf_K, 0)
    sz, lay = hs_sizes(), hs_layout()
    check(load().hs_plan(C.byref(d), C.byref(sz), C.byref(lay)), "hs_plan")
    return d, sz, lay
