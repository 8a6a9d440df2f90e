"""ctypes wrapper around oracle/hs_oracle.c (the fp32 CPU restatement of the hot path).

TEST INFRASTRUCTURE ONLY -- see the header of hs_oracle.c.  Nothing under
casualhdrsplat_amd/ imports this module; only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg do, and only as the checker / reported baseline.

PARITY UNPINNED: the reference (/root/reference) contains no code, tests or golden
vectors; the rules restated here are those frozen in SURVEY.md
section 8(a).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libhs_oracle.so")
TILE = 16


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (no FMA contraction) if the .so is missing or stale."""
    src = os.path.join(_HERE, "hs_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libhs_oracle.so"])
    return _LIB_PATH


class _Camera(C.Structure):
    _fields_ = [
        ("P", C.c_int), ("sh_degree", C.c_int), ("M", C.c_int), ("W", C.c_int), ("H", C.c_int),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("scale_modifier", C.c_float),
        ("bg", C.c_float * 3), ("viewmatrix", C.c_float * 16), ("projmatrix", C.c_float * 16),
        ("campos", C.c_float * 3), ("antialias", C.c_int), ("radiance_activation", C.c_int),
    ]


RADIANCE_ACTIVATIONS = {"relu_shift": 0, "exp": 1, "softplus": 2}


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.hso_scan.restype = C.c_int64
        _lib.hso_threshold_risk.restype = C.c_int64
        _lib.hso_pixel_reach.restype = C.c_int64
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float32))


@dataclass
class Camera:
    W: int
    H: int
    tanfovx: float
    tanfovy: float
    viewmatrix: np.ndarray  # [4,4] "transposed" convention, flattened row-major
    projmatrix: np.ndarray
    campos: np.ndarray
    bg: np.ndarray
    scale_modifier: float = 1.0
    sh_degree: int = 0
    antialias: bool = False
    radiance_activation: str = "relu_shift"

    def cstruct(self, P: int, M: int) -> _Camera:
        c = _Camera()
        c.P, c.sh_degree, c.M, c.W, c.H = P, self.sh_degree, M, self.W, self.H
        c.tanfovx, c.tanfovy, c.scale_modifier = self.tanfovx, self.tanfovy, self.scale_modifier
        c.bg[:] = [float(v) for v in np.asarray(self.bg, np.float32).reshape(3)]
        c.viewmatrix[:] = [float(v) for v in np.asarray(self.viewmatrix, np.float32).reshape(16)]
        c.projmatrix[:] = [float(v) for v in np.asarray(self.projmatrix, np.float32).reshape(16)]
        c.campos[:] = [float(v) for v in np.asarray(self.campos, np.float32).reshape(3)]
        c.antialias = int(bool(self.antialias))
        c.radiance_activation = RADIANCE_ACTIVATIONS[self.radiance_activation]
        return c

    @property
    def grid(self):
        return (self.W + TILE - 1) // TILE, (self.H + TILE - 1) // TILE


def forward(cam: Camera, means3D, opacities, shs=None, colors_precomp=None, scales=None,
            rotations=None, cov3D_precomp=None) -> dict:
    """Runs a4..a9 and returns every intermediate (all numpy arrays)."""
    L = lib()
    means3D = _f32(means3D)
    P = means3D.shape[0]
    opacities = _f32(opacities).reshape(P)
    shs, colors_precomp = _f32(shs), _f32(colors_precomp)
    scales, rotations, cov3D_precomp = _f32(scales), _f32(rotations), _f32(cov3D_precomp)
    M = shs.shape[1] if shs is not None else 0
    c = cam.cstruct(P, M)
    o = dict(
        depths=np.zeros(P, np.float32), xy=np.zeros((P, 2), np.float32),
        conic_opacity=np.zeros((P, 4), np.float32), rgb=np.zeros((P, 3), np.float32),
        radii=np.zeros(P, np.int32), tiles_touched=np.zeros(P, np.uint32),
        rect=np.zeros((P, 4), np.int32), cov3D=np.zeros((P, 6), np.float32),
        clamped=np.zeros((P, 3), np.uint8),
    )
    rc = L.hso_preprocess_fwd(C.byref(c), _p(means3D), _p(opacities), _p(shs), _p(colors_precomp),
                              _p(scales), _p(rotations), _p(cov3D_precomp), _p(o["depths"]), _p(o["xy"]),
                              _p(o["conic_opacity"]), _p(o["rgb"]), _p(o["radii"]), _p(o["tiles_touched"]),
                              _p(o["rect"]), _p(o["cov3D"]), _p(o["clamped"]))
    assert rc == 0
    o["offsets"] = np.zeros(P, np.uint32)
    R = int(L.hso_scan(_p(o["tiles_touched"]), C.c_int(P), _p(o["offsets"])))
    gx, gy = cam.grid
    ntiles = gx * gy
    o["R"] = R
    o["keys_unsorted"] = np.zeros(R, np.uint64)
    o["vals_unsorted"] = np.zeros(R, np.uint32)
    L.hso_duplicate_with_keys(C.byref(c), _p(o["depths"]), _p(o["rect"]), _p(o["radii"]), _p(o["offsets"]),
                              _p(o["keys_unsorted"]), _p(o["vals_unsorted"]))
    nbits = 32 + int(L.hso_key_tile_bits(C.c_uint32(ntiles)))
    o["sort_bits"] = nbits
    o["keys_sorted"] = np.zeros(R, np.uint64)
    o["point_list"] = np.zeros(R, np.uint32)
    rc = L.hso_sort_pairs(_p(o["keys_unsorted"]), _p(o["vals_unsorted"]), C.c_int64(R), C.c_int(nbits),
                          _p(o["keys_sorted"]), _p(o["point_list"]))
    assert rc == 0
    o["ranges"] = np.zeros((ntiles, 2), np.uint32)
    L.hso_tile_ranges(_p(o["keys_sorted"]), C.c_int64(R), C.c_int(ntiles), _p(o["ranges"]))
    o["color"] = np.zeros((3, cam.H, cam.W), np.float32)
    o["final_T"] = np.zeros((cam.H, cam.W), np.float32)
    o["n_contrib"] = np.zeros((cam.H, cam.W), np.uint32)
    o["invdepth"] = np.zeros((cam.H, cam.W), np.float32)  # expected inverse depth
    L.hso_render_fwd(C.byref(c), _p(o["ranges"]), _p(o["point_list"]), _p(o["xy"]), _p(o["conic_opacity"]),
                     _p(o["rgb"]), _p(o["color"]), _p(o["final_T"]), _p(o["n_contrib"]), _p(o["depths"]),
                     _p(o["invdepth"]))
    o["opacities_in"] = opacities
    return o


def threshold_risk(cam: Camera, fwd: dict, guard_alpha: float = 2e-5, guard_T: float = 1e-4) -> dict:
    """Guard band of the frame `fwd` (forward()'s dict) around the three piecewise-constant decisions of the
    compositing loop (hso_threshold_risk): pixels / Gaussians on which another fp32 implementation may legitimately
    decide differently, and the smallest margins seen.  guard_alpha is relative to 1/255, guard_T relative to 1e-4."""
    L = lib()
    P = fwd["radii"].shape[0]
    c = cam.cstruct(P, 0)
    pix = np.zeros((cam.H, cam.W), np.uint8)
    gs = np.zeros(P, np.uint8)
    mm = np.zeros(3, np.float64)
    n = int(L.hso_threshold_risk(C.byref(c), _p(fwd["ranges"]), _p(fwd["point_list"]), _p(fwd["xy"]),
                                 _p(fwd["conic_opacity"]), C.c_float(guard_alpha), C.c_float(guard_T), _p(pix), _p(gs),
                                 _p(mm)))
    return dict(n_risky_pixels=n, pix_risk=pix.astype(bool), gauss_risk=gs.astype(bool),
                min_margin_alpha=float(mm[0]), min_margin_T=float(mm[1]), min_abs_power=float(mm[2]))


def pixel_reach(cam: Camera, fwd: dict, pix_mask, whole_list: bool = False, guard_alpha: float = 1e-5) -> np.ndarray:
    """bool [P]: the Gaussians that contribute to the pixels selected by `pix_mask` [H,W] (hso_pixel_reach).
    whole_list: every entry of those pixels' tile lists that ANY fp32 implementation may have blended (alpha within
    guard_alpha of the 1/255 threshold or above, no early termination) instead of this oracle's own contributors."""
    L = lib()
    P = fwd["radii"].shape[0]
    c = cam.cstruct(P, 0)
    m = np.ascontiguousarray(np.asarray(pix_mask).astype(np.uint8))
    gs = np.zeros(P, np.uint8)
    L.hso_pixel_reach(C.byref(c), _p(fwd["ranges"]), _p(fwd["point_list"]), _p(fwd["xy"]), _p(fwd["conic_opacity"]), _p(m), _p(gs),
                      C.c_int(int(whole_list)), C.c_float(guard_alpha))
    return gs.astype(bool)


def backward(cam: Camera, fwd: dict, dL_dcolor_img, means3D, shs=None, colors_precomp=None,
             scales=None, rotations=None, cov3D_precomp=None, dL_dinvdepth_img=None, bounds: bool = False) -> dict:
    """Runs a10..a12 given forward()'s intermediates and dL/d(out_color) [3,H,W] (+ optional dL/d(invdepth) [H,W]).

    bounds=True adds, next to every gradient tensor `dL_dX`, `abs_dL_dX` of the same shape = sum |terms| of that
    element, and `n_terms` [P]: the render backward returns, for each of its ten per-Gaussian sums, the sum over the
    pixels of the magnitude of what it added (hso_render_bwd, abs_terms); a11/a12 are linear in those sums, so the
    magnitudes are carried through them by hso_preprocess_bwd_abs -- the same chain with every coefficient's absolute
    value and every difference turned into a sum (a net Jacobian entry that is small by cancellation still carries the
    rounding of the intermediates it cancelled from)."""
    L = lib()
    means3D = _f32(means3D)
    P = means3D.shape[0]
    shs, colors_precomp = _f32(shs), _f32(colors_precomp)
    scales, rotations, cov3D_precomp = _f32(scales), _f32(rotations), _f32(cov3D_precomp)
    M = shs.shape[1] if shs is not None else 0
    c = cam.cstruct(P, M)
    g = _f32(dL_dcolor_img).reshape(3, cam.H, cam.W)
    o = dict(
        dL_dmean2D=np.zeros((P, 2), np.float32), dL_dconic=np.zeros((P, 3), np.float32),
        dL_dopacity=np.zeros(P, np.float32), dL_dcolor=np.zeros((P, 3), np.float32),
        abs_terms=np.zeros((P, 11), np.float32) if bounds else None,
    )
    gd = None if dL_dinvdepth_img is None else _f32(dL_dinvdepth_img).reshape(cam.H, cam.W)
    o["dL_dinvdepth"] = None if gd is None else np.zeros(P, np.float32)
    rc = L.hso_render_bwd(C.byref(c), _p(fwd["ranges"]), _p(fwd["point_list"]), _p(fwd["xy"]),
                          _p(fwd["conic_opacity"]), _p(fwd["rgb"]), _p(fwd["final_T"]), _p(fwd["n_contrib"]),
                          _p(g), _p(o["dL_dmean2D"]), _p(o["dL_dconic"]), _p(o["dL_dopacity"]),
                          _p(o["dL_dcolor"]), _p(o["abs_terms"]),
                          _p(fwd["depths"]) if gd is not None else None, _p(gd), _p(o["dL_dinvdepth"]))
    assert rc == 0
    o["dL_dmeans3D"] = np.zeros((P, 3), np.float32)
    o["dL_dshs"] = np.zeros((P, M, 3), np.float32) if shs is not None else None
    o["dL_dcolors_precomp"] = np.zeros((P, 3), np.float32) if colors_precomp is not None else None
    has_sr = scales is not None and rotations is not None
    o["dL_dscales"] = np.zeros((P, 3), np.float32) if has_sr else None
    o["dL_drots"] = np.zeros((P, 4), np.float32) if has_sr else None
    o["dL_dcov3D"] = np.zeros((P, 6), np.float32) if not has_sr else None
    rc = L.hso_preprocess_bwd(C.byref(c), _p(means3D), _p(shs), _p(colors_precomp), _p(scales), _p(rotations),
                              _p(cov3D_precomp), _p(fwd["radii"]), _p(fwd["cov3D"]), _p(fwd["clamped"]),
                              _p(fwd["rgb"]), _p(o["dL_dmean2D"]), _p(o["dL_dconic"]), _p(o["dL_dcolor"]),
                              _p(o["dL_dmeans3D"]), _p(o["dL_dshs"]), _p(o["dL_dcolors_precomp"]),
                              _p(o["dL_dscales"]), _p(o["dL_drots"]), _p(o["dL_dcov3D"]),
                              _p(_f32(fwd["opacities_in"])), _p(o["dL_dopacity"]), _p(o["dL_dinvdepth"]))
    assert rc == 0
    o["dL_dmeans2D"] = np.concatenate([o["dL_dmean2D"], np.zeros((P, 1), np.float32)], axis=1)
    if bounds:
        A = o["abs_terms"]
        o["n_terms"] = A[:, 10].astype(np.int64)
        outs = ["dL_dmeans3D", "dL_dshs", "dL_dcolors_precomp", "dL_dscales", "dL_drots", "dL_dcov3D"]
        t = {q: (None if o[q] is None else np.zeros(o[q].shape, np.float32)) for q in outs}
        m2, con, col = (np.ascontiguousarray(A[:, 0:2]), np.ascontiguousarray(A[:, 2:5]), np.ascontiguousarray(A[:, 6:9]))
        op, invd = np.ascontiguousarray(A[:, 5]), np.ascontiguousarray(A[:, 9])
        rc = L.hso_preprocess_bwd_abs(C.byref(c), _p(means3D), _p(shs), _p(scales), _p(rotations), _p(fwd["radii"]),
                                      _p(fwd["cov3D"]), _p(fwd["clamped"]), _p(fwd["rgb"]), _p(m2), _p(con), _p(col),
                                      _p(t["dL_dmeans3D"]), _p(t["dL_dshs"]), _p(t["dL_dcolors_precomp"]),
                                      _p(t["dL_dscales"]), _p(t["dL_drots"]), _p(t["dL_dcov3D"]),
                                      _p(_f32(fwd["opacities_in"])), _p(op), _p(invd if gd is not None else None))
        assert rc == 0
        for q in outs:
            o["abs_" + q] = t[q]
        o["abs_dL_dopacity"] = op        # (antialiasing scaled it in place; else it passed through)
        o["abs_dL_dmeans2D"] = np.concatenate([A[:, :2], np.zeros((P, 1), np.float32)], axis=1)
    return o


def tonemap_fwd(hdr, exposure: float, table, umin: float, umax: float):
    L = lib()
    hdr = _f32(hdr)
    table = _f32(table)
    K = table.shape[1]
    n = hdr.size // 3
    ldr = np.zeros_like(hdr)
    L.hso_tonemap_fwd(_p(hdr), C.c_int64(n), C.c_float(exposure), _p(table), C.c_int(K), C.c_float(umin),
                      C.c_float(umax), _p(ldr))
    return ldr


def tonemap_bwd(hdr, exposure: float, table, umin: float, umax: float, dL_dldr):
    L = lib()
    hdr, table, g = _f32(hdr), _f32(table), _f32(dL_dldr)
    K = table.shape[1]
    n = hdr.size // 3
    dh = np.zeros_like(hdr)
    dt = np.zeros_like(table)
    de = np.zeros(1, np.float32)
    rc = L.hso_tonemap_bwd(_p(hdr), C.c_int64(n), C.c_float(exposure), _p(table), C.c_int(K), C.c_float(umin),
                           C.c_float(umax), _p(g), _p(dh), _p(dt), _p(de))
    assert rc == 0
    return dh, dt, float(de[0])


def mark_visible(cam: Camera, means3D):
    L = lib()
    means3D = _f32(means3D)
    P = means3D.shape[0]
    c = cam.cstruct(P, 0)
    v = np.zeros(P, np.uint8)
    L.hso_mark_visible(C.byref(c), _p(means3D), _p(v))
    return v.astype(bool)
