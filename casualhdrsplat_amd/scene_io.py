"""Scene I/O either side of the rasterizer: Gaussian clouds as PLY in the layout every 3DGS
trainer and viewer exchanges, and COLMAP sparse reconstructions (the "SfM" arrow of
/root/reference/assets/pipeline.png) as initialisation.  Host-side numpy code, no GPU work: the rasterizer takes the
tensors these functions return.

PLY layout (binary little endian or ascii, one `vertex` element), properties in this order:
    x y z  nx ny nz  f_dc_0..2  f_rest_0..3*(M-1)-1  opacity  scale_0..2  rot_0..3
with f_rest stored CHANNEL-major (f_rest_{c*(M-1)+k} = coefficient k+1 of channel c), opacity as a logit, scales as
logs, rot as an un-normalised (w, x, y, z) quaternion -- i.e. the pre-activation parameters of the optimiser.
"""
from __future__ import annotations

import math
import struct
from dataclasses import dataclass
from typing import Dict, Tuple

import numpy as np
import torch

SH_C0 = 0.28209479177387814

_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
              "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
              "double": "f8", "float64": "f8"}


@dataclass
class GaussianCloud:
    """Pre-activation parameters as stored on disk."""
    means3D: torch.Tensor        # [P,3]
    shs: torch.Tensor            # [P,M,3]
    opacity_logit: torch.Tensor  # [P,1]
    log_scales: torch.Tensor     # [P,3]
    rotations: torch.Tensor      # [P,4] (w,x,y,z), not normalised

    @property
    def sh_degree(self) -> int:
        return int(round(math.sqrt(self.shs.shape[1]))) - 1

    def activated(self, device=None) -> Dict[str, torch.Tensor]:
        """The tensors GaussianRasterizer.forward takes: opacities = sigmoid, scales = exp, unit quaternions."""
        d = dict(means3D=self.means3D, shs=self.shs, opacities=torch.sigmoid(self.opacity_logit),
                 scales=torch.exp(self.log_scales),
                 rotations=self.rotations / self.rotations.norm(dim=1, keepdim=True).clamp_min(1e-12))
        return {k: (v.to(device) if device is not None else v).contiguous() for k, v in d.items()}


def _property_names(M: int):
    names = ["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)]
    names += [f"f_rest_{i}" for i in range(3 * (M - 1))]
    names += ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)]
    return names


def save_ply(path: str, cloud: GaussianCloud) -> None:
    m = cloud.means3D.detach().cpu().numpy().astype(np.float32)
    sh = cloud.shs.detach().cpu().numpy().astype(np.float32)
    P, M = sh.shape[0], sh.shape[1]
    cols = [m, np.zeros((P, 3), np.float32), sh[:, 0, :],
            np.transpose(sh[:, 1:, :], (0, 2, 1)).reshape(P, 3 * (M - 1)),  # channel-major rest
            cloud.opacity_logit.detach().cpu().numpy().astype(np.float32).reshape(P, 1),
            cloud.log_scales.detach().cpu().numpy().astype(np.float32),
            cloud.rotations.detach().cpu().numpy().astype(np.float32)]
    table = np.ascontiguousarray(np.concatenate(cols, axis=1), dtype="<f4")
    names = _property_names(M)
    assert table.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % P
    header += "".join(f"property float {n}\n" for n in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(table.tobytes())


def _read_ply_vertices(path: str) -> Dict[str, np.ndarray]:
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, in_vertex = None, None, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                if in_vertex:
                    in_vertex = False          # elements after `vertex` are ignored (they follow it in the body)
                elif tok[1] == "vertex":
                    in_vertex, count = True, int(tok[2])
                elif count is None:
                    raise ValueError(f"{path}: an element precedes `vertex`; not a Gaussian-cloud PLY")
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list property in the vertex element")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if count is None or fmt is None:
            raise ValueError(f"{path}: no vertex element / format line")
        if fmt == "ascii":
            rows = np.loadtxt(f, dtype=np.float64, max_rows=count, ndmin=2)
            if rows.shape != (count, len(props)):
                raise ValueError(f"{path}: expected {count}x{len(props)} ascii values, got {rows.shape}")
            return {n: rows[:, i] for i, (n, _) in enumerate(props)}
        if fmt not in ("binary_little_endian", "binary_big_endian"):
            raise ValueError(f"{path}: unsupported PLY format {fmt}")
        end = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(n, end + t) for n, t in props])
        raw = f.read(dt.itemsize * count)
        if len(raw) != dt.itemsize * count:
            raise ValueError(f"{path}: truncated vertex data")
        rec = np.frombuffer(raw, dtype=dt, count=count)
        return {n: rec[n] for n, _ in props}


def load_ply(path: str) -> GaussianCloud:
    v = _read_ply_vertices(path)
    for need in ("x", "y", "z", "f_dc_0", "opacity", "scale_0", "rot_0"):
        if need not in v:
            raise ValueError(f"{path}: property `{need}` missing; not a Gaussian-cloud PLY")
    P = v["x"].shape[0]
    nrest = sum(1 for k in v if k.startswith("f_rest_"))
    if nrest % 3:
        raise ValueError(f"{path}: {nrest} f_rest properties is not a multiple of 3")
    M = nrest // 3 + 1
    if int(round(math.sqrt(M))) ** 2 != M:
        raise ValueError(f"{path}: {M} SH coefficients per channel is not a square")

    def col(*names):
        return np.stack([np.asarray(v[n], np.float32) for n in names], axis=1)

    sh = np.zeros((P, M, 3), np.float32)
    sh[:, 0, :] = col("f_dc_0", "f_dc_1", "f_dc_2")
    if M > 1:
        rest = col(*[f"f_rest_{i}" for i in range(nrest)]).reshape(P, 3, M - 1)
        sh[:, 1:, :] = np.transpose(rest, (0, 2, 1))
    t = torch.from_numpy
    return GaussianCloud(t(col("x", "y", "z")), t(sh), t(col("opacity")), t(col("scale_0", "scale_1", "scale_2")),
                         t(col("rot_0", "rot_1", "rot_2", "rot_3")))


# ------------------------------------------------------------------------------------------- COLMAP sparse models
_CAMERA_MODELS = {0: ("SIMPLE_PINHOLE", 3), 1: ("PINHOLE", 4), 2: ("SIMPLE_RADIAL", 4), 3: ("RADIAL", 5),
                  4: ("OPENCV", 8), 5: ("OPENCV_FISHEYE", 8), 6: ("FULL_OPENCV", 12), 7: ("FOV", 5),
                  8: ("SIMPLE_RADIAL_FISHEYE", 4), 9: ("RADIAL_FISHEYE", 5), 10: ("THIN_PRISM_FISHEYE", 12)}
_MODEL_PARAMS = {name: n for name, n in _CAMERA_MODELS.values()}


@dataclass
class ColmapCamera:
    id: int
    model: str
    width: int
    height: int
    params: np.ndarray

    def focal(self) -> Tuple[float, float]:
        if self.model in ("SIMPLE_PINHOLE", "SIMPLE_RADIAL", "RADIAL", "SIMPLE_RADIAL_FISHEYE", "RADIAL_FISHEYE"):
            return float(self.params[0]), float(self.params[0])
        return float(self.params[0]), float(self.params[1])


@dataclass
class ColmapImage:
    id: int
    qvec: np.ndarray   # (w, x, y, z), world-to-camera rotation
    tvec: np.ndarray   # world-to-camera translation
    camera_id: int
    name: str


def _unpack(f, fmt):
    n = struct.calcsize(fmt)
    b = f.read(n)
    if len(b) != n:
        raise ValueError("truncated COLMAP binary file")
    return struct.unpack(fmt, b)


def read_points3D(path: str):
    """points3D.txt / points3D.bin -> (xyz float64 [N,3], rgb uint8 [N,3], reprojection error float64 [N])."""
    xyz, rgb, err = [], [], []
    if path.endswith(".txt"):
        with open(path) as f:
            for line in f:
                line = line.strip()
                if not line or line[0] == "#":
                    continue
                t = line.split()
                xyz.append([float(t[1]), float(t[2]), float(t[3])])
                rgb.append([int(t[4]), int(t[5]), int(t[6])])
                err.append(float(t[7]))
    else:
        with open(path, "rb") as f:
            (n,) = _unpack(f, "<Q")
            for _ in range(n):
                _id, x, y, z, r, g, b, e, track = _unpack(f, "<QdddBBBdQ")
                f.seek(8 * track, 1)
                xyz.append([x, y, z]); rgb.append([r, g, b]); err.append(e)
    return (np.asarray(xyz, np.float64).reshape(-1, 3), np.asarray(rgb, np.uint8).reshape(-1, 3),
            np.asarray(err, np.float64))


def read_cameras(path: str) -> Dict[int, ColmapCamera]:
    cams = {}
    if path.endswith(".txt"):
        with open(path) as f:
            for line in f:
                line = line.strip()
                if not line or line[0] == "#":
                    continue
                t = line.split()
                cams[int(t[0])] = ColmapCamera(int(t[0]), t[1], int(t[2]), int(t[3]), np.asarray(t[4:], np.float64))
    else:
        with open(path, "rb") as f:
            (n,) = _unpack(f, "<Q")
            for _ in range(n):
                cid, model_id, w, h = _unpack(f, "<iiQQ")
                name, npar = _CAMERA_MODELS[model_id]
                cams[cid] = ColmapCamera(cid, name, int(w), int(h), np.asarray(_unpack(f, "<" + "d" * npar), np.float64))
    for c in cams.values():
        if c.model not in _MODEL_PARAMS or len(c.params) != _MODEL_PARAMS[c.model]:
            raise ValueError(f"camera {c.id}: model {c.model} with {len(c.params)} parameters")
    return cams


def read_images(path: str) -> Dict[int, ColmapImage]:
    ims = {}
    if path.endswith(".txt"):
        with open(path) as f:
            lines = [ln.rstrip("\n") for ln in f if not ln.startswith("#")]
        # two lines per image; the second (2-D points) may be empty
        i = 0
        while i < len(lines):
            if not lines[i].strip():
                i += 1
                continue
            t = lines[i].split()
            ims[int(t[0])] = ColmapImage(int(t[0]), np.asarray(t[1:5], np.float64), np.asarray(t[5:8], np.float64),
                                         int(t[8]), " ".join(t[9:]))
            i += 2
    else:
        with open(path, "rb") as f:
            (n,) = _unpack(f, "<Q")
            for _ in range(n):
                iid, qw, qx, qy, qz, tx, ty, tz, cid = _unpack(f, "<idddddddi")
                name = b""
                while True:
                    ch = f.read(1)
                    if ch in (b"\x00", b""):
                        break
                    name += ch
                (npts,) = _unpack(f, "<Q")
                f.seek(24 * npts, 1)
                ims[iid] = ColmapImage(iid, np.array([qw, qx, qy, qz]), np.array([tx, ty, tz]), cid, name.decode("utf-8"))
    return ims


def qvec_to_rotmat(q: np.ndarray) -> np.ndarray:
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def colmap_view(image: ColmapImage, camera: ColmapCamera, znear: float = 0.01, zfar: float = 100.0):
    """(viewmatrix, projmatrix, campos, tanfovx, tanfovy) in the rasterizer's transposed convention."""
    from .synthetic import projection_matrix
    R, t = qvec_to_rotmat(image.qvec), image.tvec
    w2c = np.eye(4)
    w2c[:3, :3], w2c[:3, 3] = R, t
    fx, fy = camera.focal()
    tanfovx, tanfovy = camera.width / (2.0 * fx), camera.height / (2.0 * fy)
    V = torch.from_numpy(w2c.T.copy())                               # transposed: row-vector convention
    full = V @ projection_matrix(znear, zfar, tanfovx, tanfovy).t()  # (proj @ w2c)^T
    campos = torch.from_numpy(-R.T @ t).float()
    return V.float().contiguous(), full.float().contiguous(), campos, tanfovx, tanfovy


def init_from_points(xyz: np.ndarray, rgb: np.ndarray, sh_degree: int = 3, initial_opacity: float = 0.1) -> GaussianCloud:
    """The published SfM initialisation: SH DC from the point colour, isotropic scale = distance scale of the three
    nearest neighbours (sqrt of the mean squared distance, floored at 1e-7), identity rotation, opacity 0.1."""
    from scipy.spatial import cKDTree
    xyz = np.asarray(xyz, np.float64)
    P = xyz.shape[0]
    M = (sh_degree + 1) ** 2
    k = min(4, P)
    if k > 1:
        d, _ = cKDTree(xyz).query(xyz, k=k)
        d2 = np.mean(d[:, 1:] ** 2, axis=1)
    else:
        d2 = np.ones(P)
    log_s = np.log(np.sqrt(np.maximum(d2, 1e-7)))
    sh = np.zeros((P, M, 3), np.float32)
    sh[:, 0, :] = (np.asarray(rgb, np.float64) / 255.0 - 0.5) / SH_C0
    rot = np.zeros((P, 4), np.float32)
    rot[:, 0] = 1.0
    logit = math.log(initial_opacity / (1.0 - initial_opacity))
    t = torch.from_numpy
    return GaussianCloud(t(xyz.astype(np.float32)), t(sh), torch.full((P, 1), logit, dtype=torch.float32),
                         t(np.repeat(log_s[:, None], 3, axis=1).astype(np.float32)), t(rot))
