"""Synthetic Gaussian clouds and cameras (fixed so it cannot be tuned).

The reference ships no data (/root/reference/Readme.md:57 "Still working on...."), and
there is no network, so every workload here is synthetic with the distribution that
SURVEY.md 8(d) froze: means uniform over the image with depth U(2,10), projected sigma
log-uniform in [0.5, 8] px, opacity sigmoid(N(0,1.5)), SH DC N(0,1), bands N(0,0.1).
All tensors are generated on the CPU from a seeded torch.Generator (so the CPU oracle
and the MI355X path see bit-identical inputs) and moved to `device` afterwards.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch


@dataclass
class Camera:
    """One pin-hole view in the matrix convention of GaussianRasterizationSettings:
    `viewmatrix` / `projmatrix` are the *transposed* world-to-view / full-projection
    matrices (row-vector convention), i.e. flat[4*j+i] is element (i,j)."""
    W: int
    H: int
    tanfovx: float
    tanfovy: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    campos: torch.Tensor


def projection_matrix(znear: float, zfar: float, tanfovx: float, tanfovy: float) -> torch.Tensor:
    """OpenGL-style perspective matrix with z_sign = +1 (column-vector convention)."""
    top, right = tanfovy * znear, tanfovx * znear
    Pm = torch.zeros(4, 4, dtype=torch.float64)
    Pm[0, 0] = 2.0 * znear / (2 * right)
    Pm[1, 1] = 2.0 * znear / (2 * top)
    Pm[3, 2] = 1.0
    Pm[2, 2] = zfar / (zfar - znear)
    Pm[2, 3] = -(zfar * znear) / (zfar - znear)
    return Pm


def make_camera(W: int, H: int, R: torch.Tensor | None = None, t: torch.Tensor | None = None,
                znear: float = 0.01, zfar: float = 100.0) -> Camera:
    """Camera with fx = fy = 1000 * W / 1920.  R,t are world-to-view
    (x_view = R x_world + t); identity at the origin looking down +z by default."""
    fx = fy = 1000.0 * (W / 1920.0)
    tanfovx, tanfovy = W / (2 * fx), H / (2 * fy)
    w2c = torch.eye(4, dtype=torch.float64)
    if R is not None:
        w2c[:3, :3] = R.to(torch.float64)
    if t is not None:
        w2c[:3, 3] = t.to(torch.float64)
    proj = projection_matrix(znear, zfar, tanfovx, tanfovy)
    full = proj @ w2c
    campos = torch.linalg.inv(w2c)[:3, 3]
    return Camera(W, H, tanfovx, tanfovy, w2c.t().contiguous().float(), full.t().contiguous().float(),
                  campos.float())


def yaw_camera(W: int, H: int, yaw_deg: float, centre_depth: float = 6.0) -> Camera:
    """View obtained by yawing the default camera about the cloud centre (0,0,centre_depth)."""
    a = math.radians(yaw_deg)
    Ry = torch.tensor([[math.cos(a), 0.0, math.sin(a)], [0.0, 1.0, 0.0], [-math.sin(a), 0.0, math.cos(a)]],
                      dtype=torch.float64)
    c = torch.tensor([0.0, 0.0, centre_depth], dtype=torch.float64)
    # x_view = Ry (x - c) + c
    return make_camera(W, H, Ry, c - Ry @ c)


def euler_rotation(roll: float, pitch: float, yaw: float) -> torch.Tensor:
    """R = Rz(roll) Rx(pitch) Ry(yaw) (radians, float64): roll about the optical axis, pitch about x, yaw about y."""
    cr, sr, cp, sp, cy, sy = (math.cos(roll), math.sin(roll), math.cos(pitch), math.sin(pitch), math.cos(yaw),
                              math.sin(yaw))
    Rz = torch.tensor([[cr, -sr, 0.0], [sr, cr, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float64)
    Rx = torch.tensor([[1.0, 0.0, 0.0], [0.0, cp, -sp], [0.0, sp, cp]], dtype=torch.float64)
    Ry = torch.tensor([[cy, 0.0, sy], [0.0, 1.0, 0.0], [-sy, 0.0, cy]], dtype=torch.float64)
    return Rz @ Rx @ Ry


def random_camera(W: int, H: int, seed: int, max_angle: float = math.pi, max_shift: float = 3.0) -> Camera:
    """A free 6-DoF view (the reference's figure draws a free camera trajectory, /root/reference/assets/pipeline.png
    "Camera motion spline"): world-to-view rotation Rz(roll) Rx(pitch) Ry(yaw) with each angle U(-max_angle, max_angle)
    and a translation U(-max_shift, max_shift)^3, so that EVERY entry of the view matrix is populated.  A cloud is put in
    front of it with make_scene(..., place_in=camera)."""
    g = torch.Generator().manual_seed(1_000_003 * 7 + seed)
    a = (torch.rand(3, generator=g, dtype=torch.float64) * 2 - 1) * max_angle
    t = (torch.rand(3, generator=g, dtype=torch.float64) * 2 - 1) * max_shift
    return make_camera(W, H, euler_rotation(float(a[0]), float(a[1]), float(a[2])), t)


def camera_w2c(cam: Camera) -> torch.Tensor:
    """World-to-view matrix [4,4] float64 (column-vector convention) of a Camera."""
    return cam.viewmatrix.t().to(torch.float64)


def perturbed_poses(base: Camera, n: int, seed: int = 0, rot_step_deg: float = 0.25, step: float = 0.01) -> list[Camera]:
    """n virtual poses of a motion-blurred exposure around `base` that differ in ROTATION as well as translation: pose k
    is base followed by a rotation of k * rot_step_deg about each of roll / pitch / yaw (signs drawn from `seed`) and a
    shift of k * step along a random unit direction, both in base's view frame (pose 0 is base itself)."""
    g = torch.Generator().manual_seed(2_000_003 + seed)
    sgn = torch.where(torch.rand(3, generator=g) < 0.5, -1.0, 1.0).to(torch.float64)
    d = torch.randn(3, generator=g, dtype=torch.float64)
    d = d / d.norm()
    w2c = camera_w2c(base)
    cams = []
    for k in range(n):
        a = math.radians(rot_step_deg) * k * sgn
        dR = euler_rotation(float(a[0]), float(a[1]), float(a[2]))
        R = dR @ w2c[:3, :3]
        t = dR @ w2c[:3, 3] + k * step * d
        cams.append(make_camera(base.W, base.H, R, t))
    return cams


@dataclass
class Scene:
    means3D: torch.Tensor      # [P,3]
    scales: torch.Tensor       # [P,3]
    rotations: torch.Tensor    # [P,4] unit (w,x,y,z)
    opacities: torch.Tensor    # [P,1]
    shs: torch.Tensor          # [P,M,3]
    sh_degree: int
    bg: torch.Tensor           # [3]
    dL_dimage: torch.Tensor    # [3,H,W] fixed upstream gradient
    camera: Camera
    exposure: torch.Tensor = field(default_factory=lambda: torch.tensor(0.5))
    crf_table: torch.Tensor | None = None   # [3,K]
    crf_range: tuple = (-6.0, 3.0)

    def to(self, device):
        kw = {}
        for k, v in self.__dict__.items():
            kw[k] = v.to(device) if isinstance(v, torch.Tensor) else v
        cam = self.camera
        kw["camera"] = Camera(cam.W, cam.H, cam.tanfovx, cam.tanfovy, cam.viewmatrix.to(device),
                              cam.projmatrix.to(device), cam.campos.to(device))
        return Scene(**kw)


def sigmoid_crf_table(K: int = 256, u_range=(-6.0, 3.0)) -> torch.Tensor:
    """Monotone sigmoid-shaped camera response on log-exposure, slightly different per channel."""
    u = torch.linspace(u_range[0], u_range[1], K, dtype=torch.float64)
    rows = [torch.sigmoid(1.2 * (u + 1.0 + 0.15 * ch)) for ch in range(3)]
    return torch.stack(rows).float().contiguous()


def make_scene(P: int, W: int, H: int, sh_degree: int = 0, seed: int = 0, hdr: bool = False,
               camera: Camera | None = None, crf_K: int = 256, place_in: Camera | None = None) -> Scene:
    """`camera`: the view the scene is rendered from (cloud stays in the DEFAULT camera's frustum -- a yawed camera sees
    it from the side).  `place_in`: a general camera (random_camera) the cloud is laid out IN FRONT OF instead: the same
    draws, re-expressed in world coordinates x_w = R^T (x_v - t), and that camera becomes the scene's."""
    g = torch.Generator().manual_seed(seed)
    if place_in is not None:
        assert camera is None, "give either camera= or place_in="
        camera = place_in
    cam = camera if camera is not None else make_camera(W, H)
    base = make_camera(W, H)  # the cloud is laid out in the default camera's frustum (then moved in front of place_in)
    fx = W / (2 * base.tanfovx)

    def U(*shape):
        return torch.rand(*shape, generator=g, dtype=torch.float64)

    def N(*shape):
        return torch.randn(*shape, generator=g, dtype=torch.float64)

    px, py = U(P) * W, U(P) * H
    z = 2.0 + 8.0 * U(P)
    ndc_x = (2 * px + 1) / W - 1
    ndc_y = (2 * py + 1) / H - 1
    means = torch.stack([ndc_x * base.tanfovx * z, ndc_y * base.tanfovy * z, z], dim=1)
    if place_in is not None:
        w2c = camera_w2c(place_in)
        means = (means - w2c[:3, 3]) @ w2c[:3, :3]      # rows: R^T (x_v - t)
    sigma_px = torch.exp(math.log(0.5) + (math.log(8.0) - math.log(0.5)) * U(P))
    aniso = torch.exp(U(P, 3) - 0.5)
    scales = (sigma_px * z / fx)[:, None] * aniso
    q = N(P, 4)
    q = q / q.norm(dim=1, keepdim=True)
    opac = torch.sigmoid(1.5 * N(P, 1))
    M = (sh_degree + 1) ** 2
    shs = torch.empty(P, M, 3, dtype=torch.float64)
    shs[:, 0] = N(P, 3)
    if M > 1:
        shs[:, 1:] = 0.1 * N(P, M - 1, 3)
    if hdr:
        shs[:, 0] = shs[:, 0] * torch.exp(3.0 * U(P, 1))
    dL = N(3, H, W)
    return Scene(means.float(), scales.float(), q.float(), opac.float(), shs.float().contiguous(), sh_degree,
                 torch.zeros(3), dL.float(), cam, torch.tensor(0.5),
                 sigmoid_crf_table(crf_K) if hdr else None)


def blur_poses(W: int, H: int, n: int = 8, step: float = 0.01) -> list[Camera]:
    """c4: n virtual poses, camera translated along +x by k*step (world frame)."""
    cams = []
    for k in range(n):
        # camera centre at (k*step,0,0): x_view = x_world - centre
        cams.append(make_camera(W, H, None, torch.tensor([-k * step, 0.0, 0.0])))
    return cams
