"""Physical image-formation model of CasualHDRSplat on top of the rasterizer.

What the reference establishes (it ships no code): /root/reference/Readme.md:54 -- "a unified model based on the
physical image formation process, integrating camera motion blur and exposure-induced brightness variations",
jointly estimating camera motion, exposure time and camera response curve while reconstructing the HDR scene --
and /root/reference/assets/pipeline.png: trajectory control knots T_j define a camera motion spline; virtual
camera poses are sampled inside the exposure window of frame i; each renders a virtual sharp HDR image H_k;
exposure dt_i and the shared implicit CRF F_theta turn it into a virtual sharp LDR image I_k; their average is
the estimated blurry LDR image B_i that is compared with the captured frame.

Everything here is small host-side PyTorch (SE(3) math on a handful of poses, a tiny MLP); the heavy lifting --
N renders, tone-map, averaging, and the gradients w.r.t. Gaussians, exposure, CRF table and camera poses -- is
ONE call into the HIP rasterizer.  Parameterisations the reference does not specify ([DESIGN]): linear
interpolation in se(3) between the two knots bracketing the exposure window; CRF = monotone MLP on log-exposure
sampled on K knots.
"""
from __future__ import annotations

import math
from typing import Callable, Optional

import torch
import torch.nn as nn

from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer


def _hat(w: torch.Tensor) -> torch.Tensor:
    z = torch.zeros_like(w[..., 0])
    return torch.stack([torch.stack([z, -w[..., 2], w[..., 1]], -1), torch.stack([w[..., 2], z, -w[..., 0]], -1),
                        torch.stack([-w[..., 1], w[..., 0], z], -1)], -2)


def se3_exp(xi: torch.Tensor) -> torch.Tensor:
    """Exponential map se(3) -> SE(3).  xi[..., :3] = translation part rho, xi[..., 3:] = rotation vector omega.
    Returns [..., 4, 4] world-to-camera matrices.  Differentiable (series near theta = 0)."""
    rho, om = xi[..., :3], xi[..., 3:]
    th2 = (om * om).sum(-1)
    th = torch.sqrt(th2 + 1e-20)
    small = th2 < 1e-8
    A = torch.where(small, 1 - th2 / 6, torch.sin(th) / th)
    B = torch.where(small, 0.5 - th2 / 24, (1 - torch.cos(th)) / (th2 + 1e-20))
    Cc = torch.where(small, 1.0 / 6 - th2 / 120, (1 - A) / (th2 + 1e-20))
    K = _hat(om)
    K2 = K @ K
    eye = torch.eye(3, dtype=xi.dtype, device=xi.device).expand(K.shape)
    R = eye + A[..., None, None] * K + B[..., None, None] * K2
    V = eye + B[..., None, None] * K + Cc[..., None, None] * K2
    t = (V @ rho[..., None])[..., 0]
    T = torch.zeros(*xi.shape[:-1], 4, 4, dtype=xi.dtype, device=xi.device)
    T[..., :3, :3] = R
    T[..., :3, 3] = t
    T[..., 3, 3] = 1.0
    return T


class TrajectorySpline(nn.Module):
    """Learnable camera trajectory: one se(3) control knot per captured-frame boundary (T_j in the figure).
    `poses(i, n)` samples n virtual world-to-camera poses uniformly inside frame i's exposure window, which spans
    the segment between knots i and i+1 (interpolated in the Lie algebra)."""

    def __init__(self, init_w2c: torch.Tensor):
        """init_w2c: [J, 4, 4] initial world-to-camera matrices of the knots (e.g. from SfM)."""
        super().__init__()
        self.register_buffer("base", init_w2c.clone().float())
        self.delta = nn.Parameter(torch.zeros(init_w2c.shape[0], 6))  # left-multiplied se(3) corrections

    def knot(self, j: int) -> torch.Tensor:
        return se3_exp(self.delta[j]) @ self.base[j]

    def poses(self, i: int, n: int) -> torch.Tensor:
        T0, T1 = self.knot(i), self.knot(i + 1)
        rel = T1 @ torch.linalg.inv(T0)  # motion over the exposure window
        xi = se3_log(rel)
        ts = (torch.arange(n, dtype=T0.dtype, device=T0.device) + 0.5) / n
        return se3_exp(ts[:, None] * xi[None, :]) @ T0


def se3_log(T: torch.Tensor) -> torch.Tensor:
    """Logarithm SE(3) -> se(3) for rotations well below pi (adjacent video frames)."""
    R, t = T[:3, :3], T[:3, 3]
    cos = ((R.diagonal().sum() - 1) / 2).clamp(-1 + 1e-7, 1 - 1e-7)
    th = torch.acos(cos)
    small = th < 1e-4
    k = torch.where(small, 0.5 + th * th / 12, th / (2 * torch.sin(th) + 1e-20))
    om = k * torch.stack([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    K = _hat(om)
    th2 = th * th
    A = torch.where(small, 1 - th2 / 6, torch.sin(th) / (th + 1e-20))
    B = torch.where(small, 0.5 - th2 / 24, (1 - torch.cos(th)) / (th2 + 1e-20))
    coef = torch.where(small, torch.full_like(th, 1.0 / 12), (1 - A / (2 * B)) / (th2 + 1e-20))
    Vinv = torch.eye(3, dtype=T.dtype, device=T.device) - 0.5 * K + coef * (K @ K)
    return torch.cat([Vinv @ t, om])


class ImplicitCRF(nn.Module):
    """Shared implicit camera response F_theta: log-exposure u -> LDR value per channel, sampled on K knots so the
    rasterizer can evaluate it per pixel (`crf_table`).  Monotone by construction (positive increments)."""

    def __init__(self, K: int = 256, u_range=(-6.0, 3.0), hidden: int = 32):
        super().__init__()
        self.K, self.u_range = K, (float(u_range[0]), float(u_range[1]))
        self.net = nn.Sequential(nn.Linear(1, hidden), nn.Tanh(), nn.Linear(hidden, hidden), nn.Tanh(), nn.Linear(hidden, 3))
        self.register_buffer("knots", torch.linspace(self.u_range[0], self.u_range[1], K)[:, None])

    def table(self) -> torch.Tensor:
        inc = torch.nn.functional.softplus(self.net(self.knots))        # [K,3] positive slopes
        cdf = torch.cumsum(inc, dim=0)
        tab = (cdf - cdf[:1]) / (cdf[-1:] - cdf[:1] + 1e-12)            # 0 at u_min, 1 at u_max, increasing
        return tab.t().contiguous()                                     # [3,K]


def projection_matrix(tanfovx: float, tanfovy: float, znear: float = 0.01, zfar: float = 100.0, device=None):
    Pm = torch.zeros(4, 4, device=device)
    Pm[0, 0], Pm[1, 1] = 1.0 / tanfovx, 1.0 / tanfovy
    Pm[3, 2] = 1.0
    Pm[2, 2] = zfar / (zfar - znear)
    Pm[2, 3] = -(zfar * znear) / (zfar - znear)
    return Pm


class HDRBlurFormation(nn.Module):
    """B_i = mean_k F_theta(dt_i * H(G, T_i(t_k))) -- one rasterizer call per captured frame."""

    def __init__(self, trajectory: TrajectorySpline, n_frames: int, W: int, H: int, tanfovx: float, tanfovy: float,
                 n_virtual: int = 8, crf: Optional[ImplicitCRF] = None, blur_domain: str = "ldr", sh_degree: int = 3,
                 rasterizer_factory: Callable = GaussianRasterizer):
        super().__init__()
        self.trajectory = trajectory
        self.crf = crf if crf is not None else ImplicitCRF()
        self.log_exposure = nn.Parameter(torch.zeros(n_frames))          # dt_i = exp(log_exposure_i)
        self.W, self.H, self.tanfovx, self.tanfovy = W, H, tanfovx, tanfovy
        self.n_virtual, self.blur_domain, self.sh_degree = n_virtual, blur_domain, sh_degree
        self._factory = rasterizer_factory

    def cameras(self, i: int):
        """(viewmatrices [N,4,4], projmatrices [N,4,4], camposes [N,3]) in the rasterizer's transposed convention."""
        w2c = self.trajectory.poses(i, self.n_virtual)
        proj = projection_matrix(self.tanfovx, self.tanfovy, device=w2c.device).to(w2c.dtype)
        full = proj[None] @ w2c
        campos = -(w2c[:, :3, :3].transpose(1, 2) @ w2c[:, :3, 3:])[..., 0]
        return w2c.transpose(1, 2).contiguous(), full.transpose(1, 2).contiguous(), campos.contiguous()

    def forward(self, i: int, means3D, opacities, shs, scales, rotations, bg=None):
        V, PV, Cp = self.cameras(i)
        dev = means3D.device
        bg = torch.zeros(3, device=dev) if bg is None else bg
        settings = GaussianRasterizationSettings(
            image_height=self.H, image_width=self.W, tanfovx=self.tanfovx, tanfovy=self.tanfovy, bg=bg,
            scale_modifier=1.0, viewmatrix=V[0], projmatrix=PV[0], sh_degree=self.sh_degree, campos=Cp[0],
            prefiltered=False, debug=False, exposure=torch.exp(self.log_exposure[i]), crf_table=self.crf.table(),
            crf_range=self.crf.u_range, viewmatrices=V, projmatrices=PV, camposes=Cp, blur_domain=self.blur_domain)
        means2D = torch.zeros_like(means3D, requires_grad=means3D.requires_grad)
        ldr, radii, hdr = self._factory(settings)(means3D, means2D, opacities, shs=shs, scales=scales,
                                                   rotations=rotations)
        return ldr, hdr, radii, means2D


def knots_from_lookat(n: int, radius: float = 0.05, depth: float = 6.0) -> torch.Tensor:
    """Helper for synthetic tests: n knots on a small arc in front of the cloud, all looking down +z."""
    Ts = []
    for j in range(n):
        a = (j / max(n - 1, 1) - 0.5) * 2 * radius
        T = torch.eye(4)
        T[0, 3] = -a                      # camera centre at (a, 0, 0)
        T[:3, :3] = torch.tensor([[math.cos(a / depth), 0, math.sin(a / depth)], [0, 1, 0],
                                  [-math.sin(a / depth), 0, math.cos(a / depth)]])
        Ts.append(T)
    return torch.stack(Ts)
