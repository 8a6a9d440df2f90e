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
ONE call into the HIP rasterizer.

The figure's trajectory is a "Camera motion spline" through FOUR control knots T_j .. T_{j+3}, and the thick arc on it
is the "Exposure time range" whose length is the learnable exposure time dt_i.  Both are modelled (round 5):

  * `TrajectorySpline(kind="cubic")`: cumulative uniform cubic B-spline in SE(3) over four knots (C2-continuous camera
    motion); `kind="linear"` keeps the two-knot form (geodesic between the knots bracketing the window).
  * the virtual poses of frame i are sampled at times t_i + (s_k - 1/2) * dt_i * window_scale, s_k = (k + 1/2) / n, i.e.
    inside [t_i - w/2, t_i + w/2] with w the exposure time expressed in knot intervals: dt_i sets the blur EXTENT as well
    as the brightness, and dL/d(dt_i) has a motion-blur term that arrives through the rasterizer's pose gradients
    (SURVEY.md 8f n1).  `window_from_exposure=False` pins the window to a fixed length (the round-4 behaviour).

Parameterisations the reference does not specify ([DESIGN]): left-multiplied se(3) corrections on SfM knots, uniform
knot times (knot j at time j), the CRF as a monotone MLP on log-exposure sampled on K knots.
"""
from __future__ import annotations

import math
import os
from typing import Callable, Optional

import torch
import torch.nn as nn

from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer


def _mm(A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """A @ B for stacks of 3 x 3 / 4 x 4 blocks as one broadcast multiply and one sum: a BLAS call per product costs the
    host five times as much at this size (26 us against 2 x 5), and a step makes dozens of them."""
    return (A.unsqueeze(-1) * B.unsqueeze(-3)).sum(-2)


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
    K2 = _mm(K, K)
    eye = torch.eye(3, dtype=xi.dtype, device=xi.device).expand(K.shape)
    R = eye + A[..., None, None] * K + B[..., None, None] * K2
    V = eye + B[..., None, None] * K + Cc[..., None, None] * K2
    t = _mm(V, rho[..., None])[..., 0]
    T = torch.zeros(*xi.shape[:-1], 4, 4, dtype=xi.dtype, device=xi.device)
    T[..., :3, :3] = R
    T[..., :3, 3] = t
    T[..., 3, 3] = 1.0
    return T


# Cumulative basis of the uniform cubic B-spline (Lovegrove et al., "Spline fusion", 2013): pose(u) =
# exp(B3 xi3) exp(B2 xi2) exp(B1 xi1) T_j with xi_k = log(T_{j+k} T_{j+k-1}^-1); B(0) = (5/6, 1/6, 0), B(1) = (1, 5/6, 1/6),
# so neighbouring segments agree in value, velocity and acceleration.
def _cubic_cumulative_basis(u: torch.Tensor):
    u2, u3 = u * u, u * u * u
    return (5.0 + 3.0 * u - 3.0 * u2 + u3) / 6.0, (1.0 + 3.0 * u + 3.0 * u2 - 2.0 * u3) / 6.0, u3 / 6.0


class _SplinePoses(torch.autograd.Function):
    """pose_at on the GPU as ONE kernel (libhdrsplat's hs_spline_poses: forward-mode dual numbers in float64, one thread per
    sample time) instead of ~300 tiny tensor kernels forward and ~700 backward: the launch returns the poses and their
    Jacobian with respect to the knot corrections governing each sample's segment and to the sample time; backward is a
    multiply-sum with it and a deterministic scatter into the knots' rows."""

    @staticmethod
    def forward(ctx, delta, base, t, kind):
        from . import _lib as L
        from .rasterizer import _stream
        J, T = delta.shape[0], t.numel()
        d32 = delta.detach().to(torch.float32).contiguous()
        b32 = base.detach().to(torch.float32).contiguous()
        t32 = t.detach().to(torch.float32).contiguous().reshape(-1)
        w2c = torch.empty(T, 4, 4, dtype=torch.float32, device=delta.device)
        jac = torch.empty(T, 12, 25, dtype=torch.float32, device=delta.device)
        seg = torch.empty(T, dtype=torch.int32, device=delta.device)
        with torch.cuda.device(delta.device):
            L.check(L.load().hs_spline_poses(J, T, 1 if kind == "cubic" else 0, d32.data_ptr(), b32.data_ptr(), t32.data_ptr(),
                                             w2c.data_ptr(), jac.data_ptr(), seg.data_ptr(), _stream()), "hs_spline_poses")
        ctx.save_for_backward(jac, seg)
        ctx.J, ctx.dtypes = J, (delta.dtype, t.dtype)
        return w2c.to(delta.dtype)

    @staticmethod
    @torch.autograd.function.once_differentiable   # (the saved Jacobian is a constant: no second-order terms through here)
    def backward(ctx, g):
        jac, seg = ctx.saved_tensors
        T, J = jac.shape[0], ctx.J
        gi = (g[:, :3, :].reshape(T, 12, 1).to(torch.float32) * jac).sum(1)                     # [T, 25]
        # rows seg .. seg + 3 of the knots receive gi[:, 6k : 6k + 6]: a one-hot multiply-sum (fixed summation order: the
        # same bits every run, unlike index_add_'s atomics)
        rows = seg.to(torch.int64)[:, None] + torch.arange(4, device=g.device)[None, :]        # [T, 4]
        onehot = (rows[..., None] == torch.arange(J, device=g.device)).to(torch.float32)       # [T, 4, J]
        gd = (onehot[..., None] * gi[:, :24].reshape(T, 4, 1, 6)).sum((0, 1))                   # [J, 6]
        return gd.to(ctx.dtypes[0]), None, gi[:, 24].to(ctx.dtypes[1]), None


class TrajectorySpline(nn.Module):
    """Learnable camera trajectory over se(3) control knots (T_j in the figure; knot j sits at time j).

    kind="cubic": cumulative uniform cubic B-spline in SE(3) -- the pose at time t in [j + 1, j + 2) is governed by the
    four knots T_j .. T_{j+3} (the four knots the figure draws around one exposure window); defined for t in
    [1, J - 2]; C2-continuous; reproduces constant-velocity motion exactly (knots exp(k xi) T_0 give exp(t xi) T_0,
    so such knots are interpolated), otherwise it approximates its control knots like any B-spline.
    kind="linear": geodesic between the two knots bracketing t (t in [0, J - 1]).
    """

    def __init__(self, init_w2c: torch.Tensor, kind: str = "cubic"):
        """init_w2c: [J, 4, 4] initial world-to-camera matrices of the knots (e.g. from SfM).  kind: "cubic" (default: the
        figure's model -- four control knots around every exposure window, /root/reference/assets/pipeline.png) or "linear"."""
        super().__init__()
        if kind not in ("linear", "cubic"):
            raise ValueError("kind must be 'linear' or 'cubic'")
        if kind == "cubic" and init_w2c.shape[0] < 4:
            raise ValueError("a cubic spline segment needs four control knots")
        if kind == "linear" and init_w2c.shape[0] < 2:
            raise ValueError("a trajectory needs at least two control knots")
        self.kind = kind
        # on the GPU pose_at is one kernel of libhdrsplat (hs_spline_poses); False: the tensor-operation form below, which is
        # also the CPU path and the kernel's oracle in the tests
        self.fused = os.environ.get("HS_SPLINE_FUSED", "1") != "0"   # (0: experiments that take the kernel out of the step)
        self.register_buffer("base", init_w2c.clone().float())
        self.delta = nn.Parameter(torch.zeros(init_w2c.shape[0], 6))  # left-multiplied se(3) corrections

    @property
    def t_range(self):
        """(t_min, t_max): the times the trajectory is defined on."""
        J = self.base.shape[0]
        return (1.0, float(J - 2)) if self.kind == "cubic" else (0.0, float(J - 1))

    def knot(self, j: int) -> torch.Tensor:
        return _mm(se3_exp(self.delta[j]), self.base[j])

    def knots(self) -> torch.Tensor:
        """All corrected control knots [J, 4, 4]."""
        return _mm(se3_exp(self.delta), self.base.to(self.delta.dtype))

    def pose_at(self, t: torch.Tensor) -> torch.Tensor:
        """World-to-camera matrices [n, 4, 4] at times t [n]; differentiable w.r.t. the knots AND w.r.t. t.  Pure tensor
        code (segment look-up by index_select): no host read of t, so it neither synchronises nor breaks a graph capture."""
        t = t.reshape(-1)
        J = self.base.shape[0]
        # (the kernel computes in float64 but takes and returns float32: a module of another dtype -- the float64 twins of
        # the tests -- keeps the tensor form, at its own precision)
        if self.fused and t.is_cuda and self.delta.is_cuda and self.delta.dtype == torch.float32:
            return _SplinePoses.apply(self.delta, self.base, t, self.kind)
        Tk = self.knots()
        inc = se3_log(_mm(Tk[1:], rigid_inv(Tk[:-1])))           # [J - 1, 6]: knot(j + 1) = exp(inc[j]) knot(j)
        fl = torch.floor(t.detach()).long()
        if self.kind == "linear":
            j = fl.clamp(0, J - 2)
            u = t - j.to(t.dtype)
            return _mm(se3_exp(u[:, None] * inc.index_select(0, j)), Tk.index_select(0, j))
        j = (fl - 1).clamp(0, J - 4)
        u = t - (j + 1).to(t.dtype)
        b1, b2, b3 = _cubic_cumulative_basis(u)
        x1, x2, x3 = inc.index_select(0, j), inc.index_select(0, j + 1), inc.index_select(0, j + 2)
        return _mm(_mm(_mm(se3_exp(b3[:, None] * x3), se3_exp(b2[:, None] * x2)), se3_exp(b1[:, None] * x1)), Tk.index_select(0, j))

    def window_times(self, t_mid, width, n: int) -> torch.Tensor:
        """n sample times uniformly inside [t_mid - width / 2, t_mid + width / 2] (midpoint rule); differentiable
        w.r.t. both."""
        p = next(self.parameters())
        s = (torch.arange(n, dtype=p.dtype, device=p.device) + 0.5) / n - 0.5
        return torch.as_tensor(t_mid, dtype=p.dtype, device=p.device) + s * torch.as_tensor(width, dtype=p.dtype, device=p.device)

    def poses(self, i: int, n: int) -> torch.Tensor:
        """Round-4 form: n virtual poses uniformly over the whole interval between knots i and i + 1 (kind="linear"),
        or over the segment [i + 1, i + 2) governed by knots i .. i + 3 (kind="cubic")."""
        t_mid = i + 0.5 if self.kind == "linear" else i + 1.5
        return self.pose_at(self.window_times(t_mid, 1.0, n))


def rigid_inv(T: torch.Tensor) -> torch.Tensor:
    """Inverse of rigid transforms [..., 4, 4]: [R t; 0 1]^-1 = [R^T, -R^T t; 0 1].  (Not torch.linalg.inv: that is a solver
    call with a host-side status check -- slower for 4 x 4 blocks and not allowed inside a graph capture.)"""
    Rt = T[..., :3, :3].transpose(-1, -2)
    t = -_mm(Rt, T[..., :3, 3:])
    top = torch.cat([Rt, t], -1)
    bottom = torch.zeros_like(T[..., 3:, :])
    bottom[..., 0, 3] = 1.0
    return torch.cat([top, bottom], -2)


def se3_log(T: torch.Tensor) -> torch.Tensor:
    """Logarithm SE(3) -> se(3) for rotations well below pi (adjacent video frames); T [..., 4, 4] -> [..., 6]."""
    R, t = T[..., :3, :3], T[..., :3, 3]
    cos = ((R.diagonal(dim1=-2, dim2=-1).sum(-1) - 1) / 2).clamp(-1 + 1e-7, 1 - 1e-7)
    th = torch.acos(cos)
    small = th < 1e-4
    k = torch.where(small, 0.5 + th * th / 12, th / (2 * torch.sin(th) + 1e-20))
    om = k[..., None] * torch.stack([R[..., 2, 1] - R[..., 1, 2], R[..., 0, 2] - R[..., 2, 0], R[..., 1, 0] - R[..., 0, 1]], -1)
    K = _hat(om)
    th2 = th * th
    A = torch.where(small, 1 - th2 / 6, torch.sin(th) / (th + 1e-20))
    B = torch.where(small, 0.5 - th2 / 24, (1 - torch.cos(th)) / (th2 + 1e-20))
    coef = torch.where(small, torch.full_like(th, 1.0 / 12), (1 - A / (2 * B)) / (th2 + 1e-20))
    Vinv = torch.eye(3, dtype=T.dtype, device=T.device) - 0.5 * K + coef[..., None, None] * _mm(K, K)
    return torch.cat([_mm(Vinv, t[..., None])[..., 0], om], -1)


class ImplicitCRF(nn.Module):
    """Shared implicit camera response F_theta: log-exposure u -> LDR value per channel, sampled on K knots so the
    rasterizer can evaluate it per pixel (`crf_table`).  Monotone by construction (positive increments)."""

    def __init__(self, K: int = 256, u_range=(-6.0, 3.0), hidden: int = 32):
        super().__init__()
        self.K, self.u_range = K, (float(u_range[0]), float(u_range[1]))
        self.net = nn.Sequential(nn.Linear(1, hidden), nn.Tanh(), nn.Linear(hidden, hidden), nn.Tanh(), nn.Linear(hidden, 3))
        self.register_buffer("knots", torch.linspace(self.u_range[0], self.u_range[1], K)[:, None])

    def _mlp(self, x: torch.Tensor) -> torch.Tensor:
        """self.net(x), written as broadcast multiplies and sums (the layers are 1 -> h -> h -> 3 on K points: nothing a
        BLAS call is good for).  Not only speed: inside the captured multi-frame step (graphs.GraphedStep) round 5 saw
        tensors next to a BLAS call hold wrong values from the second replay on.  Round 6 could not reproduce that with a
        BLAS call alone (scripts/repro/graph_torch_sum.py: right on every replay) -- the failure belongs to the capture of
        that long torch step as a whole, DESIGN.md 4.11 -- but nothing is lost by keeping the arithmetic BLAS-free."""
        for layer in self.net:
            if isinstance(layer, nn.Linear):
                x = (x.unsqueeze(-2) * layer.weight).sum(-1) + layer.bias     # [K, out] = sum_in x[K, 1, in] * W[out, in]
            else:
                x = layer(x)
        return x

    def table(self) -> torch.Tensor:
        inc = torch.nn.functional.softplus(self._mlp(self.knots))       # [K,3] positive slopes
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
                 rasterizer_factory: Callable = GaussianRasterizer, frame_times: Optional[torch.Tensor] = None,
                 window_from_exposure: bool = True, window_scale: float = 1.0):
        """frame_times [n_frames]: mid-exposure time of every captured frame in knot units (default: the middle of
        the i-th interval the trajectory defines).  window_from_exposure (default, the figure's "Exposure time range"): the
        exposure window of frame i is dt_i * window_scale knot intervals long (dt_i = exp(log_exposure_i), window_scale =
        knot intervals per unit of exposure time, i.e. the capture's knot rate; default 1: a frame of unit exposure -- the
        initial value of every dt_i -- spans one knot interval); False: one knot interval whatever the exposure.
        Raises ValueError when a frame's mid-exposure time lies outside trajectory.t_range (poses beyond it would be
        extrapolated from the last segment) -- e.g. more frames than a cubic spline of J knots has segments (J - 3)."""
        super().__init__()
        self.trajectory = trajectory
        self.crf = crf if crf is not None else ImplicitCRF()
        self.log_exposure = nn.Parameter(torch.zeros(n_frames))          # dt_i = exp(log_exposure_i)
        self.W, self.H, self.tanfovx, self.tanfovy = W, H, tanfovx, tanfovy
        self.n_virtual, self.blur_domain, self.sh_degree = n_virtual, blur_domain, sh_degree
        self._factory = rasterizer_factory
        t0 = trajectory.t_range[0]
        if frame_times is None:
            frame_times = t0 + 0.5 + torch.arange(n_frames, dtype=torch.float32)
        ft = torch.as_tensor(frame_times, dtype=torch.float32).clone().reshape(-1)
        t_lo, t_hi = trajectory.t_range
        if ft.numel() != n_frames or (n_frames and (float(ft.min()) < t_lo or float(ft.max()) > t_hi)):
            raise ValueError(f"HDRBlurFormation: {n_frames} frame time(s) {ft.tolist()} must lie inside the trajectory's "
                             f"t_range [{t_lo}, {t_hi}] ({trajectory.kind} spline over {trajectory.base.shape[0]} knots)")
        self.register_buffer("frame_times", ft)
        # (built once: filling a device matrix element by element from host scalars is a run of host-to-device copies, which
        # a graph capture does not allow)
        self.register_buffer("proj", projection_matrix(tanfovx, tanfovy))
        self.window_from_exposure, self.window_scale = bool(window_from_exposure), float(window_scale)

    def window(self, i: int) -> torch.Tensor:
        """Length of frame i's exposure window in knot intervals (a tensor: it carries dL/d(dt_i)'s blur term)."""
        if self.window_from_exposure:
            return torch.exp(self.log_exposure[i]) * self.window_scale
        return torch.ones((), dtype=self.log_exposure.dtype, device=self.log_exposure.device)

    def cameras(self, i: int):
        """(viewmatrices [N,4,4], projmatrices [N,4,4], camposes [N,3]) in the rasterizer's transposed convention."""
        times = self.trajectory.window_times(self.frame_times[i].to(self.log_exposure.dtype), self.window(i), self.n_virtual)
        w2c = self.trajectory.pose_at(times)
        proj = self.proj.to(w2c.dtype)
        full = _mm(proj[None], w2c)
        campos = -_mm(w2c[:, :3, :3].transpose(1, 2), w2c[:, :3, 3:])[..., 0]
        return w2c.transpose(1, 2).contiguous(), full.transpose(1, 2).contiguous(), campos.contiguous()

    def cameras_all(self):
        """The cameras of EVERY captured frame from one pass over the spline: (viewmatrices [F,N,4,4], projmatrices
        [F,N,4,4], camposes [F,N,3]).  The pose arithmetic is a few hundred tiny tensor operations whatever the number of
        poses, and its cost is the host's launch time (about 9 ms per call forward + backward in eager mode, against
        0.5 ms for the rasterizer at 20 k Gaussians): a step over several frames computes them once, hands frame i its
        slice (`forward(i, ..., cameras=...)`) and back-propagates the summed loss once."""
        dt = self.log_exposure.dtype
        n = self.n_virtual
        width = (torch.exp(self.log_exposure) * self.window_scale if self.window_from_exposure
                 else torch.ones_like(self.log_exposure))
        s = (torch.arange(n, dtype=dt, device=width.device) + 0.5) / n - 0.5
        times = self.frame_times.to(dt)[:, None] + s[None, :] * width[:, None]                 # [F, N]
        w2c = self.trajectory.pose_at(times.reshape(-1))
        proj = self.proj.to(w2c.dtype)
        full = _mm(proj[None], w2c)
        campos = -_mm(w2c[:, :3, :3].transpose(1, 2), w2c[:, :3, 3:])[..., 0]
        F = times.shape[0]
        return (w2c.transpose(1, 2).reshape(F, n, 4, 4).contiguous(), full.transpose(1, 2).reshape(F, n, 4, 4).contiguous(),
                campos.reshape(F, n, 3).contiguous())

    def forward(self, i: int, means3D, opacities, shs, scales, rotations, bg=None, cameras=None):
        """`cameras`: the result of cameras_all() (frame i takes its slice) instead of a spline pass of its own."""
        V, PV, Cp = self.cameras(i) if cameras is None else (cameras[0][i], cameras[1][i], cameras[2][i])
        dev = means3D.device
        bg = torch.zeros(3, device=dev) if bg is None else bg
        settings = GaussianRasterizationSettings(
            image_height=self.H, image_width=self.W, tanfovx=self.tanfovx, tanfovy=self.tanfovy, bg=bg,
            scale_modifier=1.0, viewmatrix=V[0], projmatrix=PV[0], sh_degree=self.sh_degree, campos=Cp[0],
            prefiltered=False, debug=False, exposure=torch.exp(self.log_exposure[i]), crf_table=self.crf.table(),
            crf_range=self.crf.u_range, viewmatrices=V, projmatrices=PV, camposes=Cp, blur_domain=self.blur_domain)
        means2D = torch.zeros_like(means3D, requires_grad=means3D.requires_grad)
        per_frame = getattr(self._factory, "for_frame", None)     # (FrameRasterizers: one persistent rasterizer per frame)
        rast = per_frame(i, settings) if per_frame is not None else self._factory(settings)
        ldr, radii, hdr = rast(means3D, means2D, opacities, shs=shs, scales=scales, rotations=rotations)
        return ldr, hdr, radii, means2D


class FrameRasterizers:
    """`rasterizer_factory` for HDRBlurFormation that keeps ONE sync-free GaussianRasterizer per captured frame -- what a
    captured step needs (graphs.GraphedStep takes persistent rasterizers with a fixed binning capacity; the formation's
    default makes a fresh, synchronous one per call):

        frames = FrameRasterizers(capacity=2_000_000)
        model = HDRBlurFormation(trajectory, n_frames, ..., rasterizer_factory=frames)
        step = GraphedStep(fn, frames.rasterizers(n_frames), params=[...])   # after one eager fn() has created them

    The settings of every call are swapped into the frame's rasterizer."""

    def __init__(self, capacity: int, **rasterizer_kwargs):
        self.capacity, self.kw, self._by_frame = int(capacity), rasterizer_kwargs, {}

    def for_frame(self, i: int, settings):
        r = self._by_frame.get(int(i))
        if r is None:
            r = self._by_frame[int(i)] = GaussianRasterizer(settings, capacity=self.capacity, **self.kw)
        r.raster_settings = settings
        return r

    def __call__(self, settings):          # (a caller that does not say which frame: frame 0's rasterizer)
        return self.for_frame(0, settings)

    def rasterizers(self, n_frames: Optional[int] = None) -> list:
        """The rasterizers made so far, in frame order (n_frames given: all of them must exist)."""
        if n_frames is not None and any(i not in self._by_frame for i in range(n_frames)):
            raise RuntimeError("FrameRasterizers: run one eager step over all frames before asking for the rasterizers")
        return [self._by_frame[i] for i in sorted(self._by_frame)]


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
