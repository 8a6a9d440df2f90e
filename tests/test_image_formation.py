"""Image-formation module: host-side SE(3)/CRF math on CPU; the end-to-end step on the GPU."""

import pytest
import torch

from casualhdrsplat_amd import image_formation as IF
from casualhdrsplat_amd import synthetic as S


def test_se3_exp_log_roundtrip_and_group_properties():
    torch.manual_seed(0)
    xi = torch.randn(6, dtype=torch.float64) * 0.3
    T = IF.se3_exp(xi)
    R = T[:3, :3]
    assert torch.allclose(R @ R.t(), torch.eye(3, dtype=torch.float64), atol=1e-12)
    assert torch.det(R).item() == pytest.approx(1.0, abs=1e-12)
    assert torch.allclose(IF.se3_log(T), xi, atol=1e-9)
    assert torch.allclose(IF.se3_exp(torch.zeros(6, dtype=torch.float64)), torch.eye(4, dtype=torch.float64))
    # one-parameter subgroup: exp(s xi) exp(t xi) = exp((s+t) xi)
    assert torch.allclose(IF.se3_exp(0.3 * xi) @ IF.se3_exp(0.7 * xi), T, atol=1e-12)
    # tiny rotations go through the series branch and stay differentiable
    x = (torch.randn(6, dtype=torch.float64) * 1e-7).requires_grad_(True)
    IF.se3_exp(x).sum().backward()
    assert torch.isfinite(x.grad).all()


def test_trajectory_samples_lie_between_knots():
    knots = IF.knots_from_lookat(4, radius=0.2).double()
    traj = IF.TrajectorySpline(knots, kind="linear").double()
    P = traj.poses(1, 8)
    assert P.shape == (8, 4, 4)
    c = [-(p[:3, :3].t() @ p[:3, 3]) for p in P]                       # camera centres
    c0 = -(knots[1][:3, :3].t() @ knots[1][:3, 3])
    c1 = -(knots[2][:3, :3].t() @ knots[2][:3, 3])
    xs = torch.stack(c)[:, 0]
    assert torch.all(xs[1:] > xs[:-1]) and xs[0] > c0[0] and xs[-1] < c1[0]     # monotone along the arc, inside the window
    for p in P:
        assert torch.allclose(p[:3, :3] @ p[:3, :3].t(), torch.eye(3, dtype=torch.float64), atol=1e-10)
    # a knot correction moves only the windows that touch the knot
    with torch.no_grad():
        traj.delta[3, 0] = 0.1
    assert torch.allclose(traj.poses(1, 8), P) and not torch.allclose(traj.poses(2, 8), IF.TrajectorySpline(knots, kind="linear").double().poses(2, 8))


def _random_knots(J, seed=0, step=0.05):
    g = torch.Generator().manual_seed(seed)
    T = [torch.eye(4, dtype=torch.float64)]
    for _ in range(J - 1):
        T.append(IF.se3_exp(step * torch.randn(6, generator=g, dtype=torch.float64)) @ T[-1])
    return torch.stack(T)


def test_cubic_spline_is_c2_continuous_across_segments():
    """/root/reference/assets/pipeline.png: a camera motion SPLINE through four control knots.  Value, velocity and
    acceleration of the cumulative cubic B-spline agree on both sides of every segment boundary (autograd derivatives
    in float64, the two sides evaluated by different knot quadruples)."""
    traj = IF.TrajectorySpline(_random_knots(7, seed=3), kind="cubic").double()
    assert traj.t_range == (1.0, 5.0)

    def d012(t0):
        t = torch.tensor([t0], dtype=torch.float64, requires_grad=True)
        T = traj.pose_at(t)[0, :3, :].reshape(-1)
        d1 = torch.stack([torch.autograd.grad(T[k], t, create_graph=True)[0][0] for k in range(12)])
        d2 = torch.stack([torch.autograd.grad(d1[k], t, retain_graph=True)[0][0] for k in range(12)])
        return T.detach(), d1.detach(), d2.detach()

    for tb in (2.0, 3.0, 4.0):
        lo, hi = d012(tb - 1e-9), d012(tb + 1e-9)   # floor() puts the two sides into neighbouring segments
        assert torch.allclose(lo[0], hi[0], atol=1e-8)
        assert torch.allclose(lo[1], hi[1], atol=1e-7), (lo[1] - hi[1]).abs().max()
        assert torch.allclose(lo[2], hi[2], atol=1e-6), (lo[2] - hi[2]).abs().max()
        assert float(lo[2].abs().max()) > 1e-4      # (a real acceleration, not 0 == 0)
    # ... whereas the two-knot form is only C0: its velocity jumps at a knot
    lin = IF.TrajectorySpline(_random_knots(7, seed=3), kind="linear").double()
    t = torch.tensor([2.0 - 1e-9], dtype=torch.float64, requires_grad=True)
    v_lo = torch.autograd.grad(lin.pose_at(t)[0, 0, 3], t)[0]
    t = torch.tensor([2.0 + 1e-9], dtype=torch.float64, requires_grad=True)
    v_hi = torch.autograd.grad(lin.pose_at(t)[0, 0, 3], t)[0]
    assert abs(float(v_lo - v_hi)) > 1e-4


def test_cubic_spline_interpolates_knots_of_a_constant_velocity_motion_and_stays_in_se3():
    """Knot interpolation: a uniform B-spline reproduces linear data, so control knots exp(k xi) T_0 (constant body
    velocity) are passed through exactly, pose(t) = exp(t xi) T_0; rotations stay orthonormal for any knots; a knot
    correction reaches exactly the four segments that use the knot."""
    xi = torch.tensor([0.03, -0.01, 0.02, 0.01, 0.02, -0.015], dtype=torch.float64)
    T0 = IF.se3_exp(torch.tensor([0.2, 0.1, -0.3, 0.05, -0.02, 0.03], dtype=torch.float64))
    knots = torch.stack([IF.se3_exp(k * xi) @ T0 for k in range(6)])
    traj = IF.TrajectorySpline(knots, kind="cubic").double()
    ts = torch.tensor([1.0, 1.37, 2.0, 2.5, 3.0, 3.999, 4.0], dtype=torch.float64)
    P = traj.pose_at(ts)
    for k, t in enumerate(ts.tolist()):
        assert torch.allclose(P[k], IF.se3_exp(t * xi) @ T0, atol=1e-9), t
    rnd = IF.TrajectorySpline(_random_knots(6, seed=5, step=0.2), kind="cubic").double()
    Q = rnd.pose_at(torch.linspace(1.0, 4.0, 13, dtype=torch.float64))
    eye = torch.eye(3, dtype=torch.float64)
    # (1e-6: the module keeps its SfM knots in float32; the spline itself multiplies exact exponentials onto them)
    assert torch.allclose(Q[:, :3, :3] @ Q[:, :3, :3].transpose(1, 2), eye.expand(13, 3, 3), atol=1e-6)
    assert torch.allclose(Q[:, 3], torch.tensor([0.0, 0.0, 0.0, 1.0], dtype=torch.float64).expand(13, 4))
    before = rnd.pose_at(torch.tensor([1.5, 2.5, 3.5], dtype=torch.float64)).detach()
    with torch.no_grad():
        rnd.delta[0, 0] = 0.1      # knot 0 governs t in [1, 2) only
    after = rnd.pose_at(torch.tensor([1.5, 2.5, 3.5], dtype=torch.float64)).detach()
    assert not torch.allclose(after[0], before[0]) and torch.allclose(after[1:], before[1:])


def test_exposure_time_sets_the_blur_extent():
    """The figure's "exposure time range" arc: with window_from_exposure the virtual poses of frame i spread over
    dt_i * window_scale knot intervals around the frame's time stamp, so dt_i reaches the poses (a motion-blur term
    in dL/d dt_i) -- and does not when the window is pinned."""
    # a DEFAULT-constructed model (round 6: the defaults are the figure's -- cubic spline, window from the exposure, one knot
    # interval per unit of exposure time); only the frames' time stamps are given
    traj = IF.TrajectorySpline(_random_knots(6, seed=7)).double()
    assert traj.kind == "cubic"
    m = IF.HDRBlurFormation(traj, 2, 64, 48, 0.5, 0.4, n_virtual=5, crf=IF.ImplicitCRF(K=8),
                            frame_times=torch.tensor([2.2, 3.1])).double()
    assert m.window_from_exposure and m.window_scale == 1.0
    with torch.no_grad():
        m.log_exposure[0] = -0.5
    w = float(torch.exp(m.log_exposure[0].detach())) * 1.0
    times = traj.window_times(m.frame_times[0].double(), m.window(0), 5)
    assert times.min() > 2.2 - w / 2 and times.max() < 2.2 + w / 2
    assert float(times.mean()) == pytest.approx(2.2, abs=1e-6)
    assert float(times[-1] - times[0]) == pytest.approx(w * 4 / 5, rel=1e-9)
    V, PV, C = m.cameras(0)
    (C[-1] - C[0]).norm().backward()                      # the extent of the camera path during the exposure
    g = m.log_exposure.grad.clone()
    assert g[0] > 0 and g[1] == 0                         # a longer exposure = a longer path; frame 1 untouched
    # the extent is (to first order) proportional to the exposure time
    with torch.no_grad():
        m.log_exposure[0] = -0.5 + 1e-4
    C2 = m.cameras(0)[2]
    fd = ((C2[-1] - C2[0]).norm() - (C[-1] - C[0]).norm()).item() / 1e-4
    assert fd == pytest.approx(float(g[0]), rel=1e-3)
    pinned = IF.HDRBlurFormation(traj, 2, 64, 48, 0.5, 0.4, n_virtual=5, crf=IF.ImplicitCRF(K=8),
                                 frame_times=torch.tensor([2.2, 3.1]), window_from_exposure=False).double()
    Cp = pinned.cameras(0)[2]
    m.zero_grad(); pinned.zero_grad()
    (Cp[-1] - Cp[0]).norm().backward()
    assert pinned.log_exposure.grad is None or float(pinned.log_exposure.grad.abs().sum()) == 0


def test_frame_times_outside_the_trajectory_are_refused():
    """A cubic spline over J knots is defined on [1, J - 2]: frames beyond it (or more frames than its J - 3 segments, with
    the default time stamps) would be rendered from extrapolated poses without notice."""
    traj = IF.TrajectorySpline(_random_knots(5, seed=1))          # t_range [1, 3]: two segments
    IF.HDRBlurFormation(traj, 2, 64, 48, 0.5, 0.4)                 # default stamps 1.5, 2.5
    with pytest.raises(ValueError, match="t_range"):
        IF.HDRBlurFormation(traj, 3, 64, 48, 0.5, 0.4)             # 3.5 lies outside
    with pytest.raises(ValueError, match="t_range"):
        IF.HDRBlurFormation(traj, 1, 64, 48, 0.5, 0.4, frame_times=torch.tensor([0.5]))
    with pytest.raises(ValueError):
        IF.TrajectorySpline(_random_knots(3, seed=1))              # the default (cubic) kind needs four knots


def test_cameras_of_all_frames_in_one_pass_equal_the_per_frame_cameras():
    """HDRBlurFormation.cameras_all(): one pass over the spline for every captured frame (the pose arithmetic's cost is the
    host's launch time, independent of the number of poses) gives frame i exactly cameras(i), values and gradients --
    also the gradient of the exposure times through the window length."""
    torch.manual_seed(5)
    for wfe in (False, True):
        m = IF.HDRBlurFormation(IF.TrajectorySpline(IF.knots_from_lookat(7, radius=0.25), kind="cubic"), 4, 64, 48, 0.5, 0.4,
                                n_virtual=5, crf=IF.ImplicitCRF(K=16), sh_degree=1, window_from_exposure=wfe, window_scale=0.6,
                                rasterizer_factory=None)
        with torch.no_grad():
            m.log_exposure.copy_(torch.tensor([0.0, -0.5, 0.4, 0.1]))
            m.trajectory.delta.normal_(0, 0.01)
        w = [torch.randn(5, 4, 4), torch.randn(5, 4, 4), torch.randn(5, 3)]
        allc = m.cameras_all()
        assert allc[0].shape == (4, 5, 4, 4) and allc[2].shape == (4, 5, 3)
        for i in range(4):
            one = m.cameras(i)
            assert all(torch.equal(a[i], b) for a, b in zip(allc, one))
        g_all = torch.autograd.grad(sum((a[2] * x).sum() for a, x in zip(allc, w)), [m.trajectory.delta, m.log_exposure],
                                    allow_unused=True)
        g_one = torch.autograd.grad(sum((a * x).sum() for a, x in zip(m.cameras(2), w)), [m.trajectory.delta, m.log_exposure],
                                    allow_unused=True)
        assert torch.allclose(g_all[0], g_one[0], rtol=1e-5, atol=1e-7)
        if wfe:
            assert g_one[1].abs().sum() > 0 and torch.allclose(g_all[1], g_one[1], rtol=1e-5, atol=1e-7)


def test_crf_table_is_monotone_and_normalised():
    crf = IF.ImplicitCRF(K=64)
    tab = crf.table()
    assert tab.shape == (3, 64)
    assert torch.all(tab[:, 1:] > tab[:, :-1])
    assert torch.allclose(tab[:, 0], torch.zeros(3)) and torch.allclose(tab[:, -1], torch.ones(3), atol=1e-6)
    tab.sum().backward()
    assert all(p.grad is not None for p in crf.parameters())


def test_camera_matrices_follow_the_rasterizer_convention():
    W, H = 160, 96
    cam = S.make_camera(W, H)
    traj = IF.TrajectorySpline(torch.eye(4)[None].repeat(2, 1, 1), kind="linear")
    model = IF.HDRBlurFormation(traj, 1, W, H, cam.tanfovx, cam.tanfovy, n_virtual=3, crf=IF.ImplicitCRF(K=16))
    V, PV, C = model.cameras(0)
    for k in range(3):  # identity trajectory == the synthetic default camera
        assert torch.allclose(V[k], cam.viewmatrix, atol=1e-6)
        assert torch.allclose(PV[k], cam.projmatrix, atol=1e-5)
        assert torch.allclose(C[k], cam.campos, atol=1e-6)


@pytest.mark.gpu
def test_end_to_end_step_reaches_every_learnable():
    """One HDR-deblur training step through the HIP rasterizer: the photometric loss back-propagates into the
    Gaussians, the trajectory knots (camera motion), the exposure and the CRF network -- the four learnables of
    /root/reference/assets/pipeline.png -- and a few Adam steps on (pose, exposure) reduce it."""
    dev = "cuda"
    W, H, P = 160, 96, 4000
    sc = S.make_scene(P, W, H, 1, seed=21, hdr=True)
    cam = sc.camera
    knots = IF.knots_from_lookat(3, radius=0.03)
    torch.manual_seed(0)

    def build():
        traj = IF.TrajectorySpline(knots, kind="linear")
        m = IF.HDRBlurFormation(traj, 2, W, H, cam.tanfovx, cam.tanfovy, n_virtual=4, crf=IF.ImplicitCRF(K=64), sh_degree=1)
        return m.to(dev)

    leaves = {k: getattr(sc, k).to(dev).requires_grad_(True) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
    target_model = build()
    with torch.no_grad():
        target_model.trajectory.delta[1, 0] = 0.02      # the "true" camera moved a little
        target_model.log_exposure[0] = 0.3
        target, _, _, _ = target_model(0, *[leaves[k].detach() for k in ("means3D", "opacities", "shs", "scales", "rotations")])
    model = build()
    model.crf.load_state_dict(target_model.crf.state_dict())
    ldr, hdr, radii, m2d = model(0, leaves["means3D"], leaves["opacities"], leaves["shs"], leaves["scales"], leaves["rotations"])
    assert ldr.shape == (3, H, W) and hdr.shape == (3, H, W) and int((radii > 0).sum()) > 0
    loss0 = ((ldr - target) ** 2).mean()
    loss0.backward()
    assert all(torch.isfinite(v.grad).all() and v.grad.abs().sum() > 0 for v in leaves.values())
    assert model.trajectory.delta.grad[:2].abs().sum() > 0 and torch.all(model.trajectory.delta.grad[2] == 0)
    assert model.log_exposure.grad[0] != 0 and model.log_exposure.grad[1] == 0
    assert sum(p.grad.abs().sum() for p in model.crf.parameters()) > 0
    opt = torch.optim.Adam([model.trajectory.delta, model.log_exposure], lr=5e-3)
    for _ in range(40):
        opt.zero_grad()
        ldr, _, _, _ = model(0, *[leaves[k].detach() for k in ("means3D", "opacities", "shs", "scales", "rotations")])
        loss = ((ldr - target) ** 2).mean()
        loss.backward()
        opt.step()
    assert loss.item() < 0.5 * loss0.item()


class _OracleRasterizer:
    """Stand-in for GaussianRasterizer inside HDRBlurFormation: the float64 pure-PyTorch autograd rasterizer of oracle/
    (test infrastructure).  Same call shape, so the module's own TrajectorySpline / exposure / ImplicitCRF feed it."""

    def __init__(self, settings):
        self.s = settings

    def __call__(self, means3D, means2D, opacities, shs=None, scales=None, rotations=None):
        from oracle import torch_rasterizer as TR
        s = self.s
        views = [TR.View(s.image_width, s.image_height, s.tanfovx, s.tanfovy, s.viewmatrices[k], s.projmatrices[k],
                         s.camposes[k]) for k in range(s.viewmatrices.shape[0])]
        ldr, hdr = TR.rasterize_hdr(views, means3D, opacities, s.sh_degree, s.bg, s.exposure, s.crf_table, s.crf_range,
                                    blur_domain=s.blur_domain, shs=shs, scales=scales, rotations=rotations)
        return ldr, torch.zeros(means3D.shape[0], dtype=torch.int32), hdr


@pytest.mark.gpu
@pytest.mark.parametrize("free", [False, True], ids=["yaw_knots", "free_knots"])
@pytest.mark.parametrize("dom", ["ldr", "hdr"])
@pytest.mark.parametrize("kind", ["linear", "cubic"])
def test_image_formation_matches_the_fp64_oracle_through_the_same_modules(dom, kind, free):
    """SURVEY.md 8(f) n2, oracle-backed: HDRBlurFormation on the MI355X (one HIP rasterizer call: N virtual poses,
    exposure, CRF, blur average, pose gradients) against the float64 autograd rasterizer driven by the SAME
    TrajectorySpline / exposure / ImplicitCRF modules -- blurred LDR image, mean radiance, and the gradients that reach
    the trajectory knots (camera motion), the exposure time and the CRF network's parameters.
    kind="cubic" (round 5): the figure's model in full -- a cumulative cubic B-spline over FOUR control knots, the virtual
    poses spread over the exposure window dt_i * window_scale around the frame's time stamp, so dL/d dt_i carries the
    motion-blur term (through the rasterizer's pose gradients) next to the brightness term."""
    dev = "cuda"
    W, H, P, deg, n_virtual = 112, 80, 1500, 1, 3
    cubic = kind == "cubic"
    if free:
        # the figure's free trajectory: a 6-DoF base camera (roll, pitch and yaw up to +-pi), the control knots roll, pitch
        # and yaw about it and shift -- every entry of every knot's rotation populated (the knots of
        # knots_from_lookat rotate about y only)
        base = S.random_camera(W, H, 11)
        sc = S.make_scene(P, W, H, deg, seed=33, hdr=True, place_in=base)
        poses = S.perturbed_poses(base, 5 if cubic else 3, seed=4, rot_step_deg=0.5 if cubic else 0.05,
                                  step=0.1 if cubic else 0.01)
        knots = torch.stack([S.camera_w2c(c).float() for c in poses])
    else:
        sc = S.make_scene(P, W, H, deg, seed=33, hdr=True)
        knots = IF.knots_from_lookat(5 if cubic else 3, radius=0.4 if cubic else 0.04)   # (cubic: ~3 px of blur per window)
    cam = sc.camera
    torch.manual_seed(3)
    crf0 = IF.ImplicitCRF(K=48)

    def build(dtype, device, factory=None):
        kw = {} if factory is None else dict(rasterizer_factory=factory)
        if cubic:   # frame 0 exposed around t = 1.6 (segment of knots 0..3), 0.67 * 1.2 = 0.8 knot intervals long
            kw.update(frame_times=torch.tensor([1.6, 2.5]), window_from_exposure=True, window_scale=1.2)
        m = IF.HDRBlurFormation(IF.TrajectorySpline(knots, kind=kind), 2, W, H, cam.tanfovx, cam.tanfovy, n_virtual=n_virtual,
                                crf=IF.ImplicitCRF(K=48), blur_domain=dom, sh_degree=deg, **kw)
        m.crf.load_state_dict(crf0.state_dict())
        with torch.no_grad():
            m.trajectory.delta[0] = torch.tensor([0.01, -0.004, 0.006, 0.002, -0.003, 0.001])
            m.trajectory.delta[1] = torch.tensor([-0.008, 0.005, 0.0, -0.001, 0.002, 0.004])
            if cubic:
                m.trajectory.delta[2] = torch.tensor([0.004, 0.003, -0.005, 0.002, 0.001, -0.002])
                m.trajectory.delta[3] = torch.tensor([-0.003, -0.006, 0.002, -0.002, 0.003, 0.001])
            m.log_exposure[0] = -0.4
        return m.to(device=device, dtype=dtype)

    names = ("means3D", "opacities", "shs", "scales", "rotations")
    gl = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1))
    gh = 0.05 * torch.randn(3, H, W, generator=torch.Generator().manual_seed(2))

    def run(m, dtype, device):
        leaves = [getattr(sc, k).to(device=device, dtype=dtype).requires_grad_(True) for k in names]
        ldr, hdr, _, _ = m(0, *leaves)
        ((ldr * gl.to(device=device, dtype=dtype)).sum() + (hdr * gh.to(device=device, dtype=dtype)).sum()).backward()
        grads = {"delta": m.trajectory.delta.grad, "log_exposure": m.log_exposure.grad,
                 "crf": torch.cat([p.grad.reshape(-1) for p in m.crf.parameters()]),
                 "means3D": leaves[0].grad, "shs": leaves[2].grad}
        return ldr.detach().cpu().double(), hdr.detach().cpu().double(), {k: v.detach().cpu().double() for k, v in grads.items()}

    ldr_g, hdr_g, g_g = run(build(torch.float32, dev), torch.float32, dev)
    ldr_o, hdr_o, g_o = run(build(torch.float64, "cpu", _OracleRasterizer), torch.float64, "cpu")
    bad = ((ldr_g - ldr_o).abs() > 1e-4 * ldr_o.abs().clamp_min(1e-2)).any(dim=0)
    assert int(bad.sum()) <= 2, int(bad.sum())           # a threshold flip between fp32 and fp64 decisions, at most
    assert ((hdr_g - hdr_o).abs() > 1e-4 * hdr_o.abs().clamp_min(1e-2)).any(dim=0).sum() <= 2
    for k in ("delta", "log_exposure", "crf", "means3D", "shs"):
        a, b = g_g[k], g_o[k]
        scale = float(b.abs().max())
        assert scale > 0, k
        assert float((a - b).abs().max()) <= 3e-4 * scale, (k, float((a - b).abs().max()) / scale)
    if cubic:
        # all four knots of the segment receive a gradient, the fifth none
        assert all(float(g_g["delta"][j].abs().max()) > 0 for j in range(4))
        assert float(g_o["delta"][4].abs().max()) == 0 and float(g_g["delta"][4].abs().max()) == 0
        # dL/d log dt_0 = brightness term + window term.  The window term on its own, on both sides: the same modules with the
        # window PINNED to the same length give the brightness term alone; the difference is what reached dt_0 through the
        # virtual poses (the rasterizer's pose gradients) -- compared HIP against float64 directly, so the
        # brightness term's size cannot hide it
        def pinned(dtype, device, factory=None):
            m = build(dtype, device, factory)
            m.window_from_exposure = False
            w0 = float(torch.exp(m.log_exposure[0].detach())) * 1.2
            m.window = lambda i, w0=w0, m=m: torch.full((), w0 if i == 0 else 1.0, dtype=m.log_exposure.dtype,
                                                        device=m.log_exposure.device)
            return run(m, dtype, device)[2]["log_exposure"][0]

        win_o = float(g_o["log_exposure"][0] - pinned(torch.float64, "cpu", _OracleRasterizer))
        win_g = float(g_g["log_exposure"][0] - pinned(torch.float32, dev))
        scale_e = abs(float(g_o["log_exposure"][0]))
        assert abs(win_o) > 1e-3 * scale_e, (win_o, scale_e)                    # a real term, far above fp32 noise of the total
        assert abs(win_g - win_o) <= 0.03 * abs(win_o) + 3e-5 * scale_e, (dom, win_g, win_o, scale_e)
    else:
        assert float(g_o["delta"][2].abs().max()) == 0 and float(g_g["delta"][2].abs().max()) == 0   # knot 2 is outside frame 0
    assert float(g_g["log_exposure"][1]) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("graph", [False, True])
def test_joint_optimisation_example_recovers_exposures_and_improves_the_fit(graph):
    """examples/train_synthetic.py -- the loop a trainer runs around the rasterizer: blurred LDR observations rendered from a
    true cloud / spline trajectory / exposure times / response curve, then radiance, opacities, exposure times and the
    trajectory optimised jointly from perturbed starts through the HIP kernels.  A short run must cut the photometric
    loss and move the exposure times (whose gradient includes the blur-extent term) towards the truth."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("train_synthetic", os.path.join(root, "examples", "train_synthetic.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    # graph=True: the step's gradient computation recorded once (graphs.GraphedStep, one sync-free rasterizer per frame:
    # image_formation.FrameRasterizers) and replayed -- same trajectory of the optimisation, a third of the time per step
    r = mod.run(P=5000, W=192, H=128, frames=3, virtual=4, steps=120, seed=3, quiet=True, graph=graph)
    f, l = r["first"], r["last"]
    assert abs(f["loss"] - 0.05) < 0.05 and all(0 <= h["loss"] < 1.0 for h in r["history"])   # (a captured loss value stays a loss value)
    assert l["loss"] < 0.5 * f["loss"], (f, l)
    assert l["psnr"] > f["psnr"] + 3.0, (f, l)
    assert l["exposure_log_err"] < 0.5 * f["exposure_log_err"], (f, l)
    assert all(torch.isfinite(torch.tensor([h["loss"] for h in r["history"]])))


@pytest.mark.gpu
def test_formation_step_is_capturable_as_one_hip_graph():
    """The whole gradient computation of a multi-frame step -- the spline pass for all frames (cameras_all), one sync-free
    rasterizer call per captured frame, the backward of the summed loss into the Gaussians, the trajectory knots and the
    exposure times -- recorded once by graphs.GraphedStep and replayed: the replay gives the eager step's gradients, also
    after the parameters were changed in place (what an optimizer does).  The pose arithmetic reads nothing on the host."""
    from casualhdrsplat_amd import GaussianRasterizer
    from casualhdrsplat_amd.graphs import GraphedStep
    dev = "cuda"
    W, H, P, F = 160, 96, 4000, 3
    sc = S.make_scene(P, W, H, 1, seed=8, hdr=True)
    cam = sc.camera
    frames = IF.FrameRasterizers(capacity=150000)   # one persistent sync-free rasterizer per captured frame
    torch.manual_seed(2)
    m = IF.HDRBlurFormation(IF.TrajectorySpline(IF.knots_from_lookat(F + 3, radius=0.2), kind="cubic"), F, W, H, cam.tanfovx,
                            cam.tanfovy, n_virtual=3, crf=IF.ImplicitCRF(K=32), sh_degree=1, window_from_exposure=True,
                            window_scale=0.5, rasterizer_factory=frames).to(dev)
    leaves = {k: getattr(sc, k).to(dev).requires_grad_(True) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
    targets = [torch.rand(3, H, W, device=dev) for _ in range(F)]
    params = list(leaves.values()) + [m.trajectory.delta, m.log_exposure]

    def grads_of_step():
        for p in params:
            p.grad = None
        cams = m.cameras_all()
        loss, radii = 0.0, []
        for i in range(F):
            ldr, _, rad, _ = m(i, *[leaves[k] for k in ("means3D", "opacities", "shs", "scales", "rotations")], cameras=cams)
            loss = loss + ((ldr - targets[i]) ** 2).mean()
            radii.append(rad)
        loss.backward()
        return (loss.detach(),) + tuple(radii)

    def eager():
        out = grads_of_step()
        torch.cuda.synchronize()
        return [p.grad.clone() for p in params] + [o.clone() for o in out]

    want = eager()
    for p in params:
        p.grad = None
    g = GraphedStep(grads_of_step, frames.rasterizers(F), params=params)
    out = g.step()
    assert len(g.check_overflow()) == F
    # gradients, the loss and the N-pose radii (max over the poses): bit for bit the eager step's.  (Round 5: the pose
    # gradients and the radii used to be cleared by hipMemsetAsync; as memset nodes of a captured step those raced with the
    # kernels behind them -- two replays of the same inputs gave different camera gradients.  Kernels clear them now.)
    for a, b in zip(list(g.grads) + list(out), want):
        assert torch.equal(a, b)
    assert int((out[1] > 0).sum()) > 0
    with torch.no_grad():      # an optimizer step in place: the replay follows, the eager step agrees
        m.trajectory.delta.add_(0.002 * torch.randn_like(m.trajectory.delta))
        m.log_exposure.add_(0.1)
        leaves["shs"].mul_(0.9)
    for rep in range(3):   # (several replays: a race shows up now and then, not every time)
        out = g.step()
        g.check_overflow()
        got = [x.clone() for x in list(g.grads) + list(out)]
        if rep == 0:
            want2 = eager()
            assert float((want2[5] - want[5]).abs().sum()) > 0     # (the trajectory gradient did change)
        for a, b in zip(got, want2):
            assert torch.equal(a, b), rep


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["linear", "cubic"])
def test_fused_spline_kernel_matches_the_tensor_implementation(kind):
    """hs_spline_poses (spline.hip: one thread per sample time, float64 dual numbers) against TrajectorySpline.pose_at written
    with tensor operations in float64 on the CPU: poses, and the gradients of a random functional of them with respect to the
    knots and to the sample times -- at generic knots, at untouched knots (delta = 0: the small-angle series of exp) and with
    two identical neighbouring knots (the clamped arc-cosine of log)."""
    torch.manual_seed(11)
    J = 7
    base = IF.knots_from_lookat(J, radius=0.3)
    base[3] = base[2]                                       # two identical neighbours: log of the identity
    for scale in (0.0, 0.02):
        ref = IF.TrajectorySpline(base.double(), kind=kind).double()
        ref.fused = False
        gpu = IF.TrajectorySpline(base, kind=kind).cuda()
        delta0 = scale * torch.randn(J, 6, dtype=torch.float64)
        with torch.no_grad():
            ref.delta.copy_(delta0)
            gpu.delta.copy_(delta0.float())
        lo, hi = ref.t_range
        t_ref = (lo + (hi - lo) * torch.rand(23, dtype=torch.float64)).requires_grad_(True)
        t_gpu = t_ref.detach().float().cuda().requires_grad_(True)
        w = torch.randn(23, 4, 4, dtype=torch.float64)
        a = ref.pose_at(t_ref)
        b = gpu.pose_at(t_gpu)
        assert b.shape == (23, 4, 4) and b.dtype == torch.float32
        assert torch.allclose(b.cpu().double(), a, rtol=1e-5, atol=2e-6)
        (a * w).sum().backward()
        (b * w.float().cuda()).sum().backward()
        for g_ref, g_gpu in ((ref.delta.grad, gpu.delta.grad), (t_ref.grad, t_gpu.grad)):
            assert torch.isfinite(g_gpu).all()
            assert torch.allclose(g_gpu.cpu().double(), g_ref, rtol=2e-4, atol=2e-5 * float(g_ref.abs().max())), (kind, scale)
        assert float(gpu.delta.grad.abs().sum()) > 0 and float(t_gpu.grad.abs().sum()) > 0
