"""Independent pinning of the derivatives: the C oracle's hand-derived backward (a10-a12) against
torch.autograd through the pure-PyTorch rasterizer in float64, plus gradcheck of the latter, the HDR
tone-map, N-pose identities."""
import numpy as np
import pytest
import torch

import helpers as Hh
from casualhdrsplat_amd import synthetic as S
from oracle import torch_rasterizer as TR


def torch_view(cam, dt):
    return TR.View(cam.W, cam.H, cam.tanfovx, cam.tanfovy, cam.viewmatrix.to(dt), cam.projmatrix.to(dt), cam.campos.to(dt))


def torch_run(sc, dt, cam=None, dL=None):
    cam = cam or sc.camera
    leaves = {k: getattr(sc, k).to(dt).clone().requires_grad_(True) for k in ["means3D", "opacities", "shs", "scales", "rotations"]}
    m2d = torch.zeros(sc.means3D.shape[0], 3, dtype=dt, requires_grad=True)
    color, st = TR.rasterize(torch_view(cam, dt), leaves["means3D"], leaves["opacities"], sc.sh_degree, sc.bg,
                             shs=leaves["shs"], scales=leaves["scales"], rotations=leaves["rotations"], means2D=m2d,
                             return_state=True)
    dL = sc.dL_dimage if dL is None else dL
    (color * dL.to(dt)).sum().backward()
    leaves["means2D"] = m2d
    return color.detach(), st, {k: v.grad.numpy() for k, v in leaves.items()}


@pytest.mark.parametrize("cam_seed", [None, 0, 1], ids=["default_camera", "free_camera_0", "free_camera_1"])
@pytest.mark.parametrize("P,W,H,deg,seed", [(300, 64, 48, 3, 3), (500, 80, 72, 1, 4), (400, 56, 56, 0, 5)])
def test_c_oracle_matches_fp64_autograd(oracle, P, W, H, deg, seed, cam_seed):
    """`cam_seed`: a free 6-DoF camera (synthetic.random_camera: roll, pitch and yaw up to +-pi, every entry of the view
    matrix populated) with the cloud laid out in front of it, instead of the identity-rotation default view."""
    sc = S.make_scene(P, W, H, deg, seed=seed, place_in=None if cam_seed is None else S.random_camera(W, H, cam_seed))
    f, b = Hh.run_oracle(oracle, sc)
    color, st, g = torch_run(sc, torch.float64)
    # integer structure identical
    assert np.array_equal(st["point_list"].numpy(), f["point_list"].astype(np.int64))
    assert np.array_equal(st["ranges"].numpy(), f["ranges"].astype(np.int64))
    assert np.array_equal(st["pre"]["radii"].numpy(), f["radii"])
    assert (st["n_contrib"].numpy() != f["n_contrib"]).sum() == 0
    # (free camera: the fp32 projection carries the rounding of a full 3 x 3 rotation and a translation of a few units --
    #  1.3e-5 measured against 7e-6 under the identity view)
    assert Hh.rel_err(f["color"], color.numpy(), 1e-2)[0] < (1e-5 if cam_seed is None else 3e-5)
    assert Hh.rel_err(f["final_T"], st["final_T"].detach().numpy(), 1e-3)[0] < 1e-4
    # derivatives: fp32 hand-derived vs fp64 autograd.  The per-pixel T/(1-alpha) recurrences make a small
    # tail of elements fp32-ill-conditioned (same for any fp32 implementation), hence frac + max bounds.
    for k, ok in [("means3D", "dL_dmeans3D"), ("means2D", "dL_dmeans2D"), ("opacities", "dL_dopacity"),
                  ("shs", "dL_dshs"), ("scales", "dL_dscales"), ("rotations", "dL_drots")]:
        ref = g[k].reshape(b[ok].shape)
        mx, frac = Hh.rel_err(b[ok], ref, Hh.grad_floor(ref))
        assert frac < 2e-2 and mx < 2e-2, (k, mx, frac)
        l2 = np.linalg.norm(b[ok].astype(np.float64) - ref) / max(np.linalg.norm(ref), 1e-30)
        assert l2 < 2e-5, (k, l2)


@pytest.mark.parametrize("P,W,H,deg,seed,max_flips,cam_seed", [(3000, 160, 112, 3, 6, 0, None), (20000, 320, 240, 2, 7, 16, None),
                                                               (3000, 160, 112, 3, 6, 4, 2)])
def test_c_oracle_matches_fp64_autograd_at_scale(oracle, P, W, H, deg, seed, max_flips, cam_seed):
    """The same pinning beyond toy size (the independent twin used to meet the oracle at <= 500
    Gaussians only): thousands of Gaussians, tile lists hundreds of entries long.  Integer structure identical; on a frame
    this size float64 and float32 decide a handful of thresholds differently (pixels inside the oracle's guard band) -- the
    Gaussians on those pixels' tile lists are then left out of the comparison, every other row is held to the bar."""
    sc = S.make_scene(P, W, H, deg, seed=seed, place_in=None if cam_seed is None else S.random_camera(W, H, cam_seed))
    f, b = Hh.run_oracle(oracle, sc)
    color, st, g = torch_run(sc, torch.float64)
    assert np.array_equal(st["point_list"].numpy(), f["point_list"].astype(np.int64))
    assert np.array_equal(st["ranges"].numpy(), f["ranges"].astype(np.int64))
    assert np.array_equal(st["pre"]["radii"].numpy(), f["radii"])
    # a differing decision shows as another contributor count, or (a skipped entry in the middle of the list) as a final
    # transmittance off by >= 1/255 relative; identical decisions leave it within ~1e-6 here
    Tf = st["final_T"].detach().numpy()
    differ = (st["n_contrib"].numpy() != f["n_contrib"]) | (np.abs(f["final_T"] - Tf) > 2e-3 * np.maximum(Tf, 1e-4))
    assert differ.sum() <= max_flips, int(differ.sum())
    # ... and only where the fp32 oracle itself calls the decision fragile (float64 against float32: a wider band than the
    # 1e-5 the HIP tests use between two fp32 implementations)
    pix_risk, _ = Hh.oracle_risk(oracle, sc, [f], guard_alpha=1e-4, guard_T=5e-4)
    assert not (differ & ~pix_risk[0]).any(), int((differ & ~pix_risk[0]).sum())
    rows = ~oracle.pixel_reach(Hh.oracle_camera(oracle, sc), f, differ, whole_list=True) if differ.any() else np.ones(P, bool)
    assert rows.mean() > 0.9, rows.mean()
    err = np.abs(f["color"] - color.numpy()) / np.maximum(np.abs(color.numpy()), 1e-2)
    assert err[:, ~differ].max() < 1e-4   # (fp32 accumulation over lists of hundreds of entries; 3.7e-5 measured)
    for k, ok in [("means3D", "dL_dmeans3D"), ("means2D", "dL_dmeans2D"), ("opacities", "dL_dopacity"),
                  ("shs", "dL_dshs"), ("scales", "dL_dscales"), ("rotations", "dL_drots")]:
        ref = g[k].reshape(b[ok].shape)[rows]
        got = b[ok][rows]
        mx, frac = Hh.rel_err(got, ref, Hh.grad_floor(ref))
        assert frac < 2e-2 and mx < 5e-2, (k, mx, frac)
        l2 = np.linalg.norm(got.astype(np.float64) - ref) / max(np.linalg.norm(ref), 1e-30)
        assert l2 < 2e-5, (k, l2)


def test_torch_rasterizer_gradcheck_fp64():
    """Finite differences on a tiny scene (5 Gaussians, 16x16).  Skip/termination decisions are piecewise
    constant, so the scene keeps every alpha well inside (1/255, 0.99) and T above 1e-4."""
    torch.manual_seed(0)
    W = H = 16
    cam = S.make_camera(W, H)
    view = torch_view(cam, torch.float64)
    P = 5
    fx = W / (2 * cam.tanfovx)
    z = torch.tensor([3.0, 4.0, 5.0, 6.0, 7.0], dtype=torch.float64)
    px = torch.tensor([5.0, 9.0, 7.0, 11.0, 4.0], dtype=torch.float64)
    py = torch.tensor([6.0, 8.0, 10.0, 5.0, 11.0], dtype=torch.float64)
    means = torch.stack([((2 * px + 1) / W - 1) * cam.tanfovx * z, ((2 * py + 1) / H - 1) * cam.tanfovy * z, z], 1)
    scales = (6.0 * z / fx)[:, None] * torch.tensor([[1.0, 0.8, 1.2]], dtype=torch.float64)
    q = torch.randn(P, 4, dtype=torch.float64)
    q = q / q.norm(dim=1, keepdim=True)
    opac = torch.full((P, 1), 0.35, dtype=torch.float64)
    shs = 0.3 * torch.randn(P, 4, 3, dtype=torch.float64)
    shs[:, 0] += 1.0  # keep colours positive: no clamp kinks
    dL = torch.randn(3, H, W, dtype=torch.float64)
    bg = torch.tensor([0.1, 0.2, 0.3], dtype=torch.float64)
    tab = S.sigmoid_crf_table(32).to(torch.float64)
    expo = torch.tensor(0.7, dtype=torch.float64)

    def fn(m, s, r, o, sh, e, t):
        hdr = TR.rasterize(view, m, o, 1, bg, shs=sh, scales=s, rotations=r)
        ldr = TR.tonemap(hdr, e, t, (-6.0, 3.0))
        return ((ldr + 0.1 * hdr) * dL).sum()

    ins = [t.clone().requires_grad_(True) for t in (means, scales, q, opac, shs, expo, tab)]
    assert torch.autograd.gradcheck(fn, ins, eps=1e-6, atol=1e-6, rtol=1e-4, nondet_tol=0.0)


def test_tonemap_matches_c_oracle_and_autograd(oracle):
    torch.manual_seed(1)
    Hd = torch.rand(3, 40, 30, dtype=torch.float64) * 8
    Hd[0, 0, 0] = 0.0       # below the table: flat
    Hd[1, 0, 0] = 1e4       # above the table: flat
    tab = S.sigmoid_crf_table(64).to(torch.float64).requires_grad_(True)
    expo = torch.tensor(0.5, dtype=torch.float64, requires_grad=True)
    Hd.requires_grad_(True)
    g = torch.randn(3, 40, 30, dtype=torch.float64)
    ldr = TR.tonemap(Hd, expo, tab, (-6.0, 3.0))
    (ldr * g).sum().backward()
    ldr_c = oracle.tonemap_fwd(Hd.detach().numpy(), 0.5, tab.detach().numpy(), -6.0, 3.0)
    assert np.allclose(ldr_c, ldr.detach().numpy(), rtol=2e-5, atol=1e-6)
    dh, dtab, dexp = oracle.tonemap_bwd(Hd.detach().numpy(), 0.5, tab.detach().numpy(), -6.0, 3.0, g.numpy())
    assert np.allclose(dh, Hd.grad.numpy(), rtol=2e-3, atol=1e-5)
    assert np.allclose(dtab, tab.grad.numpy(), rtol=1e-3, atol=2e-4)
    assert dexp == pytest.approx(float(expo.grad), rel=1e-3)
    assert dh[0, 0, 0] == 0 and dh[1, 0, 0] == 0


def test_n_identical_poses_equal_single_pose():
    sc = S.make_scene(200, 48, 32, 1, seed=9, hdr=True)
    dt = torch.float64
    v = torch_view(sc.camera, dt)
    kw = dict(shs=sc.shs.to(dt), scales=sc.scales.to(dt), rotations=sc.rotations.to(dt))
    args = (sc.means3D.to(dt), sc.opacities.to(dt), 1, sc.bg)
    for dom in ("ldr", "hdr"):
        l1, h1 = TR.rasterize_hdr([v], *args, sc.exposure.to(dt), sc.crf_table.to(dt), sc.crf_range, blur_domain=dom, **kw)
        l4, h4 = TR.rasterize_hdr([v] * 4, *args, sc.exposure.to(dt), sc.crf_table.to(dt), sc.crf_range, blur_domain=dom, **kw)
        assert torch.allclose(l1, l4, rtol=1e-12, atol=1e-14) and torch.allclose(h1, h4, rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize("antialiasing", [False, True])
def test_c_oracle_inverse_depth_and_antialiasing_match_fp64_autograd(oracle, antialiasing):
    """The C restatement of the two n3 extras (expected inverse depth as a fourth blended channel; the antialiasing
    opacity compensation with its covariance gradient) against float64 autograd of the pure-PyTorch rasterizer."""
    P, W, H, deg = 400, 72, 56, 2
    sc = S.make_scene(P, W, H, deg, seed=17)
    cam = sc.camera
    gen = torch.Generator().manual_seed(5)
    gD = torch.randn(H, W, generator=gen) * 4
    ocam = Hh.oracle_camera(oracle, sc)
    ocam.antialias = antialiasing
    kw = dict(shs=sc.shs.numpy(), scales=sc.scales.numpy(), rotations=sc.rotations.numpy())
    f = oracle.forward(ocam, sc.means3D.numpy(), sc.opacities.numpy(), **kw)
    b = oracle.backward(ocam, f, sc.dL_dimage.numpy(), sc.means3D.numpy(), dL_dinvdepth_img=gD.numpy(), **kw)

    dt = torch.float64
    leaves = {k: getattr(sc, k).to(dt).clone().requires_grad_(True) for k in ["means3D", "opacities", "shs", "scales", "rotations"]}
    color, st = TR.rasterize(torch_view(cam, dt), leaves["means3D"], leaves["opacities"], deg, sc.bg, shs=leaves["shs"],
                             scales=leaves["scales"], rotations=leaves["rotations"], return_state=True,
                             antialiasing=antialiasing)
    ((color * sc.dL_dimage.to(dt)).sum() + (st["invdepth"] * gD.to(dt)).sum()).backward()
    assert (st["n_contrib"].numpy() != f["n_contrib"]).sum() == 0
    assert Hh.rel_err(f["color"], color.detach().numpy(), 1e-2)[0] < 1e-5
    assert Hh.rel_err(f["invdepth"], st["invdepth"].detach().numpy(), 1e-3)[0] < 1e-5
    assert float(st["invdepth"].detach().max()) > 0.05
    if antialiasing:  # the compensation is exercised: it changes the opacity the render sees
        assert np.abs(f["conic_opacity"][:, 3] - sc.opacities.numpy().reshape(-1))[f["radii"] > 0].max() > 1e-3
    for k, ok in [("means3D", "dL_dmeans3D"), ("opacities", "dL_dopacity"), ("shs", "dL_dshs"), ("scales", "dL_dscales"),
                  ("rotations", "dL_drots")]:
        ref = leaves[k].grad.numpy().reshape(b[ok].shape)
        mx, frac = Hh.rel_err(b[ok], ref, Hh.grad_floor(ref))
        assert frac < 2e-2 and mx < 3e-2, (k, mx, frac)
        l2 = np.linalg.norm(b[ok].astype(np.float64) - ref) / max(np.linalg.norm(ref), 1e-30)
        assert l2 < 5e-5, (k, l2)


@pytest.mark.parametrize("act", ["exp", "softplus"])
def test_c_oracle_radiance_activations_match_fp64_autograd(oracle, act):
    """SURVEY.md 7.3 / 8a a1 `radiance_activation`: colour = e^s or ln(1 + e^s) of the SH sum instead of the published
    max(s + 0.5, 0).  The C restatement (forward and the hand-derived factor d colour / d s in the SH backward)
    against float64 autograd of the pure-PyTorch rasterizer."""
    P, W, H, deg = 400, 72, 56, 2
    sc = S.make_scene(P, W, H, deg, seed=21)
    f, b = Hh.run_oracle(oracle, sc, radiance_activation=act)
    dt = torch.float64
    leaves = {k: getattr(sc, k).to(dt).clone().requires_grad_(True) for k in ["means3D", "opacities", "shs", "scales", "rotations"]}
    color, st = TR.rasterize(torch_view(sc.camera, dt), leaves["means3D"], leaves["opacities"], deg, sc.bg, shs=leaves["shs"],
                             scales=leaves["scales"], rotations=leaves["rotations"], return_state=True,
                             radiance_activation=act)
    (color * sc.dL_dimage.to(dt)).sum().backward()
    vis = f["radii"] > 0
    want = st["pre"]["rgb"].detach().numpy()[vis]
    assert np.allclose(f["rgb"][vis], want, rtol=2e-6, atol=1e-7)
    assert f["rgb"][vis].min() > 0 and not f["clamped"].any()      # positive radiance, no clamp mask
    raw = np.log(np.expm1(want)) if act == "softplus" else np.log(want)
    assert raw.min() < -0.6                                        # relu_shift would have clamped some of these
    assert (st["n_contrib"].numpy() != f["n_contrib"]).sum() == 0
    assert Hh.rel_err(f["color"], color.detach().numpy(), 1e-2)[0] < 1e-5
    for k, ok in [("means3D", "dL_dmeans3D"), ("opacities", "dL_dopacity"), ("shs", "dL_dshs"), ("scales", "dL_dscales"),
                  ("rotations", "dL_drots")]:
        ref = leaves[k].grad.numpy().reshape(b[ok].shape)
        mx, frac = Hh.rel_err(b[ok], ref, Hh.grad_floor(ref))
        assert frac < 2e-2 and mx < 3e-2, (k, mx, frac)
        l2 = np.linalg.norm(b[ok].astype(np.float64) - ref) / max(np.linalg.norm(ref), 1e-30)
        assert l2 < 5e-5, (k, l2)


def test_softplus_gradient_of_dark_gaussians(oracle):
    """radiance_activation='softplus' far below zero (s = -17 .. -25: colour 4e-8 .. 1e-11): d colour / d s = sigmoid(s)
    must come out as ~colour, not as the 0 that 1 - expf(-colour) rounds to in fp32 -- dark Gaussians would stop
    receiving SH gradient.  The C oracle's SH-gradient rows of those Gaussians against float64 autograd, relative to
    the rows themselves (they are far below the tensor-wide floor of the other tests)."""
    P, W, H, deg = 300, 64, 48, 1
    sc = S.make_scene(P, W, H, deg, seed=33)
    dark = torch.arange(P) % 3 == 0
    sc.shs[dark, 0] = torch.linspace(-25.0, -17.0, int(dark.sum()))[:, None] / 0.28209479177387814
    sc.shs[dark, 1:] = 0.0
    f, b = Hh.run_oracle(oracle, sc, radiance_activation="softplus")
    dt = torch.float64
    leaves = {k: getattr(sc, k).to(dt).clone().requires_grad_(True) for k in ["means3D", "opacities", "shs", "scales", "rotations"]}
    color = TR.rasterize(torch_view(sc.camera, dt), leaves["means3D"], leaves["opacities"], deg, sc.bg, shs=leaves["shs"],
                         scales=leaves["scales"], rotations=leaves["rotations"], radiance_activation="softplus")
    (color * sc.dL_dimage.to(dt)).sum().backward()
    want = leaves["shs"].grad.numpy()[:, 0]
    got = b["dL_dshs"][:, 0]
    rows = dark.numpy() & (f["radii"] > 0) & (np.abs(want).max(axis=1) > 0)
    assert rows.sum() > 50
    assert 1e-13 < f["rgb"][rows].max() < 1e-7                      # the regime where 1 - expf(-col) is exactly 0
    rel = np.abs(got[rows] - want[rows]).max(axis=1) / np.abs(want[rows]).max(axis=1)
    assert rel.max() < 2e-3, float(rel.max())
