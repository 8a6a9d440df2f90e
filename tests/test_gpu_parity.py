"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI behind
GaussianRasterizer, against the CPU oracle and the committed golden fixtures.

Bars (BASELINE.json north_star): tile assignment / indexing bit-exact; pixel and gradient values within
1e-4 relative fp32.  Float details:
  * images: |got-ref| <= 1e-4 * max(|ref|, 1e-2) on every pixel, except that a pixel whose
    alpha / transmittance sits within an ulp of a threshold (1/255, 1e-4) may flip a contributor between
    the GPU's exp and glibc's expf -- such pixels are counted and bounded (<= 2e-5 of the pixels);
  * decisions: where HIP and oracle disagree on a pixel (contributor count, or transmittance off by one contributor:
    helpers.decision_masks) that pixel must lie inside the oracle's threshold guard band (oracle.threshold_risk); the
    golden fixtures are reject-sampled to have NO such pixel, so on them zero flips are demanded;
  * gradients: helpers.assert_grads_close with the STRICT bar -- tensors of >= 5000 Gaussians (helpers.STRICT_LARGE, round
    6): >= 99.8 % of elements within 1e-4 relative with a floor of 1e-3 * RMS, worst element <= 6e-3, relative L2 <= 5e-6;
    smaller ones: 99 %, 1e-2, 1e-5 -- on EVERY row except the Gaussians on the tile lists of pixels where a decision
    ACTUALLY differed (no differing pixel: every row strict) and, in HDR mode, the contributors of pixels within 4 ulp of a
    CRF knot; those rows get a finite bar (helpers.AT_RISK) and must stay as few as MEASURED for that very frame
    (`min_strict` = the frame's own share less half a per cent).  The masked pass (dL zeroed on the excused pixels on both
    sides) then holds every row to the bar and every element to 1e-4 |ref| + C 2^-24 sum w|term| (assert_grads_bounded).
    The strict bar is what profiles/r02_parity_table.json measures: HIP and the fp32 C oracle are equally far from
    float64 autograd (test_hip_is_as_close_to_fp64_truth_as_the_fp32_oracle asserts that triangle directly).
"""
import contextlib
import glob
import os

import numpy as np
import pytest
import torch

import helpers as Hh
from casualhdrsplat_amd import synthetic as S

pytestmark = pytest.mark.gpu

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
# libhdrsplat_test.so = the product sources + -DHS_TESTING: the only build that reads HS_FAULT_INJECT (csrc/Makefile)
_TEST_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "casualhdrsplat_amd", "libhdrsplat_test.so")


def u32(a):
    return np.asarray(a).astype(np.int64) & 0xFFFFFFFF


def check_structure(st, f, pose=0, P=None):
    """Bit-exact integer/structural parity of one pose against the oracle forward dict f."""
    P = P or f["radii"].shape[0]
    sl = slice(pose * P, (pose + 1) * P)
    for k in ("depths", "xy", "conic_opacity"):
        assert np.array_equal(Hh.bits(st[k][sl]), Hh.bits(f[k])), k
    assert np.array_equal(st["radii"][sl], f["radii"])
    assert np.array_equal(u32(st["tiles_touched"][sl]), u32(f["tiles_touched"]))
    assert np.abs(st["rgb"][sl] - f["rgb"]).max() <= 1e-6 * max(1.0, np.abs(f["rgb"]).max())


def assert_image_close(got, ref, what):
    """Every value within 1e-4 relative (floor 1e-2), except the few pixels where a contributor sits within an
    ulp of the alpha >= 1/255 or T >= 1e-4 threshold and flips between the GPU exp and glibc expf: those
    are counted (<= 2e-5 of the pixels) and bounded by the size of one minimal contribution.  (Images behind the CRF /
    the pose average, where no per-pixel decision record exists; check_image is the per-pose form.)"""
    e = np.abs(np.asarray(got, np.float64) - ref) / np.maximum(np.abs(ref), 1e-2)
    bad = e > 1e-4
    nbad = int(bad.reshape(-1, bad.shape[-2] * bad.shape[-1]).any(axis=0).sum()) if e.ndim == 3 else int(bad.sum())
    npix = e.shape[-1] * e.shape[-2]
    assert nbad <= max(2, int(2e-5 * npix)), (what, "pixels beyond 1e-4", nbad)
    assert e.max() <= 0.5, (what, float(e.max()))
    return nbad


def check_image(got, ref, masks, what, pose=0):
    """One pose's image against the oracle's, given helpers.decision_masks: a value off by more than 1e-4 relative
    (floor 1e-2) is allowed ONLY on a pixel where the two implementations demonstrably decided differently (which
    decision_masks has already confined to the guard band); such pixels are few (<= 2e-5 of the frame) and off by at most
    the size of one flipped contribution."""
    differs = masks["differs"][pose]
    e = np.abs(np.asarray(got, np.float64) - ref) / np.maximum(np.abs(ref), 1e-2)
    bad = (e > 1e-4).any(axis=0) if e.ndim == 3 else (e > 1e-4)
    assert not (bad & ~differs).any(), (what, "a pixel beyond 1e-4 where no decision differs", float(e.max()), int((bad & ~differs).sum()))
    assert int(differs.sum()) <= max(2, int(2e-5 * differs.size)), (what, "differing pixels", int(differs.sum()))
    assert e.max() <= 0.5, (what, float(e.max()))
    return int(differs.sum())


def free_camera_scene(P, W, H, deg, seed, cam_seed, **kw):
    """The synthetic cloud seen by a free 6-DoF camera (synthetic.random_camera: roll, pitch and yaw up to +-pi, a
    translation of a few units -- every entry of the view matrix populated; VERDICT r5 missing #4: until round 6 every GPU
    parity frame had zeros at V(0,1), V(1,0), V(1,2), V(2,1)).  cam_seed None: the default identity-rotation view."""
    return S.make_scene(P, W, H, deg, seed=seed, place_in=None if cam_seed is None else S.random_camera(W, H, cam_seed), **kw)


# (rows on the strict bar, measured on the MI355X in round 6 -- HIP path and oracle are both deterministic, so the share is
#  a property of the frame: 1.0 / 1.0 / 0.9824 / 0.9974 / 1.0 and, under the free cameras, 1.0 / 1.0 / 0.9964 / 0.9993; the
#  test asserts each frame's own share less half a per cent, not a blanket 0.85 / 0.95 -- VERDICT r5 next #5)
STRICT_SHARE = {(1000, 0, None): 0.995, (1000, 3, None): 0.995, (5000, 2, None): 0.977, (20000, 1, None): 0.992, (100000, 0, None): 0.995,
                (1000, 3, 0): 0.995, (5000, 2, 1): 0.995, (20000, 3, 2): 0.991, (100000, 0, 3): 0.994}


@pytest.mark.parametrize("P,W,H,deg,seed,cam_seed", [
    (1000, 128, 128, 0, 0, None), (1000, 128, 128, 3, 1, None), (5000, 200, 136, 2, 2, None), (20000, 500, 300, 1, 3, None),
    (100000, 800, 800, 0, 0, None),
    # free cameras (general SO(3) world-to-view rotation + translation), three sizes up to BASELINE c2
    (1000, 128, 128, 3, 1, 0), (5000, 200, 136, 2, 2, 1), (20000, 500, 300, 3, 3, 2), (100000, 800, 800, 0, 0, 3)])
def test_ldr_forward_backward_vs_oracle(oracle, P, W, H, deg, seed, cam_seed):
    sc = free_camera_scene(P, W, H, deg, seed, cam_seed)
    g = Hh.run_hip(sc)
    f, b = Hh.run_oracle(oracle, sc)
    st = g["state"]
    R = f["R"]
    assert st["num_rendered"] == R
    check_structure(st, f)
    assert np.array_equal(u32(st["offsets"]), u32(f["offsets"]))
    assert np.array_equal(st["keys_sorted"].view(np.uint64)[:R], f["keys_sorted"])
    assert np.array_equal(u32(st["point_list"][:R]), u32(f["point_list"]))
    assert np.array_equal(u32(st["ranges"]), u32(f["ranges"]))
    assert np.array_equal(g["radii"], f["radii"])
    m = Hh.decision_masks(oracle, sc, [f], st, what=f"P={P}")
    check_image(g["color"], f["color"], m, "color")
    check_image(st["final_T"][0], f["final_T"], m, "final_T")
    # strict bar on every row except the tile lists of pixels where a decision actually differed: >= 95 % of the rows at
    # BASELINE c2 size (the smaller frames: a single differing pixel already reaches a few per cent of 1000 Gaussians)
    Hh.assert_grads_close(g, b, what=f"P={P}", at_risk=m["rows"], min_strict=STRICT_SHARE[(P, deg, cam_seed)])
    # the pass that excuses nothing: dL zeroed on the pixels where a decision differed, both sides --
    # every row on the strict bar, every element inside 1e-4 |ref| + C_BOUND 2^-24 sqrt(n) sum|terms|
    g2, b2, _ = Hh.masked_backward_pass(oracle, sc, m, [f], hdr=False)
    # (the bar of the masked pass at c3 / c4 -- _masked_pass -- on every size; the fraction: 2e-3 where the full-size
    #  frames, with millions of elements, measure 3e-4 and these 1e-3: measured worst 9.7e-4 at 100 000 Gaussians)
    Hh.assert_grads_close(g2, b2, what=f"P={P} masked", frac_tol=2e-3, max_tol=1e-2, l2_tol=2e-6)
    Hh.assert_grads_bounded(g2, b2, what=f"P={P} masked")


@pytest.fixture(params=["default", "tickets", "scan_in_emission", "radix_tile_sort", "scan_in_emission+counting", "hier",
                        "hier+tickets", "lsd_depth_sort"])
def binning_mode(request):
    """The binning stage's alternate forms inside the driver's suite: chain positions of the radix
    passes from start-order tickets (hs_sort_tickets(1): what a process on a shared GPU runs) and the pair emission
    computing its block offsets itself by decoupled look-back (HS_SCAN_IN_EMISSION=1: what frames of >= 2^21 instances
    run), each forced onto frames that would take the default form.  Small frames (<= 4096 tiles, a few hundred thousand
    instances: most of this suite) sort their pairs by counting instead of radix passes: "tickets" and "scan_in_emission"
    keep the radix passes on them as well (HS_TILE_SORT=radix), "radix_tile_sort" only does that, and
    "scan_in_emission+counting" runs the counting sort behind the chained-scan emission.  "hier" (round 6): the hierarchical
    tile sort -- one element per (8 x 8-tile super-tile, instance), one stable radix pass, expansion by counting.
    Frames below 2^21 instances sort their instances by depth by counting since round 6 (one counting pass + range sorts in
    LDS): the "tickets" forms and "lsd_depth_sort" keep the look-back passes of the depth sort on them."""
    from casualhdrsplat_amd import _lib as L
    lib = L.load()
    was, env = lib.hs_sort_tickets(-1), {k: os.environ.get(k) for k in ("HS_SCAN_IN_EMISSION", "HS_TILE_SORT", "HS_DEPTH_SORT")}
    if request.param in ("tickets", "hier+tickets", "lsd_depth_sort"):
        os.environ["HS_DEPTH_SORT"] = "lsd"
    if request.param in ("tickets", "hier+tickets"):
        lib.hs_sort_tickets(1)
    if request.param.startswith("hier"):   # round 6: coarse stable pass over (super-tile, instance) elements + expansion
        os.environ["HS_TILE_SORT"] = "hier"
    if request.param.startswith("scan_in_emission"):
        os.environ["HS_SCAN_IN_EMISSION"] = "1"
    if request.param in ("tickets", "scan_in_emission", "radix_tile_sort"):
        os.environ["HS_TILE_SORT"] = "radix"
    yield request.param
    lib.hs_sort_tickets(was)
    for k, v in env.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


@pytest.mark.parametrize("case", ["ldr_20000", "tiles_14400", "c1_hdr_fixture"])
def test_binning_modes_vs_oracle(oracle, binning_mode, case):
    """Three frames of the suite -- 20 000 Gaussians at 500 x 300, the 14 400-tile frame, a c1-scale HDR golden fixture
    with 8 poses -- through every form of the binning stage: the same bit-exact structure, images and gradients."""
    if case == "ldr_20000":
        test_ldr_forward_backward_vs_oracle(oracle, 20000, 500, 300, 1, 3, None)
    elif case == "tiles_14400":
        test_frame_of_14400_tiles_vs_oracle(oracle)
    else:
        path = [p for p in GOLDEN if os.path.basename(p) == "c1_hdr_deg3_n8_ldrblur.npz"]
        assert path, "fixture c1_hdr_deg3_n8_ldrblur.npz is missing"
        test_against_golden_fixtures(path[0])


def _act(z):
    return str(z["radiance_activation"]) if "radiance_activation" in z.files else "relu_shift"


def _golden_antialias_invdepth(z, sc):
    """HIP path with settings.antialiasing + return_invdepth against the committed oracle output."""
    from casualhdrsplat_amd import GaussianRasterizationSettings, GaussianRasterizer, inspect_state
    dev = "cuda"
    rs, _, _ = Hh.settings_from_scene(sc, dev)
    rs = rs._replace(antialiasing=True)
    names = ("means3D", "opacities", "shs", "scales", "rotations")
    leaf = {k: getattr(sc, k).to(dev).requires_grad_(True) for k in names}
    m2 = torch.zeros(sc.means3D.shape[0], 3, device=dev, requires_grad=True)
    out = GaussianRasterizer(rs, return_invdepth=True)(leaf["means3D"], m2, leaf["opacities"], shs=leaf["shs"],
                                                       scales=leaf["scales"], rotations=leaf["rotations"])
    st = inspect_state(out[0])
    ((out[0] * sc.dL_dimage.to(dev)).sum() + (out[2] * torch.from_numpy(z["dL_dinvdepth"]).to(dev)).sum()).backward()
    R = st["num_rendered"]
    assert np.array_equal(Hh.bits(st["conic_opacity"].cpu().numpy()), Hh.bits(z["o_conic_opacity"]))  # incl. the AA opacity
    assert np.array_equal(u32(st["point_list"][:R].cpu().numpy()), u32(z["o_point_list"]))
    assert np.array_equal(u32(st["ranges"].cpu().numpy()), u32(z["o_ranges"]))
    assert int((u32(st["n_contrib"][0].cpu().numpy()) != u32(z["o_n_contrib"])).sum()) == 0
    assert Hh.rel_err(out[0].detach().cpu().numpy(), z["o_color"], 1e-2)[0] <= 1e-4
    assert Hh.rel_err(out[2].detach().cpu().numpy(), z["o_invdepth"], 1e-3)[0] <= 1e-4
    g = {"d_" + k: v.grad.cpu().numpy() for k, v in leaf.items()}
    g["d_means2D"] = m2.grad.cpu().numpy()
    Hh.assert_grads_close(g, {k: z["o_" + k] for _, k in Hh.GRAD_KEYS}, what="antialias+invdepth golden")


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_against_golden_fixtures(path):
    from test_golden_cpu import scene_from_golden
    z = np.load(path)
    sc, cams, hdr, dom = scene_from_golden(z)
    if "antialias" in z.files:
        return _golden_antialias_invdepth(z, sc)
    g = Hh.run_hip(sc, cameras=cams if len(cams) > 1 else None, hdr=hdr, blur_domain=dom, radiance_activation=_act(z))
    st = g["state"]
    assert np.array_equal(u32(st["point_list"][:st["num_rendered"]]), u32(z["o_point_list"]))
    assert np.array_equal(u32(st["ranges"]), u32(z["o_ranges"]))
    if bool(int(z["guard_banded"])):
        assert Hh.rel_err(g["color"], z["o_color"], 1e-2)[0] <= 1e-4
    else:
        assert_image_close(g["color"], z["o_color"], os.path.basename(path))
    ref = {k: z["o_" + k] for _, k in Hh.GRAD_KEYS}
    # guard-banded fixture: every fp32 implementation takes the stored decisions -- zero flips, on every pose
    guard_banded = bool(int(z["guard_banded"]))
    nc_ref = u32(z["o_n_contrib"]).reshape(u32(st["n_contrib"]).shape)
    if guard_banded:
        assert int((u32(st["n_contrib"]) != nc_ref).sum()) == 0
    if hdr:
        if guard_banded:
            assert Hh.rel_err(g["hdr"], z["o_hdr"], 1e-2)[0] <= 1e-4
        else:
            assert_image_close(g["hdr"], z["o_hdr"], os.path.basename(path))
        tab = z["o_dL_dcrf_table"]
        assert Hh.rel_err(g["d_crf_table"], tab, 1e3 * Hh.grad_floor(tab))[0] <= 2e-4
        assert float(g["d_exposure"]) == pytest.approx(float(z["o_dL_dexposure"]), rel=2e-4, abs=1e-3)
    else:
        assert np.array_equal(st["keys_sorted"].view(np.uint64)[:st["num_rendered"]], z["o_keys_sorted"])
        assert np.array_equal(Hh.bits(st["depths"]), Hh.bits(z["o_depths"]))
    at_risk = None
    if hdr:
        # the alpha / T decisions of a fixture are guard-banded; the CRF interval of a pixel is guard-banded too in the
        # c1_* fixtures (no pixel within 8 ulp of a knot: every row strict), in the older ones the contributors of the
        # few pixels within 4 ulp of a knot go to the at-risk bar
        from oracle import c_oracle as O
        r = Hh.run_oracle_hdr(O, sc, cams, dom, radiance_activation=_act(z))
        got_imgs, ref_imgs = ([g["hdr"]], [r["hdr"]]) if (dom == "hdr" or len(cams) == 1) else \
            (list(st["pose_hdr"][:len(cams)]), [f["color"] for f in r["fwd"]])
        m = Hh.decision_masks(O, sc, r["fwd"], st, cams, crf_got=got_imgs, crf_ref=ref_imgs, what=os.path.basename(path))
        # (the two 8-pose c1 frames could not be reject-sampled: decisions may differ inside the guard band, which
        # decision_masks has just asserted, and only the rows on those pixels' tile lists leave the strict bar)
        assert m["n_differ"] == 0 or not guard_banded
        at_risk = m["rows"]
        if "crf_knot_guarded" in z.files and m["n_differ"] == 0:
            assert not at_risk.any(), int(at_risk.sum())   # c1 fixtures: EVERY row on the strict bar
    Hh.assert_grads_close(g, ref, what=os.path.basename(path), at_risk=at_risk, min_strict=0.96)   # (measured: 0.9675 for hdr_deg1_n4_hdrblur, 1.0 for every other fixture)


def test_hip_is_as_close_to_fp64_truth_as_the_fp32_oracle(oracle):
    """The testable form of "as accurate as the reference precision" (VERDICT r1 #2b): on guard-banded scenes (all
    three implementations take identical decisions) the error of the HIP gradients against float64 autograd of the
    pure-PyTorch rasterizer is no larger than that of the fp32 C oracle against the same truth -- in relative L2 within
    10 %, in the 99.9th percentile and the worst element within a factor of two (order statistics of ~1e3-1e4 elements
    fluctuate by that much between two equally accurate fp32 evaluation orders; profiles/r02_parity_table.json)."""
    from test_oracle_cross import torch_run
    for P, W, H, deg, seed0 in ((1000, 128, 128, 3, 0), (600, 96, 96, 1, 0)):
        sc, seed = Hh.guarded_scene(oracle, P, W, H, deg, seed=seed0)
        f, b = Hh.run_oracle(oracle, sc)
        g = Hh.run_hip(sc)
        _, st64, g64 = torch_run(sc, torch.float64)
        assert int((st64["n_contrib"].numpy() != f["n_contrib"]).sum()) == 0
        assert int((u32(g["state"]["n_contrib"][0]) != u32(f["n_contrib"])).sum()) == 0
        for k, ok in Hh.GRAD_KEYS:
            truth = g64[k].reshape(b[ok].shape).astype(np.float64)
            floor = Hh.grad_floor(truth)

            def err(x):
                e = np.abs(np.asarray(x, np.float64).reshape(truth.shape) - truth) / np.maximum(np.abs(truth), floor)
                l2 = np.linalg.norm(np.asarray(x, np.float64).reshape(truth.shape) - truth) / np.linalg.norm(truth)
                return float(e.max()), float(np.percentile(e, 99.9)), float(l2)

            (mh, ph, lh), (mo, po, lo) = err(g["d_" + k]), err(b[ok])
            assert lh <= 1.1 * lo + 1e-8 and ph <= 2.0 * po + 1e-6 and mh <= 2.0 * mo + 1e-5, (P, seed, k, (mh, ph, lh), (mo, po, lo))
            assert lh <= 1e-5 and ph <= 2e-3, (k, lh, ph)


@pytest.mark.parametrize("act", ["exp", "softplus"])
def test_radiance_activations_vs_oracle(oracle, act):
    """colour = e^s / ln(1 + e^s) of the SH sum.  HIP against the C oracle: a single view, and
    the N-pose HDR image formation (tone-map, blur average, CRF / exposure gradients) on top of it."""
    sc = S.make_scene(6000, 224, 144, 2, seed=12)
    g = Hh.run_hip(sc, radiance_activation=act)
    f, b = Hh.run_oracle(oracle, sc, radiance_activation=act)
    st = g["state"]
    check_structure(st, f)
    assert np.array_equal(u32(st["point_list"][:f["R"]]), u32(f["point_list"]))
    assert f["rgb"][f["radii"] > 0].min() > 0 and not st["clamped"].any()
    m = Hh.decision_masks(oracle, sc, [f], st, what=act)
    check_image(g["color"], f["color"], m, act)
    Hh.assert_grads_close(g, b, what=act, at_risk=m["rows"], min_strict=0.99)          # (measured 1.0)
    # N poses + exposure / CRF epilogue
    sch = S.make_scene(2500, 160, 96, 1, seed=13, hdr=True)
    sch.shs[:, 0] *= 0.25   # keep e^s inside the CRF table's range for most Gaussians
    cams = S.blur_poses(160, 96, 3, step=0.02)
    g = Hh.run_hip(sch, cameras=cams, hdr=True, radiance_activation=act)
    r = Hh.run_oracle_hdr(oracle, sch, cams, "ldr", radiance_activation=act)
    assert_image_close(g["color"], r["ldr"], act)
    assert_image_close(g["hdr"], r["hdr"], act)
    m = Hh.decision_masks(oracle, sch, r["fwd"], g["state"], cams, crf_got=list(g["state"]["pose_hdr"][:3]),
                          crf_ref=[f["color"] for f in r["fwd"]], what=act + " hdr")
    Hh.assert_grads_close(g, r, what=act + " hdr", at_risk=m["rows"], min_strict={"exp": 0.99, "softplus": 0.86}[act])   # (measured 1.0 / 0.8688)
    tab = r["dL_dcrf_table"]
    assert Hh.rel_err(g["d_crf_table"], tab, 1e3 * Hh.grad_floor(tab))[0] <= 2e-4
    assert float(g["d_exposure"]) == pytest.approx(r["dL_dexposure"], rel=1e-3)


def test_precomputed_colors_and_covariance(oracle):
    sc = S.make_scene(3000, 160, 112, 0, seed=7)
    f0, _ = Hh.run_oracle(oracle, sc, backward=False)
    cols = torch.rand(3000, 3, generator=torch.Generator().manual_seed(1))
    cov = torch.from_numpy(f0["cov3D"].copy())
    g = Hh.run_hip(sc, use_colors_precomp=cols, use_cov_precomp=cov)
    f, b = Hh.run_oracle(oracle, sc, use_colors_precomp=cols, use_cov_precomp=cov)
    st = g["state"]
    check_structure(st, f)
    assert np.array_equal(u32(st["point_list"][:f["R"]]), u32(f["point_list"]))
    m = Hh.decision_masks(oracle, sc, [f], st, what="precomp")
    check_image(g["color"], f["color"], m, "color")
    Hh.assert_grads_close(g, b, keys=[("means3D", "dL_dmeans3D"), ("opacities", "dL_dopacity"),
                                      ("colors_precomp", "dL_dcolors_precomp"), ("cov3D_precomp", "dL_dcov3D")],
                          at_risk=m["rows"], min_strict=0.99)     # (measured 1.0)
    assert g["d_shs" if "d_shs" in g else "d_colors_precomp"] is not None


def test_hdr_with_direct_radiance_gradient(oracle):
    """Loss on both outputs: LDR (through the CRF) and the linear radiance image."""
    sc = S.make_scene(4000, 192, 128, 3, seed=4, hdr=True)
    gh = torch.randn(3, 128, 192, generator=torch.Generator().manual_seed(5))
    g = Hh.run_hip(sc, hdr=True, grad_hdr=gh)
    r = Hh.run_oracle_hdr(oracle, sc, dL_hdr=gh.numpy())
    assert_image_close(g["color"], r["ldr"], "ldr")
    assert_image_close(g["hdr"], r["hdr"], "hdr")
    m = Hh.decision_masks(oracle, sc, r["fwd"], g["state"], crf_got=[g["hdr"]], crf_ref=[r["hdr"]], what="hdr+radiance")
    check_image(g["hdr"], r["hdr"], m, "hdr")
    Hh.assert_grads_close(g, r, at_risk=m["rows"], min_strict=0.97)       # (measured 0.9768)
    assert float(g["d_exposure"]) == pytest.approx(r["dL_dexposure"], rel=1e-3)


@pytest.mark.parametrize("free", [False, True], ids=["x_shift_poses", "free_rotating_poses"])
@pytest.mark.parametrize("dom", ["ldr", "hdr"])
def test_motion_blur_n_poses(oracle, dom, free):
    """free: the eight poses of the exposure differ in ROTATION (roll, pitch and yaw of 0.25 degrees per pose, about a
    free 6-DoF base camera) as well as in translation -- synthetic.perturbed_poses."""
    if free:
        base = S.random_camera(160, 96, 5)
        sc = S.make_scene(3000, 160, 96, 2, seed=6, hdr=True, place_in=base)
        cams = S.perturbed_poses(base, 8, seed=1, rot_step_deg=0.25, step=0.02)
    else:
        sc = S.make_scene(3000, 160, 96, 2, seed=6, hdr=True)
        cams = S.blur_poses(160, 96, 8, step=0.02)
    g = Hh.run_hip(sc, cameras=cams, hdr=True, blur_domain=dom)
    r = Hh.run_oracle_hdr(oracle, sc, cams, dom)
    st = g["state"]
    for k, f in enumerate(r["fwd"]):
        check_structure(st, f, pose=k, P=3000)
    assert st["num_rendered"] == sum(f["R"] for f in r["fwd"])
    assert np.array_equal(g["radii"], np.max(np.stack([f["radii"] for f in r["fwd"]]), axis=0))
    assert_image_close(g["color"], r["ldr"], "ldr")
    assert_image_close(g["hdr"], r["hdr"], "hdr")
    got_imgs, ref_imgs = ([g["hdr"]], [r["hdr"]]) if dom == "hdr" else (list(st["pose_hdr"][:8]), [f["color"] for f in r["fwd"]])
    m = Hh.decision_masks(oracle, sc, r["fwd"], st, cams, crf_got=got_imgs, crf_ref=ref_imgs, what=dom)
    for k, f in enumerate(r["fwd"]):
        check_image(st["pose_hdr"][k], f["color"], m, f"pose {k}", pose=k)
    # (worst element: 2e-2 as at c4 -- a sum over eight poses' worth of pixel terms; measured 1.16e-2 on ONE element of the
    #  free-pose frame, a 3-pixel Gaussian deep inside long lists whose terms cancel to 1e-4 of their magnitudes: 0.08 of the
    #  per-element bound below)
    # (rows on the strict bar, measured: x-shift poses 0.9177 / 1.0 (ldr / hdr domain), free rotating poses 0.8667 / 0.9473)
    share = {("ldr", False): 0.91, ("hdr", False): 0.99, ("ldr", True): 0.86, ("hdr", True): 0.94}[(dom, free)]
    Hh.assert_grads_close(g, r, at_risk=m["rows"], min_strict=share, max_tol=2e-2)
    # ... and the pass that excuses nothing: dL zeroed on the differing / knot pixels on both sides, every row strict, every
    # element inside 1e-4 |ref| + C_BOUND 2^-24 sum w|term|
    g2, r2, _ = Hh.masked_backward_pass(oracle, sc, m, r["fwd"], cameras=cams, hdr=True, blur_domain=dom)
    Hh.assert_grads_close(g2, r2, what=f"{dom} masked")
    Hh.assert_grads_bounded(g2, r2, what=f"{dom} masked")
    tab = r["dL_dcrf_table"]
    if m["n_differ"]:
        # a decision differed on some pixel (inside the guard band, asserted above): one flipped contribution moves that
        # pixel's log-exposure and with it |dL| of weight between two table entries -- 3.5e-2 of a knot's sum on a frame of
        # 15 000 pixels.  The table gradient is then checked GIVEN the decisions, as at full size (test_c3_..., test_c4_...);
        # in the radiance domain the CRF sees the MEAN image: one image, differing where any pose's decision did
        if dom == "ldr":
            tab, _ = Hh.crf_grads_given_decisions(oracle, sc, m, ref_imgs, got_imgs)
        else:
            tab, _ = Hh.crf_grads_given_decisions(oracle, sc, {"differs": m["differs"].any(axis=0, keepdims=True)},
                                                  [r["hdr"]], [g["hdr"]])
    assert Hh.rel_err(g["d_crf_table"], tab, 1e3 * Hh.grad_floor(tab))[0] <= 2e-4


def test_n_identical_poses_equal_single_pose():
    sc = S.make_scene(5000, 256, 144, 1, seed=8, hdr=True)
    one = Hh.run_hip(sc, hdr=True)
    four = Hh.run_hip(sc, cameras=[sc.camera] * 4, hdr=True)
    assert np.allclose(one["color"], four["color"], rtol=1e-6, atol=1e-7)
    for k in ("means3D", "shs", "scales", "rotations", "opacities"):
        a, b = one["d_" + k], four["d_" + k]
        assert np.allclose(a, b, rtol=1e-4, atol=1e-5 * np.abs(a).max()), k


def test_non_hdr_multi_pose_average(oracle):
    sc = S.make_scene(2000, 128, 80, 0, seed=3)
    cams = S.blur_poses(128, 80, 3, step=0.03)
    g = Hh.run_hip(sc, cameras=cams)
    fs = [Hh.run_oracle(oracle, sc, cam=c, backward=False)[0] for c in cams]
    want = np.mean(np.stack([f["color"] for f in fs]), axis=0)
    assert_image_close(g["color"], want, "mean of poses")
    acc = None
    for c, f in zip(cams, fs):
        b = oracle.backward(Hh.oracle_camera(oracle, sc, c), f, sc.dL_dimage.numpy() / 3, sc.means3D.numpy(),
                            shs=sc.shs.numpy(), scales=sc.scales.numpy(), rotations=sc.rotations.numpy())
        acc = b if acc is None else {k: (acc[k] + b[k] if isinstance(b[k], np.ndarray) and b[k].dtype == np.float32 else acc[k]) for k in acc}
    Hh.assert_grads_close(g, acc)


def test_edge_cases(oracle):
    from casualhdrsplat_amd import GaussianRasterizer
    dev = "cuda"
    # (1) empty cloud: background only
    sc = S.make_scene(10, 70, 50, 0, seed=0)
    sc.bg = torch.tensor([0.2, 0.4, 0.6])
    rs, _, _ = Hh.settings_from_scene(sc, dev)
    r = GaussianRasterizer(rs)
    e = torch.zeros(0, 3, device=dev)
    color, radii = r(e, e, torch.zeros(0, 1, device=dev), shs=torch.zeros(0, 1, 3, device=dev), scales=e, rotations=torch.zeros(0, 4, device=dev))
    assert radii.numel() == 0 and torch.allclose(color, sc.bg.to(dev)[:, None, None].expand(3, 50, 70))
    # (2) everything behind the near plane
    sc2 = S.make_scene(500, 70, 50, 1, seed=1)
    sc2.means3D[:, 2] = -sc2.means3D[:, 2]
    g = Hh.run_hip(sc2)
    assert g["state"]["num_rendered"] == 0 and np.all(g["radii"] == 0) and np.all(g["color"] == 0)
    assert all(np.all(g["d_" + k] == 0) for k in ("means3D", "shs", "scales", "rotations", "opacities"))
    # (3) one Gaussian, ragged image (not a multiple of 16), bright background
    sc3 = S.make_scene(1, 37, 23, 0, seed=2)
    sc3.bg = torch.tensor([1.0, 0.5, 0.25])
    g = Hh.run_hip(sc3)
    f, b = Hh.run_oracle(oracle, sc3)
    assert_image_close(g["color"], f["color"], "single")
    Hh.assert_grads_close(g, b, frac_tol=0.34, max_tol=5e-4)  # 3-4 elements per tensor: bound every one at 5e-4
    # (4) a huge Gaussian covering every tile + many tiny ones (ragged list lengths)
    sc4 = S.make_scene(800, 208, 120, 0, seed=3)
    sc4.scales[0] = 3.0
    sc4.means3D[0] = torch.tensor([0.0, 0.0, 4.0])
    g = Hh.run_hip(sc4)
    f, b = Hh.run_oracle(oracle, sc4)
    assert g["state"]["num_rendered"] == f["R"] and f["tiles_touched"][0] == 13 * 8
    assert np.array_equal(u32(g["state"]["point_list"][:f["R"]]), u32(f["point_list"]))
    m = Hh.decision_masks(oracle, sc4, [f], g["state"], what="huge")
    check_image(g["color"], f["color"], m, "huge")
    Hh.assert_grads_close(g, b, at_risk=m["rows"], min_strict=0.99)   # (measured 1.0)


def test_mark_visible(oracle):
    from casualhdrsplat_amd import GaussianRasterizer
    sc = S.make_scene(5000, 64, 64, 0, seed=0)
    sc.means3D[::3, 2] -= 3.0
    rs, _, _ = Hh.settings_from_scene(sc, "cuda")
    vis = GaussianRasterizer(rs).markVisible(sc.means3D.cuda()).cpu().numpy()
    assert np.array_equal(vis, oracle.mark_visible(Hh.oracle_camera(oracle, sc), sc.means3D.numpy()))
    assert 0 < vis.sum() < 5000


def test_fixed_capacity_mode_and_overflow():
    sc = S.make_scene(20000, 320, 200, 1, seed=5)
    ref = Hh.run_hip(sc)
    R = ref["state"]["num_rendered"]
    ok = Hh.run_hip(sc, capacity=R + 1000)
    assert ok["state"]["num_rendered"] == R
    assert np.array_equal(ok["color"], ref["color"])
    for k in ("means3D", "shs", "scales"):
        assert np.array_equal(ok["d_" + k], ref["d_" + k]), k  # deterministic, capacity-independent
    with pytest.raises(RuntimeError, match="binning capacity"):
        Hh.run_hip(sc, capacity=R // 2)


def _rasterizer_and_inputs(sc, capacity, requires_grad, **kw):
    from casualhdrsplat_amd import GaussianRasterizer
    rs, _, _ = Hh.settings_from_scene(sc, "cuda")
    rast = GaussianRasterizer(rs, capacity=capacity, **kw)
    leaf = {k: t.clone().cuda().requires_grad_(requires_grad) for k, t in
            dict(means3D=sc.means3D, means2D=torch.zeros_like(sc.means3D), opacities=sc.opacities, shs=sc.shs,
                 scales=sc.scales, rotations=sc.rotations).items()}
    call = lambda: rast(leaf["means3D"], leaf["means2D"], leaf["opacities"], shs=leaf["shs"], scales=leaf["scales"],
                        rotations=leaf["rotations"])
    return rast, leaf, call


def test_overflow_in_a_no_grad_forward_grows_capacity_and_replays():
    """Sync-free mode, forward that no backward will follow: an overflowing frame is never handed
    out empty -- the rasterizer grows its capacity to 1.5 x the device-reported pair count and replays; the result
    equals the synchronous path bit for bit."""
    from casualhdrsplat_amd import inspect_state
    sc = S.make_scene(20000, 320, 200, 1, seed=5)
    ref = Hh.run_hip(sc)
    R = ref["state"]["num_rendered"]
    for mode in ("no_grad", "no_leaf_requires_grad"):
        rast, leaf, call = _rasterizer_and_inputs(sc, R // 3, requires_grad=False, keep_state=True)
        if mode == "no_grad":
            with torch.no_grad():
                out = call()
        else:
            out = call()
        assert rast.overflow_replays == 1 and rast.capacity >= R, (rast.overflow_replays, rast.capacity, R)
        assert rast.last_num_rendered == R
        assert np.array_equal(out[0].cpu().numpy(), ref["color"]), mode
        assert np.array_equal(out[1].cpu().numpy(), ref["radii"])
        st = inspect_state(rast)  # no autograd graph: the state comes from keep_state=True
        assert st["num_rendered"] == R
        assert np.array_equal(st["point_list"].cpu().numpy()[:R], ref["state"]["point_list"][:R])
        out2 = call()  # the grown capacity stays: no second replay
        assert rast.overflow_replays == 1 and np.array_equal(out2[0].cpu().numpy(), ref["color"])


@pytest.mark.parametrize("form", ["count", "hier", "radix"])
def test_overflow_in_a_training_step_raises_then_the_repeated_step_succeeds(form):
    """A training forward does not wait for its counters; its backward finds the overflow, raises (the loss was
    computed from an empty frame) and leaves the rasterizer with a capacity that fits, so repeating the step works
    and matches the synchronous path.  Through each of the three tile sorts (every one has its own verdict path: the pair
    emission's, the hierarchical emission's, the counting scatter's)."""
    with tile_sort(form):
        _overflow_in_a_training_step()


def _overflow_in_a_training_step():
    from casualhdrsplat_amd import BinningOverflow
    sc = S.make_scene(20000, 320, 200, 1, seed=5)
    ref = Hh.run_hip(sc)
    R = ref["state"]["num_rendered"]
    rast, leaf, call = _rasterizer_and_inputs(sc, R // 2, requires_grad=True)
    out = call()
    with pytest.raises(BinningOverflow, match="binning capacity"):
        rast.check_overflow()          # explicit early check: one event wait
    assert rast.capacity >= R
    rast.capacity = R // 2             # undo, to exercise the lazy path through the backward
    out = call()
    assert float(out[0].detach().abs().max()) == 0.0   # the device rendered the frame empty (background 0), nothing out of bounds
    with pytest.raises(BinningOverflow, match="empty frame"):
        (out[0] * sc.dL_dimage.cuda()).sum().backward()
    out = call()                       # capacity was grown by the failed backward's verdict
    assert rast.capacity >= R and rast.last_num_rendered == R
    (out[0] * sc.dL_dimage.cuda()).sum().backward()
    assert np.array_equal(out[0].detach().cpu().numpy(), ref["color"])
    for k in ("means3D", "shs", "scales"):
        assert np.array_equal(leaf[k].grad.cpu().numpy(), ref["d_" + k]), k


def test_run_to_run_bitwise_determinism():
    sc = S.make_scene(50000, 640, 360, 3, seed=9, hdr=True)
    a = Hh.run_hip(sc, hdr=True)
    for rep in range(3):
        b = Hh.run_hip(sc, hdr=True)
        assert np.array_equal(a["color"], b["color"]) and np.array_equal(a["hdr"], b["hdr"])
        # every output, the CRF-table and exposure gradients included (fixed-point LDS accumulation, fixed-order sums)
        for k in ("means3D", "means2D", "opacities", "shs", "scales", "rotations", "crf_table", "exposure"):
            assert np.array_equal(a["d_" + k], b["d_" + k]), (k, rep)
    assert np.abs(a["d_crf_table"]).max() > 0 and float(np.abs(a["d_exposure"]).max()) > 0


def test_backward_replays_give_identical_gradients():
    """The render backward hands its last tiles out through a queue whose counter resets itself, and the tile sort's
    scratch is cleared inside the forward: re-enqueueing the backward (or single stages of it) on the state of ONE
    forward -- what bench.py's stage timing and a retained-graph second backward do -- must give the same bits every
    time, whichever workgroups took the queued tiles."""
    from casualhdrsplat_amd import GaussianRasterizer, _lib as L
    from casualhdrsplat_amd.rasterizer import replay_backward, replay_forward
    dev = "cuda"
    sc = S.make_scene(60000, 720, 400, 2, seed=17, hdr=True)     # 1125 tiles: 90 of them go through the queue
    rs, expo, crf = Hh.settings_from_scene(sc, dev, hdr=True, requires_grad=True)
    leaves = [getattr(sc, k).to(dev).requires_grad_(True) for k in ("means3D", "opacities", "shs", "scales", "rotations")]
    out = GaussianRasterizer(rs)(leaves[0], torch.zeros(60000, 3, device=dev, requires_grad=True), leaves[1],
                                 shs=leaves[2], scales=leaves[3], rotations=leaves[4])
    dL = sc.dL_dimage.to(dev)
    ref = None
    for rep in range(6):
        if rep == 3:   # the forward's own stages replayed in between (binning scratch, activity bytes rewritten)
            replay_forward(out[0], L.HS_STAGE_BIN | L.HS_STAGE_RENDER)
        g = replay_backward(out[0], dL) if rep % 2 == 0 else (replay_backward(out[0], dL, L.HS_BWD_RENDER | L.HS_BWD_CRF),
                                                             replay_backward(out[0], dL))[1]
        # every gradient slice (the flat buffer itself has uninitialised 16-byte pads between them)
        got = {k: v.detach().cpu().numpy().copy() for k, v in g.items()
               if isinstance(v, torch.Tensor) and not k.startswith("_") and k != "view_colors"}
        if ref is None:
            ref = got
            assert len(ref) >= 8 and np.abs(ref["means3D"]).max() > 0 and np.abs(ref["crf_table"]).max() > 0
        for k in ref:
            assert np.array_equal(Hh.bits(got[k]), Hh.bits(ref[k])), (rep, k)


def test_frame_of_14400_tiles_vs_oracle(oracle):
    """2560 x 1440 = 14400 tiles: more than 1024 tiles per XCD, so the tile-ordering kernel walks two tiles per thread,
    and still a few-round launch, so the backward takes its tiles through the ordered list and the queue.  Against the
    oracle: structure bit-exact, image and gradients within the contract."""
    P, W, H, deg = 40000, 2560, 1440, 1
    sc = S.make_scene(P, W, H, deg, seed=23)
    f, b = Hh.run_oracle(oracle, sc)
    g = Hh.run_hip(sc, capacity=f["R"] + 1000)
    st = g["state"]
    assert st["num_rendered"] == f["R"]
    check_structure(st, f)
    assert np.array_equal(u32(st["point_list"][:f["R"]]), u32(f["point_list"])) and np.array_equal(u32(st["ranges"]), u32(f["ranges"]))
    m = Hh.decision_masks(oracle, sc, [f], st, what="14400 tiles")
    check_image(g["color"], f["color"], m, "14400 tiles")
    Hh.assert_grads_close(g, b, what="14400 tiles", at_risk=m["rows"], min_strict=0.995)   # (measured 0.9995)


def skewed_scene(P=30000, W=640, H=384, deg=1, seed=31):
    """Half of the cloud inside ONE 8 x 8-tile super-tile (the 128 x 128-pixel block at tiles (8..15, 8..15)): the skewed
    load of the hierarchical tile sort -- one super-tile holds hundreds of chunks, its neighbours a handful."""
    sc = S.make_scene(P, W, H, deg, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    half = torch.rand(P, generator=g) < 0.5
    fx = W / (2 * sc.camera.tanfovx)
    px = 128.0 + 128.0 * torch.rand(P, generator=g, dtype=torch.float64)
    py = 128.0 + 128.0 * torch.rand(P, generator=g, dtype=torch.float64)
    z = sc.means3D[:, 2].double()
    x = ((2 * px + 1) / W - 1) * sc.camera.tanfovx * z
    y = ((2 * py + 1) / H - 1) * sc.camera.tanfovy * z
    sc.means3D[half, 0] = x[half].float()
    sc.means3D[half, 1] = y[half].float()
    return sc


@contextlib.contextmanager
def tile_sort(form):
    """HS_TILE_SORT=form for the forwards inside (read by the library at every forward)."""
    was = os.environ.get("HS_TILE_SORT")
    os.environ["HS_TILE_SORT"] = form
    try:
        yield
    finally:
        if was is None:
            os.environ.pop("HS_TILE_SORT", None)
        else:
            os.environ["HS_TILE_SORT"] = was


@pytest.mark.parametrize("form", ["hier", "radix"])
def test_skewed_scene_vs_oracle(oracle, form):
    """Half the cloud inside one super-tile: structure bit-exact, images and gradients within the
    contract, through the hierarchical tile sort and through the radix passes."""
    sc = skewed_scene()
    f, b = Hh.run_oracle(oracle, sc)
    with tile_sort(form):
        g = Hh.run_hip(sc)
    st = g["state"]
    R = f["R"]
    assert st["num_rendered"] == R
    assert int(st["tile_sort"]) == {"radix": 0, "hier": 2}[form]
    check_structure(st, f)
    assert np.array_equal(u32(st["offsets"]), u32(f["offsets"]))
    assert np.array_equal(st["keys_sorted"].view(np.uint64)[:R], f["keys_sorted"])
    assert np.array_equal(u32(st["point_list"][:R]), u32(f["point_list"]))
    assert np.array_equal(u32(st["ranges"]), u32(f["ranges"]))
    # the pair slots in depth order: inst_sorted is a permutation ordered by depth bits (ties by instance), offs_sorted
    # the inclusive scan of its tile counts
    order = np.lexsort((np.arange(len(f["depths"])), np.where(f["radii"] > 0, Hh.bits(f["depths"]), 0xFFFFFFFF)))
    vis = int((f["radii"] > 0).sum())
    assert np.array_equal(u32(st["inst_sorted"])[:vis], order[:vis])
    assert np.array_equal(u32(st["offs_sorted"]), np.cumsum(u32(f["tiles_touched"])[u32(st["inst_sorted"])]))
    lens = (f["ranges"][:, 1].astype(np.int64) - f["ranges"][:, 0]).reshape(24, 40)
    assert lens[8:16, 8:16].sum() > 0.4 * R            # ... the scene is as skewed as it says
    m = Hh.decision_masks(oracle, sc, [f], st, what="skewed")
    check_image(g["color"], f["color"], m, "skewed")
    Hh.assert_grads_close(g, b, what="skewed", at_risk=m["rows"], min_strict=0.99)   # (measured 1.0)


@pytest.mark.parametrize("cfg", ["c3", "c4"])
def test_hierarchical_tile_sort_equals_the_radix_passes_at_full_size(cfg):
    """BASELINE c3 and c4 through both tile sorts of large frames: every array the binning stage leaves -- point_list,
    ranges, the rebuilt sorted keys, the depth-ordered instance list and pair offsets, num_rendered -- and, downstream,
    images and every gradient, bit for bit.  (The hierarchical form is the default of large frames since round 6, hence the
    one test_c3_... / test_c4_... hold against the oracle; this test ties the radix passes to it.)"""
    P, W, H = 1_000_000, 1920, 1080
    sc = S.make_scene(P, W, H, 3, seed=0, hdr=True)
    cams = S.blur_poses(W, H, 8) if cfg == "c4" else None
    out = {}
    for form in ("radix", "hier"):
        with tile_sort(form):
            out[form] = Hh.run_hip(sc, cameras=cams, hdr=True)
        torch.cuda.empty_cache()
    a, b = out["radix"], out["hier"]
    assert int(a["state"]["tile_sort"]) == 0 and int(b["state"]["tile_sort"]) == 2
    R = a["state"]["num_rendered"]
    assert R == b["state"]["num_rendered"] and R > (50_000_000 if cfg == "c4" else 6_000_000)
    for k in ("point_list", "keys_sorted"):
        assert np.array_equal(a["state"][k][:R], b["state"][k][:R]), k
    for k in ("ranges", "inst_sorted", "offs_sorted", "n_contrib", "final_T", "tiles_touched"):
        assert np.array_equal(a["state"][k], b["state"][k]), k
    for k in a:
        if k != "state" and a[k] is not None:
            assert np.array_equal(Hh.bits(a[k]) if a[k].dtype == np.float32 else a[k], Hh.bits(b[k]) if b[k].dtype == np.float32 else b[k]), k


@contextlib.contextmanager
def environment(**kv):
    """os.environ[k] = v for the forwards inside (the library reads these switches at every forward)."""
    was = {k: os.environ.get(k) for k in kv}
    os.environ.update({k: str(v) for k, v in kv.items()})
    try:
        yield
    finally:
        for k, v in was.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _depth_sort_case(name):
    """Scenes for the depth sort by counting: (scene, cameras, hdr, extra environment, ranges expected off chip)."""
    if name == "c2":
        return S.make_scene(100_000, 800, 800, 0, seed=2), None, False, {}, False
    if name == "c3":
        return S.make_scene(1_000_000, 1920, 1080, 3, seed=0, hdr=True), None, True, {}, False
    if name == "wild":          # culled instances (behind the camera, off screen), duplicates (equal keys), huge and tiny ones
        return Hh.make_wild(S.make_scene(30_000, 640, 384, 1, seed=5), np.random.default_rng(3)), None, False, {}, False
    if name == "five_poses":    # 250 000 instances: the largest frame in blocks of 1024
        sc = S.make_scene(50_000, 480, 320, 1, seed=7, hdr=True)
        return sc, S.blur_poses(480, 320, 5, step=0.02), True, {}, False
    if name == "300k":          # blocks of 4096, 74 rows
        return S.make_scene(300_000, 1024, 768, 0, seed=11), None, False, {}, False
    if name == "tiny":
        return S.make_scene(37, 96, 64, 2, seed=13), None, False, {}, False
    if name == "mostly_culled":  # nine tenths of the cloud behind the camera: they must not become one bucket of the sort
        sc = S.make_scene(200_000, 640, 480, 0, seed=17)
        behind = torch.rand(200_000, generator=torch.Generator().manual_seed(1)) < 0.9
        sc.means3D[behind, 2] = -sc.means3D[behind, 2]
        return sc, None, False, {}, False
    if name == "all_culled":    # the camera looks away: no visible instance, no varying bit, nothing to sort
        sc = S.make_scene(5_000, 320, 240, 0, seed=17)
        sc.means3D[:, 2] = -sc.means3D[:, 2]
        return sc, None, False, {}, False
    if name == "one_depth":     # every Gaussian at z = 5 exactly: no varying bit at all, ONE bucket of 20 000 equal keys
        sc = S.make_scene(20_000, 320, 240, 0, seed=19)
        k = 5.0 / sc.means3D[:, 2:3]
        sc.means3D = sc.means3D * k
        sc.scales = sc.scales * k
        sc.means3D[:, 2] = 5.0
        return sc, None, False, {}, True
    if name == "two_depths":    # a wall at two depths 2 ulp apart: two buckets of 10 000 equal keys -> ranges that cannot fit
        sc = S.make_scene(20_000, 320, 240, 0, seed=19)
        k = 5.0 / sc.means3D[:, 2:3]
        sc.means3D = sc.means3D * k
        sc.scales = sc.scales * k
        z = torch.full((20_000,), 5.0)
        z[::2] = float(np.nextafter(np.nextafter(np.float32(5.0), np.float32(6.0)), np.float32(6.0)))
        sc.means3D[:, 2] = z
        return sc, None, False, {}, True
    if name == "wild_cap64":    # every range of more than 64 instances goes the slow way (through memory, chunk by chunk)
        return Hh.make_wild(S.make_scene(30_000, 640, 384, 1, seed=5), np.random.default_rng(3)), None, False, {"HS_DEPTH_RANGE_CAP": 64}, True
    if name == "c2_passes":     # no distribution sort: every range takes the passes over (instance number, key)
        return S.make_scene(100_000, 800, 800, 0, seed=2), None, False, {"HS_DEPTH_DIST_MAX": 0}, False
    if name == "wild_passes":
        return Hh.make_wild(S.make_scene(30_000, 640, 384, 1, seed=5), np.random.default_rng(3)), None, False, {"HS_DEPTH_DIST_MAX": 0}, False
    if name == "300k_cap2000":  # ... and chunks of 4096 at that: ranges of 2049 .. ~2400 instances in one chunk, none in two
        return S.make_scene(300_000, 1024, 768, 0, seed=11), None, False, {"HS_DEPTH_RANGE_CAP": 2000}, True
    raise KeyError(name)


@pytest.mark.parametrize("name", ["c2", "wild", "five_poses", "300k", "tiny", "mostly_culled", "all_culled", "one_depth", "two_depths", "wild_cap64",
                                  "300k_cap2000", "c2_passes", "wild_passes", "c3"])
def test_depth_sort_by_counting_equals_the_look_back_passes(name):
    """Round 6: frames below 2^21 instances sort their instances by depth with one stable counting pass over the top twelve
    varying key bits and range sorts inside the LDS instead of three or four look-back passes.  Same order of the visible
    instances (a stable sort has one; the culled ones, which have no pairs, stand behind them in index order instead of among
    the largest keys), and every array downstream -- pair offsets, point_list, ranges, keys, images, gradients -- bit for bit.
    The cases: BASELINE c2 and c3, culled instances and duplicate keys, several poses, both block sizes, a cloud smaller than a
    wave, nine tenths culled, no varying bit, two huge buckets of equal keys (ranges that do not fit on chip: sorted through
    memory by their workgroup, counted, and the host goes back to the passes), the off-chip path forced onto ordinary
    frames with one chunk and with several, and the in-LDS passes (what a range with clustered keys takes instead of the
    distribution sort) forced onto every range."""
    from casualhdrsplat_amd import _lib as L
    lib = L.load()
    sc, cams, hdr, extra, off_chip = _depth_sort_case(name)
    out = {}
    try:
        with environment(HS_DEPTH_SORT="lsd"):
            out["lsd"] = Hh.run_hip(sc, cameras=cams, hdr=hdr)
        with environment(HS_DEPTH_SORT="msd", **extra):
            if off_chip:
                with pytest.warns(RuntimeWarning, match="counting depth sort"):
                    out["msd"] = Hh.run_hip(sc, cameras=cams, hdr=hdr)
                assert lib.hs_depth_sort(-1) == 0      # the host went back to the passes for the rest of the process
            else:
                out["msd"] = Hh.run_hip(sc, cameras=cams, hdr=hdr)
                assert lib.hs_depth_sort(-1) == 1
    finally:
        lib.hs_depth_sort(1)
    a, b = out["lsd"], out["msd"]
    sa, sb = a["state"], b["state"]
    assert sa["depth_slow_ranges"] == 0 and (sb["depth_slow_ranges"] > 0) == off_chip, (sa["depth_slow_ranges"], sb["depth_slow_ranges"])
    R = sa["num_rendered"]
    assert R == sb["num_rendered"] and (R > 0) == (name != "all_culled")
    vis = sa["radii"] > 0
    assert np.array_equal(vis, sb["radii"] > 0)
    ia, ib = sa["inst_sorted"].astype(np.int64), sb["inst_sorted"].astype(np.int64)
    assert np.array_equal(np.sort(ia), np.arange(ia.size)) and np.array_equal(np.sort(ib), np.arange(ib.size))   # permutations
    assert np.array_equal(ia[vis[ia]], ib[vis[ib]])                                  # the visible instances: the same order
    n_vis = int(vis.sum())
    assert vis[ib[:n_vis]].all() and np.array_equal(ib[n_vis:], np.flatnonzero(~vis))     # counting form: culled last, by index
    assert np.array_equal(sa["offs_sorted"][vis[ia]], sb["offs_sorted"][vis[ib]])
    for k in ("point_list", "keys_sorted"):
        assert np.array_equal(sa[k][:R], sb[k][:R]), k
    for k in ("ranges", "n_contrib", "final_T", "tiles_touched", "offsets"):
        assert np.array_equal(sa[k], sb[k]), k
    for k in a:
        if k != "state" and a[k] is not None:
            assert np.array_equal(Hh.bits(a[k]) if a[k].dtype == np.float32 else a[k], Hh.bits(b[k]) if b[k].dtype == np.float32 else b[k]), k


@pytest.mark.parametrize("W,H,n_poses,want", [(2048, 1024, 2, 2), (2176, 2048, 1, 2), (2048, 1024, 16, 2), (2048, 1152, 16, 0)])
def test_hierarchical_tile_sort_at_its_key_limits(W, H, n_poses, want):
    """The (pose, super-tile) key table of the hierarchical tile sort at its edges: exactly 256 keys (the last frame whose
    coarse sort is ONE radix pass), 272 (the first with two), exactly 2048 (the largest table: hs_common.h kHierStMax), and
    2304 -- where HS_TILE_SORT=hier must quietly give the radix passes, as the default does.  Every array the binning stage
    leaves, images and gradients against the radix passes, bit for bit (those are held to the oracle elsewhere)."""
    P = 6000
    sc = S.make_scene(P, W, H, 1, seed=61)
    cams = S.blur_poses(W, H, n_poses, step=0.01) if n_poses > 1 else None
    out = {}
    for form in ("radix", "hier"):
        with tile_sort(form):
            out[form] = Hh.run_hip(sc, cameras=cams)
    a, b = out["radix"], out["hier"]
    assert int(a["state"]["tile_sort"]) == 0 and int(b["state"]["tile_sort"]) == want
    R = a["state"]["num_rendered"]
    assert R == b["state"]["num_rendered"] and R > 4 * P
    for k in ("point_list", "keys_sorted"):
        assert np.array_equal(a["state"][k][:R], b["state"][k][:R]), k
    for k in ("ranges", "inst_sorted", "offs_sorted", "n_contrib", "final_T", "tiles_touched"):
        assert np.array_equal(a["state"][k], b["state"][k]), k
    for k in a:
        if k != "state" and a[k] is not None:
            assert np.array_equal(Hh.bits(a[k]) if a[k].dtype == np.float32 else a[k], Hh.bits(b[k]) if b[k].dtype == np.float32 else b[k]), k


@pytest.mark.parametrize("P,W,H,n_poses", [(150000, 320, 240, 1), (3000, 1024, 1024, 1), (700, 256, 256, 16)])
def test_counting_tile_sort_edges_vs_oracle(oracle, P, W, H, n_poses):
    """The tile sort of small frames by counting (binning.hip) at its edges: 586 emission workgroups (more than eight rows
    per row slot of the column scan: the rows are walked twice instead of kept in registers), exactly 4096 tiles (the
    largest key table; every thread of the scatter owns 16 keys), 16 poses x 256 tiles (keys = pose * tiles + tile up to
    the limit).  Structure bit-exact against the oracle, and the same bits with the radix passes forced."""
    sc = S.make_scene(P, W, H, 1, seed=41)
    cams = S.blur_poses(W, H, n_poses, step=0.01) if n_poses > 1 else [sc.camera]
    fs = [Hh.run_oracle(oracle, sc, cam=c, backward=False)[0] for c in cams]
    R = sum(f["R"] for f in fs)
    pl = np.concatenate([f["point_list"].astype(np.int64) + k * P for k, f in enumerate(fs)])
    off = np.cumsum([0] + [f["R"] for f in fs])
    ranges = np.concatenate([np.where((f["ranges"][:, 1] > f["ranges"][:, 0])[:, None], f["ranges"].astype(np.int64) + off[k], 0)
                             for k, f in enumerate(fs)])
    got = {}
    for mode in ("count", "radix"):
        os.environ.pop("HS_TILE_SORT", None)
        if mode == "radix":
            os.environ["HS_TILE_SORT"] = "radix"
        try:
            g = Hh.run_hip(sc, cameras=cams if n_poses > 1 else None, capacity=R + 5)
        finally:
            os.environ.pop("HS_TILE_SORT", None)
        st = g["state"]
        assert st["num_rendered"] == R
        assert np.array_equal(st["point_list"][:R].astype(np.int64), pl), mode
        assert np.array_equal(st["ranges"].astype(np.int64), ranges), mode
        got[mode] = (st["keys_sorted"][:R].copy(), g["color"].copy())
    assert np.array_equal(got["count"][0], got["radix"][0]) and np.array_equal(Hh.bits(got["count"][1]), Hh.bits(got["radix"][1]))


def _with_depths(sc, z_new):
    """The same cloud as seen on screen, moved to the depths `z_new` (means and scales scale with z / z_old)."""
    z_new = torch.as_tensor(z_new, dtype=torch.float32)
    k = (z_new / sc.means3D[:, 2])[:, None]
    sc.means3D = sc.means3D * k
    sc.scales = sc.scales * k
    return sc


@pytest.mark.parametrize("case", ["all_equal", "two_values", "lowest_bit", "nine_bits", "wide", "half_culled", "one_visible"])
def test_depth_sort_layouts_vs_oracle(oracle, case):
    """The depth sort covers the bits of the depth keys that VARY only, in digits of up to nine bits, with the layout derived
    on the device (binning.hip): every layout it can arrive at -- one pass (all depths equal, a single varying bit, exactly
    nine bits), two, three and four passes (depths over five orders of magnitude), culled instances among the keys (they carry
    the all-ones key and stay out of the layout), a single visible instance -- must give the oracle's sorted list bit for bit."""
    P, W, H = 3000, 192, 128
    sc = S.make_scene(P, W, H, 1, seed=31)
    g = torch.Generator().manual_seed(5)
    z0 = torch.tensor(4.0)
    bits0 = int(z0.view(torch.int32))
    if case == "all_equal":
        z = torch.full((P,), 4.0)
    elif case == "two_values":
        z = torch.where(torch.rand(P, generator=g) < 0.5, torch.tensor(4.0), torch.tensor(4.5))
    elif case == "lowest_bit":       # neighbouring floats: the keys differ in bit 0 only
        z = (torch.full((P,), bits0, dtype=torch.int32) + torch.randint(0, 2, (P,), generator=g, dtype=torch.int32)).view(torch.float32)
    elif case == "nine_bits":        # exactly the nine lowest bits vary: one digit
        z = (torch.full((P,), bits0, dtype=torch.int32) + torch.randint(0, 512, (P,), generator=g, dtype=torch.int32)).view(torch.float32)
    elif case == "wide":             # 0.25 ... 2.5e4: the exponent field varies in five bits -> 30 varying bits, four passes
        z = 0.25 * torch.exp(torch.rand(P, generator=g) * 11.5)
    elif case == "half_culled":      # every other Gaussian behind the 0.2 cull distance
        z = 2.0 + 8.0 * torch.rand(P, generator=g)
    else:
        z = 2.0 + 8.0 * torch.rand(P, generator=g)
    sc = _with_depths(sc, z)
    if case == "half_culled":
        sc.means3D[::2, 2] = 0.1
    if case == "one_visible":
        sc.means3D[1:, 2] = -1.0
    f, _ = Hh.run_oracle(oracle, sc, backward=False)
    gh = Hh.run_hip(sc)
    st = gh["state"]
    R = f["R"]
    assert st["num_rendered"] == R and (R > 0)
    check_structure(st, f)
    assert np.array_equal(st["keys_sorted"].view(np.uint64)[:R], f["keys_sorted"])
    assert np.array_equal(u32(st["point_list"][:R]), u32(f["point_list"]))
    assert np.array_equal(u32(st["ranges"]), u32(f["ranges"]))
    if case == "half_culled":
        assert int((f["radii"] > 0).sum()) <= P // 2
    if case == "one_visible":
        assert int((f["radii"] > 0).sum()) == 1
    # ... and the sync-free single-enqueue forward (preprocess writes the keys and ORs them) sorts the same list
    gc = Hh.run_hip(sc, capacity=R + 1000)
    assert np.array_equal(u32(gc["state"]["point_list"][:R]), u32(f["point_list"]))
    assert np.array_equal(gh["color"], gc["color"])


def test_crf_gradient_blur_domains_run_to_run_and_tiny_gradients():
    """The fixed-point CRF-gradient accumulation scales itself to each block's largest |dL/dLDR|: a loss gradient
    eight orders of magnitude smaller gives the same table gradient up to that factor (no underflow to zero), for
    both averaging domains of the N-pose blur."""
    sc = S.make_scene(6000, 200, 120, 1, seed=4, hdr=True)
    cams = S.blur_poses(200, 120, 3, step=0.02)
    for dom in ("ldr", "hdr"):
        a = Hh.run_hip(sc, cameras=cams, hdr=True, blur_domain=dom)
        b = Hh.run_hip(sc, cameras=cams, hdr=True, blur_domain=dom)
        assert np.array_equal(a["d_crf_table"], b["d_crf_table"]) and np.array_equal(a["d_exposure"], b["d_exposure"])
        small = S.make_scene(6000, 200, 120, 1, seed=4, hdr=True)
        small.dL_dimage = sc.dL_dimage * 2.0 ** -27   # power of two: every float product scales exactly
        c = Hh.run_hip(small, cameras=cams, hdr=True, blur_domain=dom)
        assert np.array_equal(c["d_crf_table"] * 2.0 ** 27, a["d_crf_table"]), dom


def test_device_radix_sort_matches_stable_reference():
    import ctypes as C
    from casualhdrsplat_amd import _lib as L
    lib = L.load()
    gen = torch.Generator().manual_seed(0)
    for n, nbits in [(1, 40), (63, 45), (4097, 45), (300001, 45), (1 << 20, 48), (50000, 33), (70000, 64)]:
        hi = torch.randint(0, 1 << 16, (n,), generator=gen, dtype=torch.int64)
        lo = torch.randint(0, 1 << 8, (n,), generator=gen, dtype=torch.int64)  # few distinct low bits: many ties
        keys = ((hi << 32) | (lo << 12)) & ((1 << min(nbits, 62)) - 1)
        vals = torch.arange(n, dtype=torch.int32)
        kd, vd = keys.cuda(), vals.cuda()
        ko, vo = torch.empty_like(kd), torch.empty_like(vd)
        tmp = torch.empty(int(lib.hs_sort_tmp_bytes(n)), dtype=torch.uint8, device="cuda")
        L.check(lib.hs_sort_pairs(kd.data_ptr(), vd.data_ptr(), ko.data_ptr(), vo.data_ptr(), n, nbits, tmp.data_ptr(),
                                  torch.cuda.current_stream().cuda_stream), "hs_sort_pairs")
        order = torch.sort(keys, stable=True).indices
        assert torch.equal(ko.cpu(), keys[order]) and torch.equal(vo.cpu().long(), order), (n, nbits)


def test_radix_look_back_gives_up_instead_of_hanging():
    """SURVEY.md section 5 (bounded spins): HS_FAULT_INJECT=sort_ticket starts the first pass of hs_sort_pairs with ticket 1,
    so chain position 0 never publishes its status words -- the situation a damaged scratch array produces.  The blocks
    behind it must stop polling after their bound, report 2 in the fail word and return; the call must not hang."""
    import subprocess
    import sys
    code = r"""
import os, sys, time, torch
sys.path.insert(0, os.environ["HS_ROOT"])
from casualhdrsplat_amd import _lib as L
lib = L.load()
n = 3 * 4096 + 17
keys = torch.randint(0, 1 << 40, (n,), dtype=torch.int64, device="cuda")
vals = torch.arange(n, dtype=torch.int32, device="cuda")
ko, vo = torch.empty_like(keys), torch.empty_like(vals)
tmp = torch.zeros(int(lib.hs_sort_tmp_bytes(n)), dtype=torch.uint8, device="cuda")
t0 = time.time()
L.check(lib.hs_sort_pairs(keys.data_ptr(), vals.data_ptr(), ko.data_ptr(), vo.data_ptr(), n, 40, tmp.data_ptr(),
                          torch.cuda.current_stream().cuda_stream), "hs_sort_pairs")
torch.cuda.synchronize()
print("FAIL-WORD", int(tmp[:8].view(torch.int32)[1]), "SECONDS %.1f" % (time.time() - t0))
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for inject, want in (("sort_ticket", 2), ("", 0)):
        env = dict(os.environ, HS_ROOT=root, HS_FAULT_INJECT=inject, HS_LIB_PATH=_TEST_LIB)   # (the only build with the hooks)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        assert f"FAIL-WORD {want} " in r.stdout, r.stdout[-500:]


def test_blocks_behind_a_late_block_help_themselves_and_the_frame_is_unchanged():
    """HS_FAULT_INJECT=late_block: block 1 of every blockIdx-ordered radix pass starts ~3 ms late, as when other kernels
    leave its XCD no room.  The blocks behind it must not just wait: they count the late block's digits themselves, publish
    them and go on (hs_counters.reserved[4] > 0) -- and the frame (sorted lists, image, gradients) is the one of a normal
    run, bit for bit."""
    import subprocess
    import sys
    code = r"""
import os, sys, hashlib, numpy as np
sys.path.insert(0, os.environ["HS_ROOT"]); sys.path.insert(0, os.path.join(os.environ["HS_ROOT"], "tests"))
import helpers as Hh
from casualhdrsplat_amd import synthetic as S
g = Hh.run_hip(S.make_scene(60000, 640, 400, 1, seed=3), capacity=900000)
st = g["state"]; R = st["num_rendered"]
h = hashlib.sha256()
for a in (st["point_list"][:R], st["ranges"], g["color"], g["d_means3D"], g["d_shs"]):
    h.update(np.ascontiguousarray(a).tobytes())
print("FRAME", R, h.hexdigest(), "HELPS", st["look_back_helps"])
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for inject in ("", "late_block"):
        # (a frame this small sorts its instances by counting and its pairs by counting: HS_DEPTH_SORT=lsd keeps the look-back
        # passes of the depth sort, the only chains such a frame can have)
        env = dict(os.environ, HS_ROOT=root, HS_FAULT_INJECT=inject, HS_SORT_TICKETS="0", HS_DEPTH_SORT="lsd", HS_LIB_PATH=_TEST_LIB)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        outs[inject] = [ln for ln in r.stdout.splitlines() if ln.startswith("FRAME")][0].split()
    assert outs[""][1:3] == outs["late_block"][1:3] and int(outs[""][1]) > 100000, outs
    assert int(outs[""][4]) == 0 and int(outs["late_block"][4]) > 0, outs


def test_stalled_sort_chain_switches_to_ticket_order_and_the_step_is_repeated():
    """What two processes sharing a GPU can do to the blockIdx-ordered radix passes (each keeps the other's waited-for
    blocks out until both give up: hs_counters.overflow = 2, empty frame) -- provoked by HS_FAULT_INJECT=stalled_chain,
    which plants that verdict while the passes are blockIdx-ordered.  The host must switch the library to ticket order
    and (a) repeat a forward nobody differentiates by itself, (b) raise SortChainStalled -- a BinningOverflow, so existing
    handlers repeat the step -- from a training step's backward, in the synchronous mode as in the sync-free one."""
    import subprocess
    import sys
    code = r"""
import os, sys, warnings, torch
sys.path.insert(0, os.environ["HS_ROOT"]); sys.path.insert(0, os.path.join(os.environ["HS_ROOT"], "tests"))
import helpers as Hh
from casualhdrsplat_amd import synthetic as S, _lib as L, BinningOverflow, SortChainStalled, GaussianRasterizer
sc = S.make_scene(5000, 200, 136, 1, seed=3)
lib = L.load()
rs, _, _ = Hh.settings_from_scene(sc, "cuda")
leaf = {k: getattr(sc, k).cuda() for k in ("means3D", "opacities", "shs", "scales", "rotations")}
def call(rast, grad):
    for v in leaf.values(): v.requires_grad_(grad)
    return rast(leaf["means3D"], torch.zeros(5000, 3, device="cuda"), leaf["opacities"], shs=leaf["shs"],
                scales=leaf["scales"], rotations=leaf["rotations"])
for capacity in (None, 200000):
    lib.hs_sort_tickets(0)
    rast = GaussianRasterizer(rs, capacity=capacity)
    # (b) training step: the backward reports, the repeated step succeeds
    out = call(rast, True)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        try:
            out[0].sum().backward()
            print("NO-RAISE")
        except SortChainStalled as e:
            assert isinstance(e, BinningOverflow)
            print("RAISED", len(w))
    assert lib.hs_sort_tickets(-1) == 1
    out = call(rast, True)
    out[0].sum().backward()
    good = out[0].detach().clone()
    assert float(good.abs().sum()) > 0 and rast.last_num_rendered > 0
    # (a) a forward without a backward: repeated transparently
    lib.hs_sort_tickets(0)
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        img = call(rast, False)[0]
    assert torch.equal(img, good), float((img - good).abs().max())
    assert lib.hs_sort_tickets(-1) == 1 and rast.overflow_replays >= 1
    print("OK", capacity)
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HS_ROOT=root, HS_FAULT_INJECT="stalled_chain", HS_LIB_PATH=_TEST_LIB)
    env.pop("HS_SORT_TICKETS", None)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count("RAISED 1") == 2 and "OK None" in r.stdout and "OK 200000" in r.stdout, r.stdout[-1000:]


def test_overflow_found_by_a_backward_reaches_the_rasterizer_after_other_forwards():
    """The capacity request of an overflowed training step lives in the rasterizer, not in the bookkeeping of its latest
    forward: a second forward (an eval render, another view) between the overflowing forward and its backward must not
    lose it."""
    from casualhdrsplat_amd import BinningOverflow
    sc = S.make_scene(20000, 320, 200, 1, seed=5)
    R = Hh.run_hip(sc)["state"]["num_rendered"]
    rast, leaf, call = _rasterizer_and_inputs(sc, R // 2, requires_grad=True)
    out_a = call()                      # overflows (training forward: nobody looks yet)
    out_b = call()                      # another forward of the same rasterizer replaces its `_last` bookkeeping
    with pytest.raises(BinningOverflow):
        (out_a[0] * sc.dL_dimage.cuda()).sum().backward()
    del out_b
    out = call()                        # grown by the failed backward's verdict although it was not the latest forward
    assert rast.capacity >= R and rast.last_num_rendered == R
    (out[0] * sc.dL_dimage.cuda()).sum().backward()


def test_softplus_gradient_of_dark_gaussians_vs_oracle(oracle):
    """radiance_activation='softplus' with SH sums far below zero (colour 1e-11 .. 4e-8): the SH-gradient rows of those
    Gaussians are ~colour x basis x dL/dcolour, far below the tensor-wide floor of the other tests -- compared here row
    by row, relative to the row (1 - expf(-colour) would make them exactly zero)."""
    P = 3000
    sc = S.make_scene(P, 160, 112, 1, seed=33)
    dark = torch.arange(P) % 3 == 0
    sc.shs[dark, 0] = torch.linspace(-25.0, -17.0, int(dark.sum()))[:, None] / 0.28209479177387814
    sc.shs[dark, 1:] = 0.0
    g = Hh.run_hip(sc, radiance_activation="softplus")
    f, b = Hh.run_oracle(oracle, sc, radiance_activation="softplus")
    want, got = b["dL_dshs"][:, 0], g["d_shs"][:, 0]
    rows = dark.numpy() & (f["radii"] > 0) & (np.abs(want).max(axis=1) > 0)
    assert rows.sum() > 500 and 1e-13 < f["rgb"][rows].max() < 1e-7
    rel = np.abs(got[rows] - want[rows]).max(axis=1) / np.abs(want[rows]).max(axis=1)
    assert np.percentile(rel, 99) < 1e-3 and rel.max() < 5e-2, (float(np.percentile(rel, 99)), float(rel.max()))


def test_ticketed_radix_passes_and_both_emission_scans_give_the_same_frame():
    """HS_SORT_TICKETS=1 (chain positions of the radix passes from a per-pass ticket counter instead of blockIdx: the
    fallback should a dispatcher ever start workgroups out of order) sorts exactly as the default; and the pair emission
    lays the same pairs out whether its block offsets come from the kernels ahead of it (small frames) or from its own
    chained scan (HS_SCAN_IN_EMISSION=1: the path of frames with several million instances)."""
    import subprocess
    import sys
    code = r"""
import os, sys, hashlib, numpy as np
sys.path.insert(0, os.environ["HS_ROOT"]); sys.path.insert(0, os.path.join(os.environ["HS_ROOT"], "tests"))
import helpers as Hh
from casualhdrsplat_amd import synthetic as S
g = Hh.run_hip(S.make_scene(60000, 640, 400, 1, seed=3), capacity=900000)
st = g["state"]; R = st["num_rendered"]
h = hashlib.sha256()
for a in (st["point_list"][:R], st["ranges"], g["color"], g["d_means3D"], g["d_shs"]):
    h.update(np.ascontiguousarray(a).tobytes())
print("FRAME", R, h.hexdigest())
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for env in (dict(HS_SORT_TICKETS="0", HS_SCAN_IN_EMISSION="0"), dict(HS_SORT_TICKETS="1", HS_SCAN_IN_EMISSION="0"),
                dict(HS_SORT_TICKETS="0", HS_SCAN_IN_EMISSION="1")):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, HS_ROOT=root, **env))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        outs.append([ln for ln in r.stdout.splitlines() if ln.startswith("FRAME")][0])
    assert outs[0] == outs[1] == outs[2] and int(outs[0].split()[1]) > 100000, outs


def test_full_size_properties_c3():
    """BASELINE c3 size (1M Gaussians, 1080p, deg 3, HDR): properties that need no oracle."""
    from casualhdrsplat_amd import GaussianRasterizer, inspect_state
    dev = "cuda"
    P, W, H = 1_000_000, 1920, 1080
    sc = S.make_scene(P, W, H, 3, seed=0, hdr=True)
    rs, expo, crf = Hh.settings_from_scene(sc, dev, hdr=True, requires_grad=True)
    leaves = [t.to(dev).requires_grad_(True) for t in (sc.means3D, torch.zeros(P, 3), sc.opacities, sc.shs, sc.scales, sc.rotations)]
    rast = GaussianRasterizer(rs)

    def run(dL):
        for t in leaves + [expo, crf]:
            t.grad = None
        out = rast(leaves[0], leaves[1], leaves[2], shs=leaves[3], scales=leaves[4], rotations=leaves[5])
        st = inspect_state(out[0])
        torch.autograd.backward(out[0], grad_tensors=dL)
        return out, st, [t.grad.clone() for t in leaves]

    dL = sc.dL_dimage.to(dev)
    out, st, g1 = run(dL)
    R = st["num_rendered"]
    keys = st["keys_sorted"][:R]
    assert bool((keys[1:] >= keys[:-1]).all())                              # sortedness
    assert R == int(st["tiles_touched"].to(torch.int64).sum())                # len(keys) == sum tiles_touched
    rng = st["ranges"].to(torch.int64)
    lens = rng[:, 1] - rng[:, 0]
    assert int(lens.sum()) == R                                               # ranges partition [0, R)
    nz = rng[lens > 0]
    assert bool((nz[1:, 0] == nz[:-1, 1]).all()) and int(nz[0, 0]) == 0 and int(nz[-1, 1]) == R
    tiles = keys >> 32
    assert int(tiles.max()) < 120 * 68
    pl = st["point_list"][:R].to(torch.int64)
    assert int(torch.bincount(pl, minlength=P).sub(st["tiles_touched"].to(torch.int64)).abs().max()) == 0
    assert bool(torch.isfinite(out[0]).all()) and bool((out[2] >= 0).all())
    assert bool((st["n_contrib"].to(torch.int64)[0].reshape(-1) <= lens.max()).all())
    # linearity of the backward in dL/dimage: grad(2*dL) == 2*grad(dL) bit for bit (power-of-two scaling)
    _, _, g2 = run(2 * dL)
    for a, b in zip(g1, g2):
        assert torch.equal(2 * a, b)
    # checksum of checksums stays finite and non-trivial
    assert all(bool(torch.isfinite(a).all()) for a in g1) and float(g1[3].abs().sum()) > 0


def test_gradients_share_one_flat_buffer():
    """The backward hands autograd views of ONE flat fp32 buffer, so the multi-GPU step can all-reduce it in
    place (casualhdrsplat_amd.distributed._shared_flat)."""
    from casualhdrsplat_amd import GaussianRasterizer
    from casualhdrsplat_amd.distributed import _shared_flat
    sc = S.make_scene(3000, 128, 96, 3, seed=1, hdr=True)
    rs, expo, crf = Hh.settings_from_scene(sc, "cuda", hdr=True, requires_grad=True)
    leaves = [t.cuda().requires_grad_(True) for t in (sc.means3D, torch.zeros(3000, 3), sc.opacities, sc.shs, sc.scales, sc.rotations)]
    out = GaussianRasterizer(rs)(leaves[0], leaves[1], leaves[2], shs=leaves[3], scales=leaves[4], rotations=leaves[5])
    torch.autograd.backward(out[0], grad_tensors=sc.dL_dimage.cuda())
    grads = [t.grad for t in leaves + [expo, crf]]
    assert all(g is not None for g in grads)
    flat = _shared_flat(grads)
    assert flat is not None and flat.numel() >= sum(g.numel() for g in grads)
    before = leaves[3].grad.clone()
    flat.mul_(2.0)   # what an in-place all-reduce would do
    assert torch.equal(leaves[3].grad, 2 * before)


@pytest.mark.parametrize("free", [False, True], ids=["yaw_cameras", "free_cameras"])
@pytest.mark.parametrize("n_poses", [1, 3])
def test_camera_pose_gradients_vs_autograd(n_poses, free):
    """dL/d(viewmatrix, projmatrix, campos) -- the reference optimises camera motion jointly
    (Readme.md:54).  Checked against float64 autograd through the pure-PyTorch rasterizer."""
    from casualhdrsplat_amd import GaussianRasterizationSettings, GaussianRasterizer
    from oracle import torch_rasterizer as TR
    P, W, H, deg = 500, 96, 80, 2
    if free:   # free 6-DoF base camera; the poses roll, pitch and yaw by a degree each and shift
        base = S.random_camera(W, H, 7)
        sc = S.make_scene(P, W, H, deg, seed=12, place_in=base)
        cams = S.perturbed_poses(base, n_poses, seed=2, rot_step_deg=1.0, step=0.02)
    else:
        sc = S.make_scene(P, W, H, deg, seed=12)
        cams = [S.yaw_camera(W, H, 2.0 * k - 1.0) for k in range(n_poses)]
    dev = "cuda"
    V = torch.stack([c.viewmatrix for c in cams]).to(dev).requires_grad_(True)
    PV = torch.stack([c.projmatrix for c in cams]).to(dev).requires_grad_(True)
    C = torch.stack([c.campos for c in cams]).to(dev).requires_grad_(True)
    kw = dict(viewmatrices=V, projmatrices=PV, camposes=C) if n_poses > 1 else {}
    rs = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=cams[0].tanfovx, tanfovy=cams[0].tanfovy, bg=sc.bg.to(dev),
        scale_modifier=1.0, viewmatrix=V[0] if n_poses == 1 else cams[0].viewmatrix.to(dev),
        projmatrix=PV[0] if n_poses == 1 else cams[0].projmatrix.to(dev), sh_degree=deg,
        campos=C[0] if n_poses == 1 else cams[0].campos.to(dev), prefiltered=False, debug=False, **kw)
    m = sc.means3D.to(dev).requires_grad_(True)
    out = GaussianRasterizer(rs)(m, torch.zeros_like(m), sc.opacities.to(dev), shs=sc.shs.to(dev),
                                 scales=sc.scales.to(dev), rotations=sc.rotations.to(dev))
    (out[0] * sc.dL_dimage.to(dev)).sum().backward()
    got = [V.grad.cpu().double().numpy(), PV.grad.cpu().double().numpy(), C.grad.cpu().double().numpy()]

    dt = torch.float64
    Vr = torch.stack([c.viewmatrix for c in cams]).to(dt).requires_grad_(True)
    PVr = torch.stack([c.projmatrix for c in cams]).to(dt).requires_grad_(True)
    Cr = torch.stack([c.campos for c in cams]).to(dt).requires_grad_(True)
    imgs = []
    for k in range(n_poses):
        view = TR.View(W, H, cams[k].tanfovx, cams[k].tanfovy, Vr[k], PVr[k], Cr[k])
        imgs.append(TR.rasterize(view, sc.means3D.to(dt), sc.opacities.to(dt), deg, sc.bg, shs=sc.shs.to(dt),
                                 scales=sc.scales.to(dt), rotations=sc.rotations.to(dt)))
    (torch.stack(imgs).mean(dim=0) * sc.dL_dimage.to(dt)).sum().backward()
    want = [Vr.grad.numpy(), PVr.grad.numpy(), Cr.grad.numpy()]
    for name, g, w in zip(("viewmatrix", "projmatrix", "campos"), got, want):
        scale = np.abs(w).max()
        assert scale > 0
        assert np.abs(g - w).max() <= 2e-4 * scale, (name, float(np.abs(g - w).max() / scale))
    # entries the projection never reads stay exactly zero
    assert np.all(got[0].reshape(n_poses, 16)[:, [3, 7, 11, 15]] == 0)
    assert np.all(got[1].reshape(n_poses, 16)[:, [2, 6, 10, 14]] == 0)


def test_indefinite_precomputed_covariance_power_rule(oracle):
    """A user-supplied cov3D_precomp need not be positive semi-definite; the resulting indefinite conic makes
    `power > 0` reachable and the published rule (skip the pixel) must hold, also under sub-tile culling."""
    sc = S.make_scene(1500, 144, 96, 0, seed=13)
    f0, _ = Hh.run_oracle(oracle, sc, backward=False)
    cov = torch.from_numpy(f0["cov3D"].copy())
    gen = torch.Generator().manual_seed(3)
    pick = torch.randperm(1500, generator=gen)[:300]
    cov[pick, 0] *= -0.5     # negative xx variance: 2-D covariance keeps det != 0 but loses definiteness for many
    cols = torch.rand(1500, 3, generator=gen)
    g = Hh.run_hip(sc, use_colors_precomp=cols, use_cov_precomp=cov)
    f, b = Hh.run_oracle(oracle, sc, use_colors_precomp=cols, use_cov_precomp=cov)
    co = f["conic_opacity"]
    indefinite = (co[:, 0] * co[:, 2] - co[:, 1] ** 2 <= 0) | (co[:, 0] <= 0)
    assert int((indefinite & (f["radii"] > 0)).sum()) > 10          # the case is actually exercised
    st = g["state"]
    check_structure(st, f)
    assert np.array_equal(u32(st["point_list"][:f["R"]]), u32(f["point_list"]))
    m = Hh.decision_masks(oracle, sc, [f], st, what="indefinite")
    check_image(g["color"], f["color"], m, "indefinite")
    # indefinite conics let G = exp(power) grow along the negative-curvature direction until the alpha cap: elements are
    # worse conditioned in fp32 than with proper covariances (max 5e-2, L2 5e-5 instead of 1e-2 / 1e-5) -- on every row
    # no differing pixel reaches, whether or not the frame has flips
    Hh.assert_grads_close(g, b, keys=[("means3D", "dL_dmeans3D"), ("opacities", "dL_dopacity"),
                                      ("colors_precomp", "dL_dcolors_precomp"), ("cov3D_precomp", "dL_dcov3D")],
                          max_tol=5e-2, l2_tol=5e-5, at_risk=m["rows"], min_strict=0.99)   # (measured 1.0)


def test_accumulated_opacity_output_and_gradient():
    """SURVEY.md 8(f) n3 (alpha part): optional A = 1 - final_T output with gradient, against fp64 autograd."""
    from casualhdrsplat_amd import GaussianRasterizer
    from oracle import torch_rasterizer as TR
    P, W, H, deg = 800, 112, 80, 1
    sc = S.make_scene(P, W, H, deg, seed=14)
    sc.bg = torch.tensor([0.3, 0.1, 0.6])
    rs, _, _ = Hh.settings_from_scene(sc, "cuda")
    gen = torch.Generator().manual_seed(2)
    gA = torch.randn(H, W, generator=gen)
    leaves = {k: getattr(sc, k).cuda().requires_grad_(True) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
    out = GaussianRasterizer(rs, return_alpha=True)(leaves["means3D"], torch.zeros(P, 3, device="cuda"), leaves["opacities"],
                                                    shs=leaves["shs"], scales=leaves["scales"], rotations=leaves["rotations"])
    assert len(out) == 3 and out[2].shape == (H, W)
    ((out[0] * sc.dL_dimage.cuda()).sum() + (out[2] * gA.cuda()).sum()).backward()
    dt = torch.float64
    cam = sc.camera
    view = TR.View(W, H, cam.tanfovx, cam.tanfovy, cam.viewmatrix.to(dt), cam.projmatrix.to(dt), cam.campos.to(dt))
    ref = {k: getattr(sc, k).to(dt).clone().requires_grad_(True) for k in leaves}
    color, st = TR.rasterize(view, ref["means3D"], ref["opacities"], deg, sc.bg, shs=ref["shs"], scales=ref["scales"],
                             rotations=ref["rotations"], return_state=True)
    alpha = 1.0 - st["final_T"]
    ((color * sc.dL_dimage.to(dt)).sum() + (alpha * gA.to(dt)).sum()).backward()
    assert Hh.rel_err(out[2].detach().cpu().numpy(), alpha.detach().numpy(), 1e-3)[0] <= 1e-4
    got = {"d_" + k: v.grad.cpu().numpy() for k, v in leaves.items()}
    want = {"dL_d" + k: v.grad.numpy() for k, v in ref.items()}
    Hh.assert_grads_close(got, want, keys=[(k, "dL_d" + k) for k in leaves], frac_tol=2e-2, l2_tol=1e-4)


def _render_grads(sc, device, cameras=None, defer=False):
    """Backward through the rasterizer; returns (leaf dict, rasterizer)."""
    from casualhdrsplat_amd import GaussianRasterizer
    rs, _, _ = Hh.settings_from_scene(sc, device, cameras)
    leaf = {k: t.clone().to(device).requires_grad_(True) for k, t in
            dict(means3D=sc.means3D, means2D=torch.zeros_like(sc.means3D), opacities=sc.opacities, shs=sc.shs,
                 scales=sc.scales, rotations=sc.rotations).items()}
    rast = GaussianRasterizer(rs, defer_sh_grad=defer)
    out = rast(leaf["means3D"], leaf["means2D"], leaf["opacities"], shs=leaf["shs"], scales=leaf["scales"],
               rotations=leaf["rotations"])
    (out[0] * sc.dL_dimage.to(device)).sum().backward()
    return leaf, rast


@pytest.mark.parametrize("deg,n_poses", [(3, 1), (2, 1), (1, 3)])
def test_deferred_sh_gradient_equals_direct(deg, n_poses):
    """defer_sh_grad + exchange_view_gradients (world size 1) reproduces the direct backward: bit for bit with one
    pose (same arithmetic, one view), to rounding with several (poses summed in the same order, other kernel)."""
    from casualhdrsplat_amd.distributed import exchange_view_gradients
    sc = S.make_scene(3000, 160, 120, deg, seed=5)
    cams = S.blur_poses(160, 120, n_poses, step=0.02) if n_poses > 1 else None
    ref, _ = _render_grads(sc, "cuda", cams)
    got, rast = _render_grads(sc, "cuda", cams, defer=True)
    assert got["shs"].grad is None and rast.deferred["view_colors"].shape == (n_poses, 3000, 3)
    n = exchange_view_gradients([v for k, v in got.items() if k != "shs"], got["shs"], rast.deferred)
    assert n == {"all_reduced": 0, "all_gathered": 0}
    for k in ref:
        a, b = got[k].grad.cpu().numpy(), ref[k].grad.cpu().numpy()
        if n_poses == 1:
            assert np.array_equal(Hh.bits(a), Hh.bits(b)), k
        else:
            assert np.allclose(a, b, rtol=1e-5, atol=1e-6 * np.abs(b).max()), k
    assert np.abs(ref["shs"].grad.cpu().numpy()[:, 1:]).max() > 0


@pytest.mark.parametrize("deg,M", [(0, 1), (2, 9), (3, 16), (1, 16)])
def test_sh_backward_views_kernel_vs_reference(deg, M):
    """hs_sh_backward_views against plain tensor algebra (oracle/torch_rasterizer.sh_backward_views), 5 views."""
    from casualhdrsplat_amd.rasterizer import sh_backward_views
    from oracle import torch_rasterizer as TR
    g = torch.Generator().manual_seed(11)
    P, V = 2500, 5
    means = torch.randn(P, 3, generator=g) * 3
    cams = torch.randn(V, 3, generator=g) * 5 + 10
    vc = torch.randn(V, P, 3, generator=g)
    got = sh_backward_views(means.cuda(), cams.cuda(), vc.cuda(), M, deg).cpu()
    want = TR.sh_backward_views(means.double(), cams.double(), vc.double(), M, deg)
    assert got.shape == (P, M, 3)
    assert torch.allclose(got.double(), want, rtol=1e-5, atol=1e-5)
    if M > (deg + 1) ** 2:
        assert torch.count_nonzero(got[:, (deg + 1) ** 2:]) == 0


@pytest.mark.parametrize("world", [2, 4])
def test_view_parallel_step_two_ranks_on_one_gpu(world):
    """World size 2 and 4 (gloo, all ranks on this GPU): every exchange strategy of bench.py leaves each rank with the
    sum of all views' gradients.  The RCCL run over xGMI is the driver's 8-GPU bench; this covers the logic."""
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_gpu_worker.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(port), worker],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("VIEW-EXCHANGE-OK") == world, r.stdout[-2000:]


def test_backward_in_gaussian_chunks_is_bit_identical_to_the_whole_backward():
    """hs_bwd_args.g_begin / g_end (the chunked exchange of a view-parallel step, distributed.chunked_all_reduce): the
    per-Gaussian half of the backward run in ascending chunks of the Gaussians writes bit for bit the gradients of the
    single launch -- per-Gaussian rows, the camera-pose gradients (whose per-workgroup partial rows are reduced after the
    last chunk) and the densification statistics.  Single rank: no collective runs, the chunks do."""
    from casualhdrsplat_amd import DensifyStats, GaussianRasterizer
    dev = "cuda"
    sc = S.make_scene(5000, 224, 144, 3, seed=17, hdr=True)
    cams = S.blur_poses(224, 144, 3, step=0.02)
    res = []
    for chunks in (0, 1, 3, 64):
        rs, expo, crf = Hh.settings_from_scene(sc, dev, cams, hdr=True, requires_grad=True)
        rs = rs._replace(viewmatrices=rs.viewmatrices.clone().requires_grad_(True),
                         projmatrices=rs.projmatrices.clone().requires_grad_(True),
                         camposes=rs.camposes.clone().requires_grad_(True))
        leaf = [t.clone().to(dev).requires_grad_(True) for t in (sc.means3D, torch.zeros_like(sc.means3D), sc.opacities,
                                                                  sc.shs, sc.scales, sc.rotations)]
        dens = DensifyStats(5000, dev)
        rast = GaussianRasterizer(rs, densify_stats=dens, reduce_group=True if chunks else None, reduce_chunks=chunks)
        out = rast(leaf[0], leaf[1], leaf[2], shs=leaf[3], scales=leaf[4], rotations=leaf[5])
        (out[0] * sc.dL_dimage.to(dev)).sum().backward()
        assert rast.finish_reduce() == 0   # one rank: nothing on the wire
        res.append([t.grad.clone() for t in leaf + [expo, crf, rs.viewmatrices, rs.projmatrices, rs.camposes]] +
                   [dens.grad_accum.clone(), dens.denom.clone(), dens.max_radii.clone()])
    assert float(res[0][3].abs().sum()) > 0 and float(res[0][8].abs().sum()) > 0
    for other in res[1:]:
        for a, b in zip(res[0], other):
            assert torch.equal(a, b)


def test_view_parallel_step_two_ranks_over_rccl():
    """The same check over backend "nccl" (= RCCL), one GPU per rank -- runs wherever the box has two GPUs."""
    import socket
    import subprocess
    import sys
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (the driver's 8-GPU node); single-GPU boxes run the gloo variant above")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_gpu_worker.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", HS_DIST_BACKEND="nccl")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), worker],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("VIEW-EXCHANGE-OK") == 2, r.stdout[-2000:]


def test_bench_bare_multi_gpu_invocation_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no torchrun environment (the shape the driver may use): bench.py starts its
    own ranks and rank 0's JSON line comes back through the parent.  RCCL when the box has two GPUs, else both ranks
    share the GPU over gloo (plumbing only)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    if torch.cuda.device_count() < 2:
        env["HS_BENCH_BACKEND"] = "gloo"
        # (over gloo on a shared GPU an exchange costs 50-300 ms against a 0.6 ms step: with the default cap of 20 x only
        # the fallback would survive the probe; the plumbing check wants to see several strategies timed)
        env["HS_BENCH_PROBE_CAP_X"] = "100000"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", "c2", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline", "--kernel-iters", "2"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    ex = d["config"]["gradient_exchange"]
    assert ex["choice"] in ex["step_ms"] and len(ex["step_ms"]) >= 2, ex
    # the analytic cost of every library strategy rides along, so the measured probe can be read against it
    assert set(ex["model_ms"]) >= {"allreduce", "allreduce_overlap", "views", "views_overlap"}
    assert ex["model_ms"]["allreduce"]["all_reduce_ring_ms"] > ex["model_ms"]["views"]["all_reduce_ring_ms"] > 0
    # one coalesced collective per chunk of the overlapped all-reduce (counted by the run, not assumed): <= chunks + 2
    cps = ex["collectives_per_step"]
    assert cps["allreduce/rccl"] == 1 and cps["views/rccl"] == 3
    assert "allreduce_overlap/rccl" not in ex["step_ms"] or 1 <= cps["allreduce_overlap/rccl"] <= 4 + 2, cps
    # every strategy that was kept stayed under the probe's cap; the dropped ones say why
    assert all(t <= ex["first_step_cap_ms"] for t in ex["step_ms"].values()) and isinstance(ex["dropped"], dict), ex
    assert ex["bytes"]["allreduce/rccl"]["sent_per_rank_bytes"] == 100_000 * 14 * 4   # c2, SH degree 0, two ranks: 2 * 1/2 * payload
    # HS_BENCH_EXCHANGE pins a strategy and skips the probe (what an unattended 8-GPU run can fall back on)
    env["HS_BENCH_EXCHANGE"] = "views/rccl"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", "c2", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline", "--kernel-iters", "2"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    ex = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])["config"]["gradient_exchange"]
    assert ex["choice"] == "views/rccl" and ex["pinned"] is True and ex["step_ms"] == {}, ex


def test_whole_step_as_a_hip_graph_reproduces_the_eager_step():
    """graphs.GraphedStep: forward + loss + backward of the sync-free rasterizer captured once and replayed (one launch
    per step instead of ~40).  The replay gives the eager step's bits -- image and every gradient -- also after the inputs
    were updated in place (what an optimizer does), and an overflowing frame is reported by check_overflow()."""
    from casualhdrsplat_amd import BinningOverflow, GaussianRasterizer
    from casualhdrsplat_amd.graphs import GraphedStep
    sc = S.make_scene(30000, 400, 240, 2, seed=19, hdr=True)
    rs, expo, crf = Hh.settings_from_scene(sc, "cuda", hdr=True, requires_grad=True)
    names = ("means3D", "opacities", "shs", "scales", "rotations")
    leaf = {k: getattr(sc, k).cuda().requires_grad_(True) for k in names}
    m2 = torch.zeros(30000, 3, device="cuda", requires_grad=True)
    dL = sc.dL_dimage.cuda()
    plist = list(leaf.values()) + [m2, expo, crf]
    R = Hh.run_hip(sc, hdr=True)["state"]["num_rendered"]

    def make_step(rast):
        def step():
            for p in plist:
                p.grad = None
            out = rast(leaf["means3D"], m2, leaf["opacities"], shs=leaf["shs"], scales=leaf["scales"], rotations=leaf["rotations"])
            torch.autograd.backward([out[0], out[2]], grad_tensors=[dL, 0.5 * dL])
            return out
        return step

    eager = make_step(GaussianRasterizer(rs, capacity=R + 5000))
    rast_g = GaussianRasterizer(rs, capacity=R + 5000)
    g = GraphedStep(make_step(rast_g), [rast_g], params=plist)
    for rep in range(3):
        if rep:   # an "optimizer step": inputs change in place, addresses stay
            with torch.no_grad():
                leaf["means3D"].add_(0.002 * torch.randn_like(leaf["means3D"]))
                leaf["opacities"].mul_(0.98)
        out_g = g.step()
        got = [out_g[0].clone(), out_g[2].clone()] + [t.clone() for t in g.grads]   # (the eager step below rebinds p.grad)
        n_g = g.check_overflow()[0]
        out_e = eager()
        want = [out_e[0].detach(), out_e[2].detach()] + [p.grad for p in plist]
        assert n_g > 0
        for a, b in zip(got, want):
            assert torch.equal(a, b), rep
    # a frame that outgrows the captured capacity: rendered empty, and check_overflow() says so
    small = GaussianRasterizer(rs, capacity=R + 5000)
    gs = GraphedStep(make_step(small), [small])
    with torch.no_grad():
        leaf["scales"].mul_(3.0)    # nine times the footprint: far more (tile, Gaussian) pairs than the capacity
    gs.step()
    with pytest.raises(BinningOverflow, match="rebuild the GraphedStep"):
        gs.check_overflow()
    with pytest.raises(ValueError, match="capacity"):
        GraphedStep(lambda: None, [GaussianRasterizer(rs)])


def test_graphed_step_checks_every_forward_of_a_rasterizer_not_only_the_last():
    """a captured step may call ONE rasterizer several times (several views, an eval render in between).  The
    overflow counters of every one of those forwards are copied out inside the graph and looked at by check_overflow():
    a first call that outgrows the capacity (its frame renders empty, its gradients are zero) is reported even though the
    rasterizer's latest forward fitted."""
    from casualhdrsplat_amd import BinningOverflow, GaussianRasterizer
    from casualhdrsplat_amd.graphs import GraphedStep
    sc = S.make_scene(20000, 320, 200, 1, seed=21)
    rs, _, _ = Hh.settings_from_scene(sc, "cuda")
    names = ("means3D", "opacities", "shs", "scales", "rotations")
    leaf = {k: getattr(sc, k).cuda().requires_grad_(True) for k in names}
    other_scales = sc.scales.clone().cuda().requires_grad_(True)     # the scales the FIRST call of a step renders with
    m2 = torch.zeros(20000, 3, device="cuda", requires_grad=True)
    dL = sc.dL_dimage.cuda()
    plist = list(leaf.values()) + [m2, other_scales]
    R = Hh.run_hip(sc)["state"]["num_rendered"]
    rast = GaussianRasterizer(rs, capacity=R + 5000)

    def step():
        for p in plist:
            p.grad = None
        first = rast(leaf["means3D"], m2, leaf["opacities"], shs=leaf["shs"], scales=other_scales, rotations=leaf["rotations"])
        second = rast(leaf["means3D"], m2, leaf["opacities"], shs=leaf["shs"], scales=leaf["scales"], rotations=leaf["rotations"])
        torch.autograd.backward([first[0], second[0]], grad_tensors=[dL, dL])
        return first[0], second[0]

    g = GraphedStep(step, [rast])
    imgs = g.step()
    counts = g.check_overflow()
    assert len(counts) == 2 and counts[0] == counts[1] == R        # both forwards of the one rasterizer are tracked
    assert torch.equal(imgs[0].detach(), imgs[1].detach()) and float(imgs[0].detach().abs().sum()) > 0
    # now the FIRST call outgrows the capacity (nine times the footprint) while the second -- the rasterizer's latest forward -- fits
    with torch.no_grad():
        other_scales.mul_(3.0)
    imgs = g.step()
    with pytest.raises(BinningOverflow):
        g.check_overflow()
    assert float(imgs[0].detach().abs().sum()) == 0 and float(imgs[1].detach().abs().sum()) > 0   # (the empty frame, the one that fitted)


def test_steps_do_not_leak_device_memory():
    """The saved state must not reference the outputs (a cycle through the autograd node is invisible to Python's
    collector): device memory after 3 steps equals device memory after 9."""
    import gc
    sc = S.make_scene(20000, 320, 240, 1, seed=3, hdr=True)

    def steps(n):
        for _ in range(n):
            Hh.run_hip(sc, hdr=True, capacity=400000)
        gc.collect()
        torch.cuda.synchronize()
        return torch.cuda.memory_allocated()

    a = steps(3)
    b = steps(6)
    assert b <= a, (a, b)


@pytest.mark.parametrize("antialiasing,n_poses", [(False, 1), (True, 1), (True, 3)])
def test_inverse_depth_output_and_antialiasing_vs_autograd(antialiasing, n_poses):
    """expected inverse-depth image (third output of newer published rasterizers) with gradient,
    and the `antialiasing` opacity compensation, against float64 autograd through the pure-PyTorch rasterizer
    (poses averaged).  Camera-pose gradients of the depth term are covered through the view matrices."""
    from casualhdrsplat_amd import GaussianRasterizationSettings, GaussianRasterizer
    from oracle import torch_rasterizer as TR
    P, W, H, deg = 700, 112, 80, 1
    sc = S.make_scene(P, W, H, deg, seed=21)
    cams = [S.yaw_camera(W, H, 1.5 * k - 1.0) for k in range(n_poses)]
    dev = "cuda"
    V = torch.stack([c.viewmatrix for c in cams]).to(dev).requires_grad_(True)
    PV = torch.stack([c.projmatrix for c in cams]).to(dev).requires_grad_(True)
    C = torch.stack([c.campos for c in cams]).to(dev).requires_grad_(True)
    rs = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=cams[0].tanfovx, tanfovy=cams[0].tanfovy, bg=sc.bg.to(dev),
        scale_modifier=1.0, viewmatrix=V[0], projmatrix=PV[0], sh_degree=deg, campos=C[0], prefiltered=False,
        debug=False, antialiasing=antialiasing,
        **(dict(viewmatrices=V, projmatrices=PV, camposes=C) if n_poses > 1 else {}))
    gen = torch.Generator().manual_seed(4)
    gD = torch.randn(H, W, generator=gen) * 5
    names = ("means3D", "opacities", "shs", "scales", "rotations")
    leaves = {k: getattr(sc, k).to(dev).requires_grad_(True) for k in names}
    out = GaussianRasterizer(rs, return_invdepth=True)(leaves["means3D"], torch.zeros(P, 3, device=dev), leaves["opacities"],
                                                       shs=leaves["shs"], scales=leaves["scales"], rotations=leaves["rotations"])
    assert len(out) == 3 and out[2].shape == (H, W)
    ((out[0] * sc.dL_dimage.to(dev)).sum() + (out[2] * gD.to(dev)).sum()).backward()

    dt = torch.float64
    ref = {k: getattr(sc, k).to(dt).clone().requires_grad_(True) for k in names}
    Vr = torch.stack([c.viewmatrix for c in cams]).to(dt).requires_grad_(True)
    PVr = torch.stack([c.projmatrix for c in cams]).to(dt).requires_grad_(True)
    Cr = torch.stack([c.campos for c in cams]).to(dt).requires_grad_(True)
    cols, invs = [], []
    for k in range(n_poses):
        view = TR.View(W, H, cams[k].tanfovx, cams[k].tanfovy, Vr[k], PVr[k], Cr[k])
        color, st = TR.rasterize(view, ref["means3D"], ref["opacities"], deg, sc.bg, shs=ref["shs"], scales=ref["scales"],
                                 rotations=ref["rotations"], return_state=True, antialiasing=antialiasing)
        cols.append(color)
        invs.append(st["invdepth"])
    color, invd = torch.stack(cols).mean(dim=0), torch.stack(invs).mean(dim=0)
    ((color * sc.dL_dimage.to(dt)).sum() + (invd * gD.to(dt)).sum()).backward()
    assert Hh.rel_err(out[0].detach().cpu().numpy(), color.detach().numpy(), 1e-2)[0] <= 1e-4
    assert Hh.rel_err(out[2].detach().cpu().numpy(), invd.detach().numpy(), 1e-3)[0] <= 1e-4
    assert float(invd.detach().max()) > 0.05
    got = {"d_" + k: v.grad.cpu().numpy() for k, v in leaves.items()}
    want = {"dL_d" + k: v.grad.numpy() for k, v in ref.items()}
    Hh.assert_grads_close(got, want, keys=[(k, "dL_d" + k) for k in names], frac_tol=2e-2, l2_tol=1e-4)
    for name, g, w in (("viewmatrix", V.grad, Vr.grad), ("projmatrix", PV.grad, PVr.grad), ("campos", C.grad, Cr.grad)):
        g, w = g.cpu().double().numpy(), w.numpy()
        assert np.abs(g - w).max() <= 3e-4 * np.abs(w).max(), (name, float(np.abs(g - w).max() / np.abs(w).max()))


def test_densification_statistics_in_kernel(tmp_path, oracle):
    """grad_accum / denom / max_radii are updated inside the backward exactly as a trainer would
    from means2D.grad and radii; the scene itself goes through the PLY exchange layout first."""
    from casualhdrsplat_amd import DensifyStats, GaussianRasterizer
    from casualhdrsplat_amd import scene_io as IO
    P, W, H, deg = 4000, 160, 120, 2
    sc = S.make_scene(P, W, H, deg, seed=8)
    sc.means3D[:50, 2] = -5.0                                   # behind the camera: never rasterized
    cloud = IO.GaussianCloud(sc.means3D, sc.shs, torch.logit(sc.opacities.clamp(1e-4, 1 - 1e-4)), torch.log(sc.scales),
                             sc.rotations * 1.7)
    IO.save_ply(str(tmp_path / "s.ply"), cloud)
    act = IO.load_ply(str(tmp_path / "s.ply")).activated("cuda")
    assert torch.allclose(act["scales"].cpu(), sc.scales, rtol=1e-6) and torch.allclose(act["rotations"].cpu(), sc.rotations, atol=1e-6)
    stats = DensifyStats(P)
    cams = [None, S.blur_poses(W, H, 3, step=0.05)]
    want_g, want_n, want_r = torch.zeros(P), torch.zeros(P), torch.zeros(P, dtype=torch.int32)
    # the same statistics from the ORACLE's screen-space gradient and radii (not from the HIP path's own outputs)
    orc_g, orc_n, orc_r = np.zeros(P), np.zeros(P), np.zeros(P, np.int32)
    for cameras in cams:
        ocs = cameras or [sc.camera]
        g2d, rad = np.zeros((P, 2)), np.zeros(P, np.int32)
        for cam in ocs:
            f_o, b_o = Hh.run_oracle(oracle, sc, cam=cam, dL=sc.dL_dimage.numpy() / len(ocs))
            g2d += b_o["dL_dmeans2D"][:, :2]
            rad = np.maximum(rad, f_o["radii"])
        seen = rad > 0
        orc_g += np.where(seen, np.linalg.norm(g2d, axis=1), 0.0)
        orc_n += seen
        orc_r = np.maximum(orc_r, rad)
    for cameras in cams:
        rs, _, _ = Hh.settings_from_scene(sc, "cuda", cameras)
        m2 = torch.zeros(P, 3, device="cuda", requires_grad=True)
        leaves = {k: v.clone().requires_grad_(True) for k, v in act.items()}
        out = GaussianRasterizer(rs, densify_stats=stats)(leaves["means3D"], m2, leaves["opacities"], shs=leaves["shs"],
                                                          scales=leaves["scales"], rotations=leaves["rotations"])
        (out[0] * sc.dL_dimage.cuda()).sum().backward()
        vis = (out[1] > 0).cpu()
        want_g += torch.where(vis, m2.grad[:, :2].norm(dim=1).cpu(), torch.zeros(P))
        want_n += vis.float()
        want_r = torch.maximum(want_r, out[1].cpu())
    assert torch.equal(stats.denom.cpu(), want_n) and torch.equal(stats.max_radii.cpu(), want_r)
    assert torch.allclose(stats.grad_accum.cpu(), want_g, rtol=1e-5, atol=1e-12)
    assert float(want_n[:50].sum()) == 0 and float(want_n.max()) == 2 and float(want_g.max()) > 0
    assert torch.allclose(stats.mean_grad().cpu(), want_g / want_n.clamp_min(1), rtol=1e-5, atol=1e-12)
    # against the oracle: counts and radii exactly, the gradient norms within the gradient contract
    assert np.array_equal(stats.denom.cpu().numpy(), orc_n) and np.array_equal(stats.max_radii.cpu().numpy(), orc_r)
    e = np.abs(stats.grad_accum.cpu().numpy() - orc_g) / np.maximum(orc_g, 1e-3 * np.sqrt((orc_g ** 2).mean()))
    assert np.percentile(e, 99.5) <= 1e-4 and e.max() <= 1e-2, (float(np.percentile(e, 99.5)), float(e.max()))
    stats.reset()
    assert float(stats.grad_accum.abs().sum()) == 0 and int(stats.max_radii.max()) == 0


# tiny clouds (P down to 1) put a handful of elements in a tensor: one fp32-ill-conditioned element is then a large
# fraction of it, so the sweep bounds the fraction more loosely than the fixed-size tests; the worst element and the
# L2 bar stay strict
# (worst element: 1.2e-2 over the 6000 configurations of rounds 1-2, 4.3e-2 -- one-list kernel -- / 3.7e-2 -- two-group
# kernel -- on sweep seed 22 case 156, a 72 x 50 frame of "wild" e^s-bright giants: an element far below the tensor's floor
# that shares its pixels with them carries their summation error)
SWEEP_BAR = dict(frac_tol=2e-2, l2_tol=2e-5, max_tol=6e-2)


def test_randomized_configurations_vs_oracle(oracle):
    """Sweep of small random configurations (sizes that are not multiples of the tile or block sizes, every SH degree,
    1-3 poses, with and without the HDR epilogue, both blur domains, the three radiance activations, sync and
    fixed-capacity binning): structure bit-exact, images and
    gradients within the numerical contract.  Catches indexing bugs that the handful of fixed shapes cannot."""
    # HS_SWEEP_SEED / HS_SWEEP_CASES: soak runs with other seeds and more cases (scripts/soak.sh)
    rng = np.random.default_rng(int(os.environ.get("HS_SWEEP_SEED", "2026")))
    for case in range(int(os.environ.get("HS_SWEEP_CASES", "24"))):
        c = Hh.sweep_case(rng, case)
        # (these frames are small: by default their pairs are sorted by counting; one case in three keeps the radix passes,
        # one in three takes the hierarchical sort of large frames -- round 6 -- down to a handful of super-tiles)
        os.environ.pop("HS_TILE_SORT", None)
        if case % 3:
            os.environ["HS_TILE_SORT"] = ("radix", "hier")[case % 3 - 1]
        # (and one case in four sorts its instances by depth with the look-back passes instead of by counting)
        os.environ.pop("HS_DEPTH_SORT", None)
        if case % 4 == 3:
            os.environ["HS_DEPTH_SORT"] = "lsd"
        P, W, H, n_poses, hdr, act, dom = c["P"], c["W"], c["H"], c["n_poses"], c["hdr"], c["act"], c["dom"]
        sc, cams, precomp, what = c["sc"], c["cams"], c["precomp"], c["what"]
        if hdr or n_poses > 1:
            if not hdr:  # linear-radiance blur: the average of the per-pose oracle renders
                fs = [Hh.run_oracle(oracle, sc, cam=c, backward=False, radiance_activation=act)[0] for c in cams]
                g = Hh.run_hip(sc, cameras=cams, backward=False, radiance_activation=act)
                ref = np.mean(np.stack([f["color"] for f in fs]), axis=0, dtype=np.float64)
                assert_image_close(g["color"], ref, what)
                assert np.array_equal(g["radii"], np.max(np.stack([f["radii"] for f in fs]), axis=0)), what
                continue
            r = Hh.run_oracle_hdr(oracle, sc, cams, dom, radiance_activation=act)
            Rtot = sum(f["R"] for f in r["fwd"])
            g = Hh.run_hip(sc, cameras=cams, hdr=True, blur_domain=dom, capacity=None if case % 2 else Rtot + 7,
                           radiance_activation=act)
            st = g["state"]
            assert st["num_rendered"] == Rtot, what
            for k, f in enumerate(r["fwd"]):
                check_structure(st, f, pose=k, P=P)
            pl = np.concatenate([f["point_list"].astype(np.int64) + k * P for k, f in enumerate(r["fwd"])])
            assert np.array_equal(u32(st["point_list"][:Rtot]), pl), what
            # decisions may differ from the oracle's only on pixels inside its threshold guard band; every gradient row off
            # the tile lists of the pixels where they DID differ (and off the pixels at a CRF knot) is held to the bar
            n_p = len(r["fwd"])
            got_imgs, ref_imgs = ([g["hdr"]], [r["hdr"]]) if (dom == "hdr" or n_p == 1) else \
                (list(st["pose_hdr"][:n_p]), [f["color"] for f in r["fwd"]])
            m = Hh.decision_masks(oracle, sc, r["fwd"], st, cams, crf_got=got_imgs, crf_ref=ref_imgs, what=what)
            for k, f in enumerate(r["fwd"]):
                check_image(st["pose_hdr"][k], f["color"], m, what, pose=k)
            assert_image_close(g["hdr"], r["hdr"], what)
            assert_image_close(g["color"], r["ldr"], what)
            Hh.assert_grads_close(g, r, what=what, at_risk=m["rows"], **SWEEP_BAR)
        else:
            pre, keys = {}, Hh.GRAD_KEYS
            if precomp:
                f0 = Hh.run_oracle(oracle, sc, backward=False)[0]
                pre = dict(use_colors_precomp=c["colors"], use_cov_precomp=torch.from_numpy(f0["cov3D"].copy()))
                keys = [("means3D", "dL_dmeans3D"), ("means2D", "dL_dmeans2D"), ("opacities", "dL_dopacity"),
                        ("colors_precomp", "dL_dcolors_precomp"), ("cov3D_precomp", "dL_dcov3D")]
            f, b = Hh.run_oracle(oracle, sc, radiance_activation=act, **pre)
            g = Hh.run_hip(sc, capacity=None if case % 2 else f["R"] + 1, radiance_activation=act, **pre)
            st = g["state"]
            assert st["num_rendered"] == f["R"], what
            check_structure(st, f)
            assert np.array_equal(u32(st["offsets"]), u32(f["offsets"])), what
            assert np.array_equal(u32(st["point_list"][:f["R"]]), u32(f["point_list"])), what
            assert np.array_equal(u32(st["ranges"]), u32(f["ranges"])), what
            m = Hh.decision_masks(oracle, sc, [f], st, what=what)
            check_image(g["color"], f["color"], m, what)
            Hh.assert_grads_close(g, b, keys=keys, what=what, at_risk=m["rows"], **SWEEP_BAR)
    os.environ.pop("HS_TILE_SORT", None)


def test_c_abi_from_a_plain_cpp_host(tmp_path, oracle):
    """The boundary is a C ABI with plain device pointers: examples/abi_host (hipMalloc, no PyTorch) runs
    plan / forward (with the upstream-style host read of num_rendered) / backward on a scene file and must give what
    the Python layer gives through ctypes -- bit for bit, same kernels -- and agree with the oracle."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "abi_host")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(root, "casualhdrsplat_amd", "csrc")], check=True, capture_output=True)
    P, W, H, deg = 2500, 200, 136, 2
    sc = S.make_scene(P, W, H, deg, seed=31)
    cam = sc.camera
    M = sc.shs.shape[1]
    with open(tmp_path / "scene.bin", "wb") as f:
        f.write(np.array([P, M, deg, W, H], np.int32).tobytes())
        f.write(np.array([cam.tanfovx, cam.tanfovy], np.float32).tobytes())
        for t in (sc.bg, cam.viewmatrix, cam.projmatrix, cam.campos, sc.means3D, sc.opacities, sc.shs, sc.scales,
                  sc.rotations, sc.dL_dimage):
            f.write(t.contiguous().numpy().astype(np.float32).tobytes())
    r = subprocess.run([exe, str(tmp_path / "scene.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = open(tmp_path / "out.bin", "rb").read()
    R = int(np.frombuffer(raw, np.uint32, 1)[0])
    off = 4

    def take(n, dt=np.float32):
        nonlocal off
        a = np.frombuffer(raw, dt, n, off)
        off += 4 * n
        return a

    color = take(3 * H * W).reshape(3, H, W)
    radii = take(P, np.int32)
    got = {"d_means3D": take(P * 3).reshape(P, 3), "d_means2D": take(P * 3).reshape(P, 3), "d_opacities": take(P).reshape(P, 1),
           "d_shs": take(P * M * 3).reshape(P, M, 3), "d_scales": take(P * 3).reshape(P, 3), "d_rotations": take(P * 4).reshape(P, 4)}
    assert off == len(raw)
    g = Hh.run_hip(sc)
    assert R == g["state"]["num_rendered"] and np.array_equal(radii, g["radii"])
    assert np.array_equal(Hh.bits(color), Hh.bits(g["color"]))
    for k, v in got.items():
        assert np.array_equal(Hh.bits(v), Hh.bits(g[k].reshape(v.shape))), k
    f, b = Hh.run_oracle(oracle, sc)
    assert R == f["R"]
    check_image(color, f["color"], Hh.decision_masks(oracle, sc, [f], g["state"], what="abi_host"), "abi_host")


def test_full_size_properties_c4_eight_poses():
    """BASELINE c4 size (1M Gaussians, 1080p, 8 virtual poses per frame, HDR + CRF): oracle-free properties.  Pose 0
    of the batched launch equals the single-pose launch bit for bit (same per-tile lists, same kernels), the tile
    lists partition the pairs of all 8 poses, and the blur average of identical poses is the sharp image."""
    from casualhdrsplat_amd import GaussianRasterizer, inspect_state
    dev = "cuda"
    P, W, H, N = 1_000_000, 1920, 1080, 8
    sc = S.make_scene(P, W, H, 3, seed=0, hdr=True)
    cams = S.blur_poses(W, H, N)
    leaves = [t.to(dev) for t in (sc.means3D, torch.zeros(P, 3), sc.opacities, sc.shs, sc.scales, sc.rotations)]

    def render(cameras):
        rs, _, _ = Hh.settings_from_scene(sc, dev, cameras, hdr=True)
        ls = [t.clone().requires_grad_(True) for t in leaves]
        out = GaussianRasterizer(rs)(ls[0], ls[1], ls[2], shs=ls[3], scales=ls[4], rotations=ls[5])
        return out, inspect_state(out[0]), ls

    out8, st8, ls8 = render(cams)
    R = st8["num_rendered"]
    I = N * P
    assert R == int(st8["tiles_touched"].to(torch.int64).sum()) and st8["tiles_touched"].numel() == I
    keys = st8["keys_sorted"][:R]
    assert bool((keys[1:] >= keys[:-1]).all()) and int((keys >> 32).max()) < N * 120 * 68
    rng = st8["ranges"].to(torch.int64)
    lens = rng[:, 1] - rng[:, 0]
    assert int(lens.sum()) == R and rng.shape[0] == N * 120 * 68
    pl = st8["point_list"][:R].to(torch.int64)
    assert int(pl.max()) < I and int(torch.bincount(pl, minlength=I).sub(st8["tiles_touched"].to(torch.int64)).abs().max()) == 0
    out1, st1, _ = render(None)                                    # single pose = the scene camera = cams[0]
    assert torch.equal(st8["n_contrib"][0], st1["n_contrib"][0]) and torch.equal(st8["final_T"][0], st1["final_T"][0])
    assert torch.equal(st8["radii"][:P], st1["radii"])
    assert bool(torch.isfinite(out8[0]).all()) and bool(torch.isfinite(out8[2]).all())
    (out8[0] * sc.dL_dimage.to(dev)).sum().backward()
    assert all(bool(torch.isfinite(t.grad).all()) for t in ls8) and float(ls8[3].grad.abs().sum()) > 0
    del out8, st8, ls8
    torch.cuda.empty_cache()
    same, _, _ = render([cams[0]] * N)
    assert Hh.rel_err(same[0].detach().cpu().numpy(), out1[0].detach().cpu().numpy(), 1e-3)[0] <= 1e-5


def _parity_row(name, sc, m, g, r, extra=None, masked=None):
    """HS_PARITY_JSON=<file>: append this full-size frame's row (profiles/r04_parity_table.json: differing pixels, share
    of rows on the strict bar, per-tensor error of the HIP gradients against the C oracle's on those rows)."""
    path = os.environ.get("HS_PARITY_JSON")
    if not path:
        return
    import json
    rows = ~m["rows"]
    row = {"case": name, "P": int(rows.size), "guard_band_pixels": int(m["pix_risk"].sum()), "n_differ": m["n_differ"],
           "n_knot_pixels": m["n_knot_pixels"], "strict_share": float(rows.mean()), "tensors": {}}
    for gk, rk in Hh.GRAD_KEYS:
        ref = np.asarray(r[rk], np.float64)
        got = np.asarray(g["d_" + gk], np.float64).reshape(ref.shape)
        floor = Hh.grad_floor(ref)
        e = np.abs(got[rows] - ref[rows]) / np.maximum(np.abs(ref[rows]), floor)
        row["tensors"][gk] = {"max": float(e.max()), "p999": float(np.percentile(e, 99.9)),
                              "frac_gt_1e-4": float((e > 1e-4).mean()),
                              "l2": float(np.linalg.norm(got[rows] - ref[rows]) / max(np.linalg.norm(ref), 1e-30))}
    row.update(extra or {})
    if masked is not None:
        # the masked pass: dL zeroed on the excluded pixels on both sides -- EVERY row on the bar, every element inside the bound
        g2, r2, bounded = masked
        mrow = {"excluded_pixels": int(m["excluded"].sum()), "strict_share": 1.0, "c_bound": Hh.C_BOUND,
                "n_outside_bound": int(sum(v[0] for v in bounded.values())),
                "c_needed": {k: v[1] for k, v in bounded.items()}, "tensors": {}}
        for gk, rk in Hh.GRAD_KEYS:
            ref = np.asarray(r2[rk], np.float64)
            got = np.asarray(g2["d_" + gk], np.float64).reshape(ref.shape)
            e = np.abs(got - ref) / np.maximum(np.abs(ref), Hh.grad_floor(ref))
            mrow["tensors"][gk] = {"max": float(e.max()), "p999": float(np.percentile(e, 99.9)),
                                   "frac_gt_1e-4": float((e > 1e-4).mean()),
                                   "l2": float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30))}
        row["masked_pass"] = mrow
    old = json.load(open(path)) if os.path.exists(path) else {"cases": []}
    old["cases"] = [c for c in old["cases"] if c["case"] != name] + [row]
    json.dump(old, open(path, "w"), indent=1)


def _masked_pass(oracle, sc, m, fwds, what, cameras=None, workers=1, frac_tol=1e-3, max_tol=1e-2, l2_tol=2e-6):
    """the second backward with dL zeroed on the pixels where a decision differed and on the CRF-knot
    pixels, on BOTH sides.  No row is excused: every Gaussian on the (full-size) bar, every element inside
    1e-4 |ref| + C_BOUND 2^-24 sqrt(n) sum|terms| (zero outside), CRF-table / exposure gradients against the oracle's
    own within 1e-4 of max(|ref|, tensor RMS) and inside the stated fixed-point + interval-weight bound."""
    g2, r2, dLm = Hh.masked_backward_pass(oracle, sc, m, fwds, cameras=cameras, workers=workers, what=what)
    rep = Hh.assert_grads_close(g2, r2, what=what + " masked", frac_tol=frac_tol, max_tol=max_tol, l2_tol=l2_tol)
    bounded = Hh.assert_grads_bounded(g2, r2, what=what + " masked")
    tab = np.asarray(r2["dL_dcrf_table"], np.float64)
    err = np.abs(np.asarray(g2["d_crf_table"], np.float64) - tab)
    rms = float(np.sqrt((tab ** 2).mean()))
    assert float((err / np.maximum(np.abs(tab), rms)).max()) <= 1e-4, (what, "d_crf_table", float((err / np.maximum(np.abs(tab), rms)).max()))
    imgs = [f["color"] for f in fwds]
    bound = 1e-4 * np.abs(tab) + Hh.crf_grad_bound(sc, imgs, dLm)
    assert not (err > bound).any(), (what, "d_crf_table entries outside the stated bound", int((err > bound).sum()), float((err / bound).max()))
    assert float(g2["d_exposure"]) == pytest.approx(r2["dL_dexposure"], rel=1e-4, abs=1e-4 * float(np.abs(dLm).sum()) * 1e-3)
    return g2, r2, bounded, rep


@pytest.mark.parametrize("cam_seed", [None, 9], ids=["default_camera", "free_camera"])
def test_c3_full_size_vs_oracle(oracle, cam_seed):
    """(free_camera, round 6: the same cloud laid out in front of a free 6-DoF view -- synthetic.random_camera(9): roll, pitch
    and yaw of up to +-pi, every entry of the view matrix populated -- at the size the metric is quoted on.)
    BASELINE c3 itself -- the frame bench.py times: 1M Gaussians, 1920 x 1080, SH degree 3, HDR radiance + CRF, seed 0
    -- against the C oracle (25 s of one host core): depths / screen positions / conics / radii / tile counts and the whole
    sorted list (6.78 M pairs: point_list, ranges, the rebuilt 64-bit keys) bit for bit, decisions confined to the
    oracle's guard band, radiance and LDR images within 1e-4 off the pixels where a decision differs, every gradient on
    the STRICT bar for >= 95 % of the Gaussians, d_crf_table / d_exposure as in the golden test.  Paths only this size
    reaches: a depth sort of 245 blocks, 8160 tiles through the ordered list + the tail queue of the backward."""
    P, W, H = 1_000_000, 1920, 1080
    sc = free_camera_scene(P, W, H, 3, 0, cam_seed, hdr=True)
    r = Hh.run_oracle_hdr(oracle, sc)
    g = Hh.run_hip(sc, hdr=True)
    st, f = g["state"], r["fwd"][0]
    R = f["R"]
    assert st["num_rendered"] == R and R > 6_000_000
    assert int(st["tile_sort"]) == 2      # the hierarchical tile sort: the default of a frame this size
    check_structure(st, f)
    assert np.array_equal(u32(st["offsets"]), u32(f["offsets"]))
    assert np.array_equal(st["keys_sorted"].view(np.uint64)[:R], f["keys_sorted"])
    assert np.array_equal(u32(st["point_list"][:R]), u32(f["point_list"]))
    assert np.array_equal(u32(st["ranges"]), u32(f["ranges"]))
    assert np.array_equal(g["radii"], f["radii"])
    m = Hh.decision_masks(oracle, sc, [f], st, crf_got=[g["hdr"]], crf_ref=[r["hdr"]], what="c3")
    check_image(g["hdr"], r["hdr"], m, "c3 radiance")
    check_image(st["final_T"][0], f["final_T"], m, "c3 final_T")
    check_image(g["color"], r["ldr"], m, "c3 ldr")
    # at full size the bar is TIGHTER than helpers.STRICT: millions of elements give the fraction and the L2 their meaning.
    # Measured (profiles/r04_parity_fullsize.json): 12 differing pixels, 98.9 % of the rows strict, and on those <= 3.2e-4 of
    # the elements beyond 1e-4 relative (99.97 % within north_star's bar), worst element 6.7e-3, relative L2 5.4e-7
    rep = Hh.assert_grads_close(g, r, what="c3", at_risk=m["rows"], min_strict=0.985,     # (measured: 0.9888 / 0.9908 under the free camera)
                                frac_tol=1e-3, max_tol=1e-2, l2_tol=2e-6)
    # table / exposure gradients: against the oracle's tone-map backward given the same decisions (on the handful of
    # differing pixels the radiance the HIP path composited stands in), and loosely against the oracle's own
    tab, dexp = Hh.crf_grads_given_decisions(oracle, sc, m, [r["hdr"]], [g["hdr"]])
    assert Hh.rel_err(g["d_crf_table"], tab, 1e3 * Hh.grad_floor(tab))[0] <= 2e-4
    assert float(g["d_exposure"]) == pytest.approx(dexp, rel=2e-4, abs=1e-3)
    assert Hh.rel_err(g["d_crf_table"], r["dL_dcrf_table"], 1e3 * Hh.grad_floor(tab))[0] <= 2e-2
    # ... and the pass that excuses nothing: dL zeroed on the 12 + ~200 excluded pixels, both sides
    g2, r2, bounded, rep2 = _masked_pass(oracle, sc, m, [f], "c3")
    _parity_row("c3 (1M Gaussians, 1920x1080, SH3, HDR + CRF, seed 0" + ("" if cam_seed is None else ", free 6-DoF camera") +
                "): HIP vs fp32 C oracle, rows off the differing pixels",
                sc, m, g, r, {"R": int(R), "report": {k: v for k, v in rep.items()}}, masked=(g2, r2, bounded))


@pytest.mark.parametrize("free", [False, True], ids=["x_shift_poses", "free_rotating_poses"])
def test_c4_eight_poses_full_size_vs_oracle(oracle, free):
    """(free_rotating_poses, round 6: the eight poses roll, pitch and yaw by 0.03 degrees each and shift, about a free 6-DoF
    base camera -- synthetic.perturbed_poses -- with the cloud laid out in front of it.)
    BASELINE c4 itself: the c3 cloud seen from 8 virtual poses in ONE frame (8 M instances: the hierarchical tile sort
    over 1080 (pose, super-tile) keys -- two radix passes over 12 M elements --, depth-sort passes of 1954 blocks, 65 280
    virtual tiles in strip order + queue) against the C
    oracle run per pose (eight host threads): per-pose structure and the sorted list of all ~54 M pairs bit for bit,
    every pose's radiance image off the differing pixels, the blurred LDR / mean radiance, all gradients (sums over the
    eight poses) on the STRICT bar for >= 90 % of the Gaussians, CRF-table and exposure gradients."""
    P, W, H, N = 1_000_000, 1920, 1080, 8
    if free:
        base_cam = S.random_camera(W, H, 13)
        sc = S.make_scene(P, W, H, 3, seed=0, hdr=True, place_in=base_cam)
        cams = S.perturbed_poses(base_cam, N, seed=3, rot_step_deg=0.03, step=0.01)
    else:
        sc = S.make_scene(P, W, H, 3, seed=0, hdr=True)
        cams = S.blur_poses(W, H, N)
    workers = min(N, os.cpu_count() or 1)
    r = Hh.run_oracle_hdr(oracle, sc, cams, "ldr", workers=workers)
    g = Hh.run_hip(sc, cameras=cams, hdr=True, blur_domain="ldr")
    st = g["state"]
    Rs = [f["R"] for f in r["fwd"]]
    R = sum(Rs)
    assert st["num_rendered"] == R
    tiles = 120 * 68
    base = 0
    for k, f in enumerate(r["fwd"]):
        check_structure(st, f, pose=k, P=P)
        assert np.array_equal(u32(st["point_list"][base:base + Rs[k]]), u32(f["point_list"]) + k * P), k
        rg = u32(st["ranges"][k * tiles:(k + 1) * tiles])
        ref = u32(f["ranges"])
        full = ref[:, 1] > ref[:, 0]
        assert np.array_equal(rg[full], ref[full] + base), k
        assert np.array_equal(rg[~full, 0], rg[~full, 1]), k
        base += Rs[k]
    assert np.array_equal(g["radii"], np.max(np.stack([f["radii"] for f in r["fwd"]]), axis=0))
    m = Hh.decision_masks(oracle, sc, r["fwd"], st, cams, crf_got=list(st["pose_hdr"][:N]),
                          crf_ref=[f["color"] for f in r["fwd"]], what="c4", workers=workers)
    for k, f in enumerate(r["fwd"]):
        check_image(st["pose_hdr"][k], f["color"], m, f"c4 pose {k}", pose=k)
        check_image(st["final_T"][k], f["final_T"], m, f"c4 final_T {k}", pose=k)
    any_differs = m["differs"].any(axis=0)
    for name, got, ref in (("ldr", g["color"], r["ldr"]), ("hdr", g["hdr"], r["hdr"])):
        e = np.abs(np.asarray(got, np.float64) - ref) / np.maximum(np.abs(ref), 1e-2)
        assert not ((e > 1e-4).any(axis=0) & ~any_differs).any(), ("c4 " + name, float(e.max()))
    # (measured: 105 differing pixels over the eight poses, 93 % of the rows strict; on those <= 3e-4 of the elements beyond
    # 1e-4, worst element 9.5e-3 -- a sum over eight poses' worth of pixel terms --, relative L2 4.7e-7)
    rep = Hh.assert_grads_close(g, r, what="c4", at_risk=m["rows"], min_strict=0.925 if not free else 0.93, frac_tol=1e-3,   # (measured 0.9309 / 0.9355)
                                max_tol=2e-2, l2_tol=2e-6)
    tab, dexp = Hh.crf_grads_given_decisions(oracle, sc, m, [f["color"] for f in r["fwd"]], list(st["pose_hdr"][:N]))
    assert Hh.rel_err(g["d_crf_table"], tab, 1e3 * Hh.grad_floor(tab))[0] <= 2e-4
    assert float(g["d_exposure"]) == pytest.approx(dexp, rel=2e-4, abs=1e-3)
    assert Hh.rel_err(g["d_crf_table"], r["dL_dcrf_table"], 1e3 * Hh.grad_floor(tab))[0] <= 2e-2
    g2, r2, bounded, rep2 = _masked_pass(oracle, sc, m, r["fwd"], "c4", cameras=cams, workers=workers, max_tol=2e-2)
    _parity_row("c4 (c3's cloud, 8 poses per frame" + (", free rotating poses" if free else "") + "): HIP vs fp32 C oracle, rows off the differing pixels",
                sc, m, g, r, {"R": int(R), "report": {k: v for k, v in rep.items()}}, masked=(g2, r2, bounded))


def test_no_kernel_writes_outside_its_buffers():
    """HS_GUARD=1 (rasterizer.py): guard zones of 4 KB around every buffer the library writes, over frames of every form --
    three tile sorts, N poses in both blur domains, free cameras, wild clouds, an overflowing capacity, the sweep's
    configurations, BASELINE c3 and c4 -- and not one guard byte changed.  (No address sanitizer exists for the GPU on this
    pool; a kernel that overruns is harmless next to torch's rounded eager allocations and fatal inside a captured graph's
    packed memory pool.)"""
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "guard_run.py")
    r = subprocess.run([sys.executable, script, "full"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "GUARD-OK" in r.stdout, r.stdout[-3000:]


def test_debug_flag_gives_identical_results():
    """settings.debug=True (HS_FLAG_DEBUG: the library waits for every stage and names a failing one) changes nothing
    but the synchronisation."""
    sc = S.make_scene(3000, 160, 120, 2, seed=41, hdr=True)
    a = Hh.run_hip(sc, hdr=True)
    from casualhdrsplat_amd import GaussianRasterizer
    rs, expo, crf = Hh.settings_from_scene(sc, "cuda", hdr=True, requires_grad=True)
    rs = rs._replace(debug=True)
    leaf = {k: getattr(sc, k).cuda().requires_grad_(True) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
    out = GaussianRasterizer(rs)(leaf["means3D"], torch.zeros(3000, 3, device="cuda"), leaf["opacities"], shs=leaf["shs"],
                                 scales=leaf["scales"], rotations=leaf["rotations"])
    (out[0] * sc.dL_dimage.cuda()).sum().backward()
    assert np.array_equal(out[0].detach().cpu().numpy(), a["color"])
    for k in leaf:
        assert np.array_equal(leaf[k].grad.cpu().numpy(), a["d_" + k]), k


def test_all_optional_outputs_together_are_linear_in_their_gradients():
    """(ldr, radii, hdr, alpha, invdepth) in one call: the backward with all four upstream gradients at once equals the
    sum of four backwards with one gradient each (the replay is linear in them), so no output's gradient is dropped or
    mis-ordered."""
    from casualhdrsplat_amd import GaussianRasterizer
    P, W, H = 2000, 144, 96
    sc = S.make_scene(P, W, H, 2, seed=33, hdr=True)
    rs, expo, crf = Hh.settings_from_scene(sc, "cuda", hdr=True, requires_grad=True)
    gen = torch.Generator().manual_seed(9)
    ups = [sc.dL_dimage.cuda(), torch.randn(3, H, W, generator=gen).cuda(), torch.randn(H, W, generator=gen).cuda(),
           torch.randn(H, W, generator=gen).cuda() * 3]
    names = ("means3D", "opacities", "shs", "scales", "rotations")

    def run(weights):
        leaf = {k: getattr(sc, k).cuda().requires_grad_(True) for k in names}
        for t in (expo, crf):
            t.grad = None
        out = GaussianRasterizer(rs, return_alpha=True, return_invdepth=True)(
            leaf["means3D"], torch.zeros(P, 3, device="cuda"), leaf["opacities"], shs=leaf["shs"], scales=leaf["scales"],
            rotations=leaf["rotations"])
        assert len(out) == 5 and out[2].shape == (3, H, W) and out[3].shape == (H, W) and out[4].shape == (H, W)
        loss = sum(w * (o * u).sum() for w, o, u in zip(weights, (out[0], out[2], out[3], out[4]), ups) if w)
        loss.backward()
        return {k: v.grad.double() for k, v in leaf.items()} | {"crf": crf.grad.double().clone(), "exp": expo.grad.double().clone()}

    total = run([1, 1, 1, 1])
    parts = [run([1 if i == j else 0 for j in range(4)]) for i in range(4)]
    for k in total:
        s = sum(p[k] for p in parts)
        assert torch.allclose(total[k], s, rtol=2e-4, atol=2e-5 * float(s.abs().max())), k
    assert all(float(p["means3D"].abs().max()) > 0 for p in parts)


def test_rccl_entry_points_of_the_exchange_in_a_single_rank_group():
    """The 1-hop forms of the gradient exchange call RCCL-only entry points (list all-to-all = grouped sends) that the
    gloo tests cannot reach.  A one-rank "nccl" process group on this GPU runs exactly those calls -- argument shapes,
    in-place aliasing of the output views, odd lengths -- so a multi-GPU node does not meet them for the first time."""
    import subprocess
    import sys
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["HS_ROOT"])
from casualhdrsplat_amd.distributed import all_reduce_direct, _gather_rows, all_reduce_gradients, init_from_env
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
for n in (1, 7, 4096, 1000003):
    x = torch.randn(n, device="cuda"); y = x.clone()
    all_reduce_direct(x)
    assert torch.equal(x, y), n
vc = torch.randn(2, 1001, 3, device="cuda")
out = torch.empty(2, 1001, 3, device="cuda")
_gather_rows(out, vc, None, direct=True)
assert torch.equal(out, vc)
w = _gather_rows(out.zero_(), vc, None, direct=True, async_op=True); w.wait()
assert torch.equal(out, vc)
p = torch.zeros(5, 3, device="cuda", requires_grad=True); p.grad = torch.ones(5, 3, device="cuda")
assert all_reduce_gradients([p], algo="direct") == 0          # one rank: nothing to exchange
dist.barrier(); dist.destroy_process_group()
print("RCCL-ENTRY-POINTS-OK")
'''
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HS_ROOT=root, MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "RCCL-ENTRY-POINTS-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_render_stats_counters_are_consistent():
    """hs_render_stats (the diagnostic instantiation bench.py's VALU roofline relies on): the forward and the backward
    count the same active (pixel, entry) pairs -- the sum of n_contrib-bounded contributions -- the backward's lane
    groups walk only entries the forward marked as taken by them (no empty trip, never more trips than the forward: a
    trip serves up to two entries, one per group), and the per-workgroup timeline covers every tile once."""
    from casualhdrsplat_amd import GaussianRasterizer
    from casualhdrsplat_amd.rasterizer import render_stats
    sc = S.make_scene(30000, 400, 232, 1, seed=14)
    rs, _, _ = Hh.settings_from_scene(sc, "cuda")
    leaf = {k: getattr(sc, k).cuda().requires_grad_(True) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
    m2 = torch.zeros(30000, 3, device="cuda", requires_grad=True)
    out = GaussianRasterizer(rs)(leaf["means3D"], m2, leaf["opacities"], shs=leaf["shs"], scales=leaf["scales"],
                                 rotations=leaf["rotations"])
    st = render_stats(out[0], sc.dL_dimage.cuda(), timeline=True)
    assert st["bwd_active_pixels"] == st["fwd_active_pixels"] > 0
    assert st["bwd_empty_trips"] == 0 and 0 < st["bwd_trips"] <= st["fwd_trips"]
    assert sum(st[k] for k in st if k.startswith("bwd_hist_")) == st["bwd_trips"]
    assert st["bwd_staged"] <= st["fwd_staged"]   # (batches: 64 entries in the backward, 128 in the forward)
    tl = st["bwd_timeline"].numpy()
    assert tl.shape == (25 * 15, 3) and (tl[:, 1] >= tl[:, 0]).all() and set((tl[:, 2] >> 32) & 0xF) <= set(range(8))
    # the counting kernels leave the results of the real ones untouched: a backward afterwards matches a fresh run
    (out[0] * sc.dL_dimage.cuda()).sum().backward()
    ref = Hh.run_hip(sc)
    assert np.array_equal(leaf["means3D"].grad.cpu().numpy(), ref["d_means3D"])
