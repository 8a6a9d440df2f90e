"""Pins the CPU oracle (oracle/hs_oracle.c) with closed-form known-answer tests -- the list of
SURVEY.md 8(c).  The reference ships no tests or golden vectors (parity unpinned by the reference),
so these closed forms plus the fp64 autograd cross-check (test_oracle_cross.py) are what anchors it.
"""
import math

import numpy as np
import pytest

from casualhdrsplat_amd import synthetic as S

C0 = 0.28209479177387814


def cam_for(O, W, H, sh_degree=0, bg=(0.0, 0.0, 0.0), camera=None):
    c = camera or S.make_camera(W, H)
    return O.Camera(W, H, c.tanfovx, c.tanfovy, c.viewmatrix.numpy(), c.projmatrix.numpy(), c.campos.numpy(),
                    np.asarray(bg, np.float32), 1.0, sh_degree), c


def point_at_pixel(c, px, py, z):
    """3-D point that projects exactly onto pixel centre (px, py) at depth z for camera c."""
    ndc_x = (2 * px + 1) / c.W - 1
    ndc_y = (2 * py + 1) / c.H - 1
    return [ndc_x * c.tanfovx * z, ndc_y * c.tanfovy * z, z]


def iso(O, W, H, px, py, z, sigma_px, opacity, color, bg=(0, 0, 0)):
    oc, c = cam_for(O, W, H, 0, bg)
    fx = W / (2 * c.tanfovx)
    s = sigma_px * z / fx
    means = np.array([point_at_pixel(c, px, py, z)], np.float32)
    f = O.forward(oc, means, np.array([opacity], np.float32), colors_precomp=np.array([color], np.float32),
                  scales=np.full((1, 3), s, np.float32), rotations=np.array([[1, 0, 0, 0]], np.float32))
    return f, oc


def test_single_isotropic_gaussian_closed_form(oracle):
    W = H = 64
    sigma, o, col, bg = 3.0, 0.8, (0.2, 0.5, 0.9), (0.1, 0.1, 0.3)
    f, _ = iso(oracle, W, H, 32, 32, 5.0, sigma, o, col, bg)
    # centre pixel: alpha = min(0.99, o); colour = c*alpha + (1-alpha)*bg
    for ch in range(3):
        assert f["color"][ch, 32, 32] == pytest.approx(col[ch] * o + (1 - o) * bg[ch], rel=1e-5)
    # offset d: alpha = o * exp(-d^2 / (2 (sigma^2 + 0.3)))  (EWA: 2-D covariance dilated by 0.3 px^2)
    for d in (1, 2, 4, 5):
        a = o * math.exp(-d * d / (2 * (sigma * sigma + 0.3)))
        assert a >= 1 / 255
        assert f["color"][0, 32, 32 + d] == pytest.approx(col[0] * a + (1 - a) * bg[0], rel=2e-4)
        assert f["color"][1, 32 - d, 32] == pytest.approx(col[1] * a + (1 - a) * bg[1], rel=2e-4)
        assert f["final_T"][32, 32 + d] == pytest.approx(1 - a, rel=2e-4)
    assert f["n_contrib"][32, 32] == 1
    # radius = ceil(3 * sqrt(lambda_max)), lambda = sigma^2 + 0.3
    assert f["radii"][0] == math.ceil(3 * math.sqrt(sigma * sigma + 0.3))
    assert f["xy"][0, 0] == pytest.approx(32.0, abs=1e-3) and f["xy"][0, 1] == pytest.approx(32.0, abs=1e-3)
    assert f["depths"][0] == pytest.approx(5.0)


def test_alpha_clamped_at_099_and_skipped_below_1_255(oracle):
    f, _ = iso(oracle, 32, 32, 16, 16, 4.0, 2.0, 1.0, (1, 1, 1))
    assert f["color"][0, 16, 16] == pytest.approx(0.99, rel=1e-6)
    f, _ = iso(oracle, 32, 32, 16, 16, 4.0, 2.0, 0.003, (1, 1, 1))  # o < 1/255 everywhere
    assert np.all(f["color"] == 0) and np.all(f["n_contrib"] == 0) and np.all(f["final_T"] == 1)


def two_gaussians(O, z_a, z_b):
    W = H = 32
    oc, c = cam_for(O, W, H)
    fx = W / (2 * c.tanfovx)
    means = np.array([point_at_pixel(c, 16, 16, z_a), point_at_pixel(c, 16, 16, z_b)], np.float32)
    scales = np.array([[2.0 * z_a / fx] * 3, [2.0 * z_b / fx] * 3], np.float32)
    cols = np.array([[1, 0, 0], [0, 1, 0]], np.float32)
    return O.forward(oc, means, np.array([0.6, 0.5], np.float32), colors_precomp=cols, scales=scales,
                     rotations=np.array([[1, 0, 0, 0]] * 2, np.float32))


def test_two_gaussians_front_to_back_and_depth_swap(oracle):
    f = two_gaussians(oracle, 3.0, 6.0)  # red in front
    assert f["color"][0, 16, 16] == pytest.approx(0.6, rel=1e-5)
    assert f["color"][1, 16, 16] == pytest.approx(0.5 * (1 - 0.6), rel=1e-5)
    assert f["final_T"][16, 16] == pytest.approx(0.4 * 0.5, rel=1e-5)
    assert list(f["point_list"][f["ranges"][1 * 2 + 1, 0]:f["ranges"][1 * 2 + 1, 0] + 2]) == [0, 1]
    g = two_gaussians(oracle, 6.0, 3.0)  # green in front
    assert g["color"][1, 16, 16] == pytest.approx(0.5, rel=1e-5)
    assert g["color"][0, 16, 16] == pytest.approx(0.6 * 0.5, rel=1e-5)
    t = g["ranges"][3]
    assert list(g["point_list"][t[0]:t[0] + 2]) == [1, 0]
    assert g["n_contrib"][16, 16] == 2


def test_near_plane_cull(oracle):
    W = H = 32
    oc, c = cam_for(oracle, W, H)
    means = np.array([[0, 0, 0.2], [0, 0, 0.19], [0, 0, -1.0], [0, 0, 0.2001]], np.float32)
    f = oracle.forward(oc, means, np.full(4, 0.5, np.float32), colors_precomp=np.ones((4, 3), np.float32),
                       scales=np.full((4, 3), 0.001, np.float32), rotations=np.array([[1, 0, 0, 0]] * 4, np.float32))
    assert list(f["radii"][:3]) == [0, 0, 0] and f["radii"][3] > 0
    assert list(f["tiles_touched"][:3]) == [0, 0, 0]
    assert f["R"] == f["tiles_touched"][3]
    assert list(oracle.mark_visible(oc, means)) == [False, False, False, True]


def test_binning_invariants(oracle):
    sc = S.make_scene(3000, 200, 136, 0, seed=5)  # 136 = 8.5 tiles: half tile row like 1080p
    import helpers as Hh
    f, _ = Hh.run_oracle(oracle, sc, backward=False)
    R = f["R"]
    assert R == int(f["tiles_touched"].sum()) == len(f["keys_unsorted"])
    assert np.array_equal(np.cumsum(f["tiles_touched"], dtype=np.uint32), f["offsets"])
    ks = f["keys_sorted"]
    assert np.all(ks[1:] >= ks[:-1])
    # stable: equal keys keep ascending unsorted position -> for equal keys ascending Gaussian id
    eq = ks[1:] == ks[:-1]
    assert np.all(f["point_list"][1:][eq] > f["point_list"][:-1][eq])
    # same multiset
    assert np.array_equal(np.sort(f["keys_unsorted"]), ks)
    gx, gy = (200 + 15) // 16, (136 + 15) // 16
    rng = f["ranges"].astype(np.int64)
    lens = rng[:, 1] - rng[:, 0]
    assert lens.sum() == R
    nz = rng[lens > 0]
    assert nz[0, 0] == 0 and nz[-1, 1] == R and np.all(nz[1:, 0] == nz[:-1, 1])  # partition of [0, R)
    # every pair's tile lies inside its Gaussian's rect
    tiles = (ks >> np.uint64(32)).astype(np.int64)
    ty, tx = tiles // gx, tiles % gx
    rect = f["rect"][f["point_list"]]
    assert np.all((tx >= rect[:, 0]) & (tx < rect[:, 2]) & (ty >= rect[:, 1]) & (ty < rect[:, 3]))
    assert tiles.max() < gx * gy
    assert f["sort_bits"] == 32 + int(gx * gy).bit_length()


def test_rect_formula_hand_cases(oracle):
    # (pixel x, pixel y, sigma) -> rect, on a 1920x1080 frame: grid 120 x 68, last tile row half empty
    W, H = 1920, 1080
    oc, c = cam_for(oracle, W, H)
    fx = W / (2 * c.tanfovx)
    cases = [(8.0, 8.0, 1.0), (0.0, 0.0, 5.0), (1919.0, 1079.0, 4.0), (960.0, 1075.0, 2.0), (31.0, 16.0, 0.5)]
    means = np.array([point_at_pixel(c, x, y, 4.0) for x, y, _ in cases], np.float32)
    scales = np.array([[s * 4.0 / fx] * 3 for _, _, s in cases], np.float32)
    f = oracle.forward(oc, means, np.full(len(cases), 0.5, np.float32),
                       colors_precomp=np.ones((len(cases), 3), np.float32), scales=scales,
                       rotations=np.array([[1, 0, 0, 0]] * len(cases), np.float32))
    for i, (x, y, s) in enumerate(cases):
        # off-axis an isotropic 3-D Gaussian stretches radially: cov2D = s^2 (I + v v^T) + 0.3 I, v = t.xy / t.z;
        # the rule floors the discriminant at 0.1: lambda = mid + sqrt(max(0.1, mid^2 - det))
        ax, ay = means[i, 0] / means[i, 2], means[i, 1] / means[i, 2]
        a, b, cc = s * s * (1 + ax * ax) + 0.3, s * s * ax * ay, s * s * (1 + ay * ay) + 0.3
        mid, det = 0.5 * (a + cc), a * cc - b * b
        r = math.ceil(3 * math.sqrt(mid + math.sqrt(max(0.1, mid * mid - det))))
        assert f["radii"][i] == r
        px, py = f["xy"][i]
        want = [min(120, max(0, int((px - r) / 16))), min(68, max(0, int((py - r) / 16))),
                min(120, max(0, int((px + r + 15) / 16))), min(68, max(0, int((py + r + 15) / 16)))]
        assert list(f["rect"][i]) == want
    assert list(f["rect"][0]) == [0, 0, 1, 1]            # r=5 around (8,8): 8+5 < 16 stays in tile 0
    assert list(f["rect"][1][:2]) == [0, 0]              # negative (p - r)/16 truncates toward zero, clamps to 0
    assert list(f["rect"][2][2:]) == [120, 68]           # bottom-right corner clamps to the 120 x 68 grid
    assert f["rect"][3][3] == 68                         # the half-empty last tile row (1080 = 67.5 tiles) is a tile


def test_early_termination(oracle):
    # stacked alpha = 0.9 splats: T after k = 0.1^k ; test_T < 1e-4 first at k = 5 -> 4 contributors
    W = H = 32
    oc, c = cam_for(oracle, W, H)
    fx = W / (2 * c.tanfovx)
    n = 8
    means = np.array([point_at_pixel(c, 16, 16, 2.0 + k) for k in range(n)], np.float32)
    scales = np.array([[3.0 * (2.0 + k) / fx] * 3 for k in range(n)], np.float32)
    f = oracle.forward(oc, means, np.full(n, 0.9, np.float32), colors_precomp=np.ones((n, 3), np.float32),
                       scales=scales, rotations=np.array([[1, 0, 0, 0]] * n, np.float32))
    assert f["n_contrib"][16, 16] == 4
    assert f["final_T"][16, 16] == pytest.approx(1e-4, rel=1e-4)
    assert f["color"][0, 16, 16] == pytest.approx(1 - 1e-4, rel=1e-5)


def test_sh_degree0_and_clamp_mask(oracle):
    W = H = 32
    oc, c = cam_for(oracle, W, H, sh_degree=0)
    fx = W / (2 * c.tanfovx)
    means = np.array([point_at_pixel(c, 16, 16, 4.0)] * 2, np.float32)
    shs = np.array([[[1.0, -0.5, -3.0]], [[0.2, 0.2, 0.2]]], np.float32)
    f = oracle.forward(oc, means, np.full(2, 0.5, np.float32), shs=shs, scales=np.full((2, 3), 2 * 4.0 / fx, np.float32),
                       rotations=np.array([[1, 0, 0, 0]] * 2, np.float32))
    assert f["rgb"][0, 0] == pytest.approx(C0 * 1.0 + 0.5)
    assert f["rgb"][0, 1] == pytest.approx(C0 * -0.5 + 0.5)
    assert f["rgb"][0, 2] == 0.0 and f["clamped"][0, 2] == 1 and f["clamped"][0, 0] == 0
    dL = np.ones((3, H, W), np.float32)
    b = oracle.backward(oc, f, dL, means, shs=shs, scales=np.full((2, 3), 2 * 4.0 / fx, np.float32),
                        rotations=np.array([[1, 0, 0, 0]] * 2, np.float32))
    assert b["dL_dshs"][0, 0, 2] == 0.0 and b["dL_dshs"][0, 0, 0] != 0.0  # clamped channel passes no gradient


def test_sh_bands_against_independent_polynomials(oracle):
    """deg 1-3 basis against an independent evaluation (real SH in Cartesian form, scipy-free)."""
    rng = np.random.default_rng(0)
    W = H = 48
    oc, c = cam_for(oracle, W, H, sh_degree=3)
    fx = W / (2 * c.tanfovx)
    P = 6
    px, py, z = rng.uniform(8, 40, P), rng.uniform(8, 40, P), rng.uniform(2, 8, P)
    means = np.array([point_at_pixel(c, px[i], py[i], z[i]) for i in range(P)], np.float32)
    shs = rng.normal(0, 0.2, (P, 16, 3)).astype(np.float32)
    f = oracle.forward(oc, means, np.full(P, 0.5, np.float32), shs=shs,
                       scales=(2 * z / fx)[:, None].repeat(3, 1).astype(np.float32),
                       rotations=np.array([[1, 0, 0, 0]] * P, np.float32))
    d = means.astype(np.float64)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    x, y, zz = d[:, 0], d[:, 1], d[:, 2]
    pi = math.pi
    Y = np.stack([
        0.5 * math.sqrt(1 / pi) * np.ones(P),
        -math.sqrt(3 / (4 * pi)) * y, math.sqrt(3 / (4 * pi)) * zz, -math.sqrt(3 / (4 * pi)) * x,
        0.5 * math.sqrt(15 / pi) * x * y, -0.5 * math.sqrt(15 / pi) * y * zz,
        0.25 * math.sqrt(5 / pi) * (3 * zz * zz - 1), -0.5 * math.sqrt(15 / pi) * x * zz,
        0.25 * math.sqrt(15 / pi) * (x * x - y * y),
        -0.25 * math.sqrt(35 / (2 * pi)) * y * (3 * x * x - y * y), 0.5 * math.sqrt(105 / pi) * x * y * zz,
        -0.25 * math.sqrt(21 / (2 * pi)) * y * (5 * zz * zz - 1), 0.25 * math.sqrt(7 / pi) * zz * (5 * zz * zz - 3),
        -0.25 * math.sqrt(21 / (2 * pi)) * x * (5 * zz * zz - 1), 0.25 * math.sqrt(105 / pi) * zz * (x * x - y * y),
        -0.25 * math.sqrt(35 / (2 * pi)) * x * (x * x - 3 * y * y)], axis=1)
    want = np.maximum(np.einsum("pk,pkc->pc", Y, shs.astype(np.float64)) + 0.5, 0)
    assert np.allclose(f["rgb"], want, rtol=2e-5, atol=2e-6)


def test_transposed_matrix_convention_places_point_at_expected_pixel(oracle):
    """Camera translated and yawed: a known 3-D point must land on the analytically expected pixel."""
    W, H = 160, 96
    cam = S.yaw_camera(W, H, 7.0, centre_depth=5.0)
    oc, _ = cam_for(oracle, W, H, camera=cam)
    Xw = np.array([0.3, -0.2, 5.5])
    w2c = cam.viewmatrix.numpy().T.astype(np.float64)
    xv = w2c[:3, :3] @ Xw + w2c[:3, 3]
    fx, fy = W / (2 * cam.tanfovx), H / (2 * cam.tanfovy)
    px = fx * xv[0] / xv[2] + (W - 1) / 2
    py = fy * xv[1] / xv[2] + (H - 1) / 2
    f = oracle.forward(oc, Xw[None].astype(np.float32), np.array([0.9], np.float32),
                       colors_precomp=np.ones((1, 3), np.float32), scales=np.full((1, 3), 0.01, np.float32),
                       rotations=np.array([[1, 0, 0, 0]], np.float32))
    assert f["xy"][0, 0] == pytest.approx(px, abs=2e-3) and f["xy"][0, 1] == pytest.approx(py, abs=2e-3)
    assert f["depths"][0] == pytest.approx(xv[2], rel=1e-6)
    iy, ix = np.unravel_index(f["color"][0].argmax(), (H, W))
    assert (ix, iy) == (round(px), round(py))


def test_crf_identity_table_and_clamping(oracle):
    """table_c(u) = exp(u) sampled densely => LDR == H*dt inside the range; flat outside."""
    K, umin, umax = 4096, -6.0, 3.0
    u = np.linspace(umin, umax, K)
    table = np.stack([np.exp(u)] * 3).astype(np.float32)
    Hd = np.stack([np.linspace(0.01, 30, 500)] * 3).astype(np.float32)
    dt = 0.5
    ldr = oracle.tonemap_fwd(Hd, dt, table, umin, umax)
    inside = (Hd * dt > math.exp(umin)) & (Hd * dt < math.exp(umax))
    assert np.allclose(ldr[inside], (Hd * dt)[inside], rtol=2e-5)
    big = oracle.tonemap_fwd(np.full((3, 4), 1e4, np.float32), dt, table, umin, umax)
    assert np.allclose(big, math.exp(umax), rtol=1e-6)
    zero = oracle.tonemap_fwd(np.zeros((3, 4), np.float32), dt, table, umin, umax)
    assert np.allclose(zero, math.exp(umin), rtol=1e-6)
    # gradients: d LDR / d H = dt inside (identity response), 0 outside; table grads sum to sum of dL
    g = np.ones_like(Hd)
    dh, dtab, dexp = oracle.tonemap_bwd(Hd, dt, table, umin, umax, g)
    assert np.allclose(dh[inside], dt, rtol=2e-3)
    assert np.all(dh[~inside] == 0)
    assert dtab.sum() == pytest.approx(g.sum(), rel=1e-5)
    assert dexp == pytest.approx(float((Hd * inside).sum()), rel=2e-3)


def test_sort_is_stable_on_ties(oracle):
    import ctypes as C
    L = oracle.lib()
    keys = np.array([5, 3, 5, 3, 5, 1 << 40, 3], np.uint64)
    vals = np.arange(7, dtype=np.uint32)
    ko, vo = np.zeros_like(keys), np.zeros_like(vals)
    assert L.hso_sort_pairs(keys.ctypes.data_as(C.c_void_p), vals.ctypes.data_as(C.c_void_p), C.c_int64(7), C.c_int(45),
                            ko.ctypes.data_as(C.c_void_p), vo.ctypes.data_as(C.c_void_p)) == 0
    assert list(vo) == [1, 3, 6, 0, 2, 4, 5]
    assert list(ko) == sorted(keys.tolist())
    assert L.hso_key_tile_bits(C.c_uint32(8160)) == 13 and L.hso_key_tile_bits(C.c_uint32(2500)) == 12
    assert L.hso_key_tile_bits(C.c_uint32(64)) == 7
