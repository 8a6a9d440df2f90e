"""Generates the committed golden fixtures (tests/golden/*.npz) with the CPU oracle.

The reference ships no golden vectors (parity unpinned by the reference), so these
are produced by this repo's own oracle (oracle/hs_oracle.c through oracle/c_oracle.py) after it has been
pinned by tests/test_oracle_known_answers.py and tests/test_oracle_cross.py.  A fixture is data only:
inputs, every integer intermediate, images and gradients for a fixed dL/dimage.

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import helpers as Hh  # noqa: E402
from casualhdrsplat_amd import synthetic as S  # noqa: E402
from oracle import c_oracle as O  # noqa: E402

# Every fixture is GUARD-BANDED: the seed is the first one >= `seed` for which no (pixel, entry)
# decision of any pose lies within the guard band of a threshold (alpha = 1/255 within 2e-5 relative, T = 1e-4 within
# 1e-4 relative, power = 0 within 1e-6) -- oracle.threshold_risk -- so any correct fp32 implementation takes exactly the
# decisions stored here and the GPU tests can demand zero flips and the strict gradient bar on every Gaussian.
CASES = [
    # name, P, W, H, deg, first seed to try, hdr, n_poses, blur_domain, radiance_activation
    ("ldr_deg0", 600, 96, 96, 0, 0, False, 1, "ldr", "relu_shift"),
    ("ldr_deg3", 600, 96, 80, 3, 1, False, 1, "ldr", "relu_shift"),
    ("hdr_deg3", 600, 96, 96, 3, 2, True, 1, "ldr", "relu_shift"),
    ("hdr_deg1_n4_ldrblur", 400, 80, 64, 1, 3, True, 4, "ldr", "relu_shift"),
    ("hdr_deg1_n4_hdrblur", 400, 80, 64, 1, 3, True, 4, "hdr", "relu_shift"),
    # linear radiance through a positive activation of the SH sum
    ("hdr_deg2_radiance_exp", 500, 88, 80, 2, 6, True, 1, "ldr", "exp"),
    ("ldr_deg1_radiance_softplus", 500, 80, 88, 1, 7, False, 1, "ldr", "softplus"),
]
# SURVEY.md 8(c) "golden fixtures to commit": BASELINE config c1 scale (1k Gaussians, 128x128), SH degree 0 and 3, LDR and
# HDR + CRF, N in {1, 8}, first guard-banded seed >= 0..3; two with a non-zero background (the background term of the
# backward).  The HDR ones are guard-banded in the CRF interval as well (no pixel of any image the CRF is applied to
# within 8 ulp of a knot, helpers.crf_interval_risk), so the GPU test holds EVERY gradient row to the strict bar.  The
# 8-pose LDR-domain case tone-maps eight images: it uses a 32-knot table (the knot guard band scales with K).
# The two 8-pose frames cannot be reject-sampled for the alpha / T thresholds (one c1 pose is clean for about one seed in
# twelve, eight at once for one in 12^8): they are stored as they come (guard_banded = 0) and the GPU test applies the
# live-oracle discipline to them -- decisions may differ only inside the oracle's guard band, every row off the pixels
# where they did is strict -- against the committed outputs.
C1_CASES = [
    # name, deg, first seed, hdr, n_poses, blur_domain, bg, crf_K
    ("c1_ldr_deg0_bg", 0, 0, False, 1, "ldr", (0.35, 0.2, 0.6), None),
    ("c1_ldr_deg3", 3, 1, False, 1, "ldr", None, None),
    ("c1_hdr_deg3_bg", 3, 2, True, 1, "ldr", (0.1, 0.25, 0.05), None),
    ("c1_hdr_deg0_n8_hdrblur", 0, 3, True, 8, "hdr", None, None),
    ("c1_hdr_deg3_n8_ldrblur", 3, 0, True, 8, "ldr", None, 32),
]
# Free 6-DoF cameras (round 6; /root/reference/assets/pipeline.png draws a free camera trajectory): the cloud in front of
# synthetic.random_camera(cam_seed) -- a general SO(3) world-to-view rotation plus a translation, every entry of the
# view matrix populated -- and, for the 4-pose frame, poses that ROTATE about it (synthetic.perturbed_poses).
FREE_CASES = [
    # name, P, W, H, deg, first seed, hdr, n_poses, blur_domain, cam_seed
    ("ldr_deg3_free_camera", 600, 96, 80, 3, 10, False, 1, "ldr", 4),
    ("hdr_deg2_n4_free_rotating_poses", 400, 80, 64, 2, 11, True, 4, "ldr", 5),
]
# antialiasing opacity compensation + expected inverse-depth output with its own upstream gradient
EXTRA_CASES = [("ldr_deg2_antialias_invdepth", 500, 88, 72, 2, 5)]


def free_poses(base, n_poses):
    """The poses of a FREE_CASES frame: rotating and shifting about the free base camera."""
    return S.perturbed_poses(base, n_poses, seed=1, rot_step_deg=0.25, step=0.02)


def guarded(P, W, H, deg, seed, hdr, n_poses, act="relu_shift", antialias=False, bg=None, crf_K=None, knot_dom=None,
            tries=5000, decisions=True, cam_seed=None):
    """First scene with seed >= `seed` none of whose poses has a (pixel, entry) decision inside the threshold guard band;
    knot_dom ('ldr' / 'hdr'): ... and no pixel of the image(s) the CRF is applied to within 8 ulp of a CRF knot."""
    import torch
    for s_ in range(seed, seed + tries):
        base = None if cam_seed is None else S.random_camera(W, H, cam_seed)
        sc = S.make_scene(P, W, H, deg, seed=s_, hdr=hdr, place_in=base)
        if bg is not None:
            sc.bg = torch.tensor(bg, dtype=torch.float32)
        if crf_K is not None:
            sc.crf_table = S.sigmoid_crf_table(crf_K, sc.crf_range)
        cams = S.blur_poses(W, H, n_poses, step=0.02) if n_poses > 1 else [sc.camera]
        if base is not None and n_poses > 1:
            cams = free_poses(base, n_poses)
        clean, imgs = True, []
        for cam in cams:
            ocam = Hh.oracle_camera(O, sc, cam, act)
            ocam.antialias = antialias
            f = O.forward(ocam, sc.means3D.numpy(), sc.opacities.numpy(), shs=sc.shs.numpy(), scales=sc.scales.numpy(),
                          rotations=sc.rotations.numpy())
            if decisions and O.threshold_risk(ocam, f, 2e-5, 1e-4)["n_risky_pixels"]:
                clean = False
                break
            imgs.append(f["color"])
        if clean and knot_dom is not None:
            if knot_dom == "hdr":
                imgs = [np.mean(np.stack(imgs), axis=0, dtype=np.float64).astype(np.float32)]
            clean = not any(Hh.crf_interval_risk(sc, h, h, ulps=8).any() for h in imgs)
        if clean:
            return sc, cams, s_
    raise RuntimeError("no guard-banded seed")


def scene_inputs(sc, cams):
    d = dict(means3D=sc.means3D.numpy(), scales=sc.scales.numpy(), rotations=sc.rotations.numpy(),
             opacities=sc.opacities.numpy(), shs=sc.shs.numpy(), bg=sc.bg.numpy(), dL_dimage=sc.dL_dimage.numpy(),
             viewmatrices=np.stack([c.viewmatrix.numpy() for c in cams]),
             projmatrices=np.stack([c.projmatrix.numpy() for c in cams]),
             camposes=np.stack([c.campos.numpy() for c in cams]),
             tanfov=np.array([sc.camera.tanfovx, sc.camera.tanfovy], np.float64))
    if sc.crf_table is not None:
        d.update(exposure=np.array(float(sc.exposure), np.float32), crf_table=sc.crf_table.numpy(),
                 crf_range=np.array(sc.crf_range, np.float64))
    return d


def make(name, P, W, H, deg, seed, hdr, n_poses, dom, act, bg=None, crf_K=None, knot_guard=False, cam_seed=None,
         decisions=None):
    if decisions is None:
        decisions = not (knot_guard and n_poses > 1)   # the 8-pose c1 frames: see C1_CASES
    sc, cams, seed = guarded(P, W, H, deg, seed, hdr, n_poses, act, bg=bg, crf_K=crf_K,
                             knot_dom=dom if (knot_guard and hdr) else None, decisions=decisions, cam_seed=cam_seed)
    out = scene_inputs(sc, cams)
    if cam_seed is not None:
        out["cam_seed"] = np.array(cam_seed, np.int64)
    if knot_guard and hdr:
        out["crf_knot_guarded"] = np.array(1, np.int64)
    out["meta"] = np.array([P, W, H, deg, seed, int(hdr), n_poses, int(dom == "hdr")], np.int64)
    out["radiance_activation"] = np.array(act)
    out["guard_banded"] = np.array(int(decisions), np.int64)
    if not hdr:
        f, b = Hh.run_oracle(O, sc, radiance_activation=act)
        for k in ("depths", "xy", "conic_opacity", "rgb", "radii", "tiles_touched", "offsets", "keys_sorted",
                  "point_list", "ranges", "color", "final_T", "n_contrib"):
            out["o_" + k] = f[k]
        for _, k in Hh.GRAD_KEYS:
            out["o_" + k] = b[k]
    else:
        r = Hh.run_oracle_hdr(O, sc, cams, dom, radiance_activation=act)
        out["o_color"], out["o_hdr"] = r["ldr"], r["hdr"]
        out["o_point_list"] = np.concatenate([f["point_list"] + k * P for k, f in enumerate(r["fwd"])])
        base, rr = 0, []
        for f in r["fwd"]:  # the HIP path sorts all poses as one list: pose k's ranges are offset by sum R_<k
            g = f["ranges"].astype(np.int64)
            g[g[:, 1] > g[:, 0]] += base
            rr.append(g)
            base += f["R"]
        out["o_ranges"] = np.concatenate(rr).astype(np.uint32)
        out["o_num_rendered"] = np.array([f["R"] for f in r["fwd"]], np.int64)
        out["o_n_contrib"] = np.stack([f["n_contrib"] for f in r["fwd"]])
        for _, k in Hh.GRAD_KEYS:
            out["o_" + k] = r[k]
        out["o_dL_dcrf_table"], out["o_dL_dexposure"] = r["dL_dcrf_table"], np.array(r["dL_dexposure"], np.float64)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "seed", seed, {k: v.shape for k, v in out.items() if k.startswith("o_")})


def make_extra(name, P, W, H, deg, seed):
    import torch
    sc, _, seed = guarded(P, W, H, deg, seed, False, 1, antialias=True)
    out = scene_inputs(sc, [sc.camera])
    out["meta"] = np.array([P, W, H, deg, seed, 0, 1, 0], np.int64)
    out["antialias"] = np.array(1, np.int64)
    out["guard_banded"] = np.array(1, np.int64)
    gD = (torch.randn(H, W, generator=torch.Generator().manual_seed(seed + 100)) * 4).numpy()
    out["dL_dinvdepth"] = gD
    ocam = Hh.oracle_camera(O, sc)
    ocam.antialias = True
    kw = dict(shs=sc.shs.numpy(), scales=sc.scales.numpy(), rotations=sc.rotations.numpy())
    f = O.forward(ocam, sc.means3D.numpy(), sc.opacities.numpy(), **kw)
    b = O.backward(ocam, f, sc.dL_dimage.numpy(), sc.means3D.numpy(), dL_dinvdepth_img=gD, **kw)
    for k in ("depths", "xy", "conic_opacity", "rgb", "radii", "tiles_touched", "offsets", "keys_sorted", "point_list",
              "ranges", "color", "final_T", "n_contrib", "invdepth"):
        out["o_" + k] = f[k]
    for _, k in Hh.GRAD_KEYS:
        out["o_" + k] = b[k]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "seed", seed, {k: v.shape for k, v in out.items() if k.startswith("o_")})


if __name__ == "__main__":
    import glob
    O.build()
    only = sys.argv[1:]   # optional: names to (re)generate; default all
    if not only:
        for old in glob.glob(os.path.join(HERE, "*.npz")):
            os.remove(old)
    for case in EXTRA_CASES:
        if not only or case[0] in only:
            make_extra(*case)
    for c in CASES:
        if not only or c[0] in only:
            make(*c)
    for name, P, W, H, deg, seed, hdr, n_poses, dom, cam_seed in FREE_CASES:
        if not only or name in only:
            make(name, P, W, H, deg, seed, hdr, n_poses, dom, "relu_shift", knot_guard=hdr, cam_seed=cam_seed, decisions=True)
    for name, deg, seed, hdr, n_poses, dom, bg, crf_K in C1_CASES:
        if not only or name in only:
            make(name, 1000, 128, 128, deg, seed, hdr, n_poses, dom, "relu_shift", bg=bg, crf_K=crf_K, knot_guard=True)
