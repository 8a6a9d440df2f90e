"""Gradient-accuracy table: for each tensor, the error of the HIP path and of the fp32 C oracle
against the float64 PyTorch-autograd rasterizer on the same inputs -- max, 99.9th percentile and relative L2 of
|x - truth| / max(|truth|, 1e-3 RMS) -- on guard-banded scenes (no decision of any pixel within the guard band of a
threshold, so all three implementations take identical decisions) and on larger scenes restricted to the rows the
tests hold to the strict bar: every Gaussian NOT on the tile list of a pixel where HIP and oracle actually decided
differently (helpers.decision_masks; `strict_share` = their share of all rows).  The rows of the two FULL-SIZE frames
(BASELINE c3 and c4: HIP against the fp32 C oracle only -- float64 autograd of a million Gaussians is out of reach) are
written by the tests themselves (HS_PARITY_JSON=gpurun_out/r05_parity_fullsize.json pytest -k full_size_vs_oracle) and
appended here.  Round 5: every case also carries the MASKED pass (helpers.masked_backward_pass: dL zeroed on the pixels
where a decision differed, on both sides) -- strict_share 1.0 by construction, `n_outside_bound` = elements beyond
1e-4 |ref| + C 2^-24 sum w|term| (must be 0), `c_needed` = the constant each tensor would have needed.
Runs on the MI355X box; writes gpurun_out/<tag>_parity_table.json (committed under profiles/).
usage: python scripts/parity_table.py [--big]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import helpers as Hh
from casualhdrsplat_amd import synthetic as S
from oracle import c_oracle as O
from test_oracle_cross import torch_run

O.build()
TAG = os.environ.get("HS_ROUND_TAG", "r05")
KEYS = [("means3D", "dL_dmeans3D"), ("means2D", "dL_dmeans2D"), ("opacities", "dL_dopacity"), ("shs", "dL_dshs"),
        ("scales", "dL_dscales"), ("rotations", "dL_drots")]


def stats(x, truth, rows=None):
    x, truth = np.asarray(x, np.float64).reshape(truth.shape), np.asarray(truth, np.float64)
    if rows is not None:
        x, truth = x[rows], truth[rows]
    floor = max(1e-3 * float(np.sqrt((truth ** 2).mean())), 1e-30)
    e = np.abs(x - truth) / np.maximum(np.abs(truth), floor)
    return {"max": float(e.max()), "p999": float(np.percentile(e, 99.9)), "frac_gt_1e-4": float((e > 1e-4).mean()),
            "l2": float(np.linalg.norm(x - truth) / max(np.linalg.norm(truth), 1e-30))}


def guarded_seed(P, W, H, deg, start=0, tries=400):
    for seed in range(start, start + tries):
        sc = S.make_scene(P, W, H, deg, seed=seed)
        f, _ = Hh.run_oracle(O, sc, backward=False)
        if O.threshold_risk(Hh.oracle_camera(O, sc), f)["n_risky_pixels"] == 0:
            return seed
    return None


out = {"note": __doc__.split("usage")[0].strip(), "cases": []}
cases = [("c1 (1k Gaussians, 128x128, SH3) guard-banded", 1000, 128, 128, 3, True),
         ("600 Gaussians, 96x96, SH1, guard-banded", 600, 96, 96, 1, True),
         ("5k Gaussians, 200x136, SH2, rows off the differing pixels", 5000, 200, 136, 2, False)]
if "--big" in sys.argv:
    cases.append(("c2 (100k Gaussians, 800x800, SH0), rows off the differing pixels", 100000, 800, 800, 0, False))
for name, P, W, H, deg, guard in cases:
    seed = guarded_seed(P, W, H, deg) if guard else 0
    if seed is None:
        print("no guard-banded seed for", name); continue
    sc = S.make_scene(P, W, H, deg, seed=seed)
    t0 = time.time()
    f, b = Hh.run_oracle(O, sc)
    risk = O.threshold_risk(Hh.oracle_camera(O, sc), f)
    g = Hh.run_hip(sc)
    color64, st64, g64 = torch_run(sc, torch.float64)
    same_decisions = int((st64["n_contrib"].numpy() != f["n_contrib"]).sum())
    flips_hip = int((g["state"]["n_contrib"][0].astype(np.int64) != f["n_contrib"].astype(np.int64)).sum())
    m = Hh.decision_masks(O, sc, [f], g["state"], what=name)
    rows = ~m["rows"]
    case = {"case": name, "seed": seed, "guard_band_pixels": risk["n_risky_pixels"], "differing_pixels": m["n_differ"],
            "gaussians_compared": int(rows.sum()), "strict_share": float(rows.mean()), "P": P,
            "n_contrib_mismatch_fp64_vs_c": same_decisions, "n_contrib_mismatch_hip_vs_c": flips_hip, "tensors": {},
            "seconds": None}
    for k, ok in KEYS:
        truth = g64[k].reshape(b[ok].shape)
        case["tensors"][k] = {"hip_vs_fp64": stats(g["d_" + k], truth, rows), "c_fp32_vs_fp64": stats(b[ok], truth, rows),
                              "hip_vs_c_fp32": stats(g["d_" + k], b[ok].astype(np.float64), rows)}
    g2, b2, _ = Hh.masked_backward_pass(O, sc, m, [f], hdr=False)
    Hh.assert_grads_close(g2, b2, what=name + " masked")
    bounded = Hh.assert_grads_bounded(g2, b2, what=name + " masked")
    case["masked_pass"] = {"excluded_pixels": int(m["excluded"].sum()), "strict_share": 1.0, "c_bound": Hh.C_BOUND,
                           "n_outside_bound": int(sum(v[0] for v in bounded.values())),
                           "c_needed": {k: v[1] for k, v in bounded.items()},
                           "tensors": {k: stats(g2["d_" + k], b2[ok].astype(np.float64)) for k, ok in KEYS}}
    case["seconds"] = round(time.time() - t0, 1)
    out["cases"].append(case)
    print(name, "seed", seed, "risky px", risk["n_risky_pixels"], "compared", int(rows.sum()), "flips hip", flips_hip, "fp64", same_decisions)
    for k in case["tensors"]:
        t = case["tensors"][k]
        print(f"  {k:10s} hip/fp64 max {t['hip_vs_fp64']['max']:.2e} p999 {t['hip_vs_fp64']['p999']:.2e} l2 {t['hip_vs_fp64']['l2']:.2e} frac {t['hip_vs_fp64']['frac_gt_1e-4']:.1e} |"
              f" C/fp64 max {t['c_fp32_vs_fp64']['max']:.2e} p999 {t['c_fp32_vs_fp64']['p999']:.2e} l2 {t['c_fp32_vs_fp64']['l2']:.2e} frac {t['c_fp32_vs_fp64']['frac_gt_1e-4']:.1e} |"
              f" hip/C max {t['hip_vs_c_fp32']['max']:.2e} p999 {t['hip_vs_c_fp32']['p999']:.2e} frac {t['hip_vs_c_fp32']['frac_gt_1e-4']:.1e}")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
full = os.path.join(ROOT, "gpurun_out", TAG + "_parity_fullsize.json")
if os.path.exists(full):
    out["cases"] += json.load(open(full))["cases"]
json.dump(out, open(os.path.join(ROOT, "gpurun_out", TAG + "_parity_table.json"), "w"), indent=1)
