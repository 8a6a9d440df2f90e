"""Development aid (GPU box): ONE case of the randomized sweep taken apart -- per gradient tensor, the worst elements of
HIP vs the fp32 C oracle, and (single-pose linear cases) both against float64 autograd.
usage: python scripts/sweep_case.py SWEEP_SEED CASE_INDEX"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as Hh
from oracle import c_oracle as O

O.build()
sweep_seed, index = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(sweep_seed)
for k in range(index + 1):
    c = Hh.sweep_case(rng, k)
print(c["what"])
sc, cams, hdr, dom, act = c["sc"], c["cams"], c["hdr"], c["dom"], c["act"]
if hdr or c["n_poses"] > 1:
    r = Hh.run_oracle_hdr(O, sc, cams, dom, radiance_activation=act)
    g = Hh.run_hip(sc, cameras=cams, hdr=True, blur_domain=dom, radiance_activation=act)
    ref, fwds = r, r["fwd"]
else:
    f, ref = Hh.run_oracle(O, sc, radiance_activation=act)
    g = Hh.run_hip(sc, radiance_activation=act)
    fwds = [f]
vis = np.zeros(c["P"], bool)
for f in fwds:
    vis |= f["radii"] > 0
    print("R", f["R"], "visible", int((f["radii"] > 0).sum()), "max tiles/G", int(f["tiles_touched"].max()), "max radius", int(f["radii"].max()))
for gk, rk in Hh.GRAD_KEYS:
    a, b = np.asarray(g["d_" + gk], np.float64).reshape(np.asarray(ref[rk]).shape), np.asarray(ref[rk], np.float64)
    fl = Hh.grad_floor(b)
    e = np.abs(a - b) / np.maximum(np.abs(b), fl)
    l2 = np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
    rows = np.abs(a - b).reshape(a.shape[0], -1).max(axis=1)
    worst = np.argsort(rows)[-3:][::-1]
    print(f"{gk:10s} max {e.max():.2e} frac {float((e > 1e-4).mean()):.2e} l2 {l2:.2e}  |ref| max {np.abs(b).max():.3e} rms {np.sqrt((b**2).mean()):.3e}")
    for w in worst:
        print(f"      row {w}: |diff| {rows[w]:.3e}  ref {np.abs(b[w]).max():.3e}  share of ||diff|| {rows[w] / max(np.linalg.norm(a - b), 1e-30):.2f}"
              f"  scale {sc.scales[w].numpy()} opac {float(sc.opacities[w]):.3f} radius {[int(f['radii'][w]) for f in fwds]} tiles {[int(f['tiles_touched'][w]) for f in fwds]}")
