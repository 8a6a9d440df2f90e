"""Development aid: the elements of a masked backward pass that need the largest constant in helpers.assert_grads_bounded.
usage (GPU box): python scripts/bound_outliers.py [P W H deg seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
from casualhdrsplat_amd import synthetic as S
from oracle import c_oracle as O
O.build()
a = [int(x) for x in sys.argv[1:6]] if len(sys.argv) > 5 else [100000, 800, 800, 0, 0]
P, W, Hh, deg, seed = a
sc = S.make_scene(P, W, Hh, deg, seed=seed)
g = H.run_hip(sc)
f, b = H.run_oracle(O, sc)
m = H.decision_masks(O, sc, [f], g["state"], what="probe")
g2, b2, dLm = H.masked_backward_pass(O, sc, m, [f], hdr=False)
print("excluded pixels", int(m["excluded"].sum()))
for gk, rk in H.GRAD_KEYS:
    r = np.asarray(b2[rk], np.float64); gg = np.asarray(g2["d_" + gk], np.float64).reshape(r.shape)
    Sx = np.asarray(b2["abs_" + rk], np.float64).reshape(r.shape)
    need = np.maximum(np.abs(gg - r) - 1e-4 * np.abs(r), 0) / np.maximum(2.0 ** -24 * Sx, 1e-300)
    flat = np.argsort(need.ravel())[::-1][:4]
    for i in flat:
        idx = np.unravel_index(i, r.shape); gi = idx[0]
        print(gk, idx, "need %.1f" % need[idx], "got %.6e ref %.6e S %.3e" % (gg[idx], r[idx], Sx[idx]), "n", int(b2["n_terms"][gi]),
              "radius", int(f["radii"][gi]), "opac %.4f" % float(sc.opacities[gi]), "scales", sc.scales[gi].numpy(),
              "conic_o", f["conic_opacity"][gi], "dconic", b2["dL_dconic"][gi], "abs", b2["abs_terms"][gi][:6])
