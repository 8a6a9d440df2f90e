"""Development aid (GPU box): one configuration of the randomized sweep taken apart -- HIP (HDR path), HIP fed the
oracle's dL/dH directly (no CRF stage), the fp32 C oracle and float64 autograd, per gradient tensor.
usage: python scripts/soak_case.py P W H deg seed"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as Hh
from casualhdrsplat_amd import synthetic as S
from oracle import c_oracle as O
from test_oracle_cross import torch_run

P, W, H, deg, seed = [int(v) for v in sys.argv[1:6]]
sc = S.make_scene(P, W, H, deg, seed=seed, hdr=True)
r = Hh.run_oracle_hdr(O, sc, None, "ldr")
f = r["fwd"][0]
dt = float(sc.exposure); tab = sc.crf_table.numpy(); umin, umax = sc.crf_range
dH, _, _ = O.tonemap_bwd(f["color"], dt, tab, umin, umax, sc.dL_dimage.numpy())
dH = dH.astype(np.float32)
g_hdr = Hh.run_hip(sc, hdr=True)
# HIP with the linear-radiance path and the oracle's dL/dH as upstream gradient
sc2 = S.make_scene(P, W, H, deg, seed=seed, hdr=True)
sc2.dL_dimage = torch.from_numpy(dH)
g_lin = Hh.run_hip(sc2)
_, st64, g64 = torch_run(sc, torch.float64, dL=torch.from_numpy(dH))
print("n_contrib flips hip/oracle:", int((g_hdr["state"]["n_contrib"][0].astype(np.uint32) != f["n_contrib"]).sum()),
      " fp64/oracle:", int((st64["n_contrib"].numpy() != f["n_contrib"]).sum()))
m = Hh.decision_masks(O, sc, [f], g_hdr["state"], crf_got=[g_hdr["hdr"]], crf_ref=[f["color"]])
print("guard-band pixels", int(m["pix_risk"].sum()), "differing pixels", m["n_differ"], "CRF-knot pixels", m["n_knot_pixels"],
      "rows off the strict bar", int(m["rows"].sum()))
for k, ok in Hh.GRAD_KEYS:
    truth = g64[k].reshape(r[ok].shape).astype(np.float64)
    def err(x, ref=truth):
        x = np.asarray(x, np.float64).reshape(ref.shape)
        fl = Hh.grad_floor(ref)
        e = np.abs(x - ref) / np.maximum(np.abs(ref), fl)
        return "max %.2e p99.9 %.2e l2 %.2e" % (e.max(), np.percentile(e, 99.9), np.linalg.norm(x - ref) / np.linalg.norm(ref))
    print(f"{k:10s} oracle-fp64 [{err(r[ok])}]  hipHDR-fp64 [{err(g_hdr['d_' + k])}]  hipLIN-fp64 [{err(g_lin['d_' + k])}]  hipHDR-oracle [{err(g_hdr['d_' + k], np.asarray(r[ok], np.float64))}]")
# where do the HDR-path gradients dL/dH differ?  (HIP does not expose dL/dH; compare the LDR images and knots instead)
u = np.log(np.maximum(f["color"] * dt, 1e-30))
K = tab.shape[1]
x = (u - umin) / (umax - umin) * (K - 1)
fr = np.abs(x - np.round(x))
print("pixels within 1e-4 of a CRF knot:", int((fr < 1e-4).sum()), "within 1e-5:", int((fr < 1e-5).sum()), "of", fr.size)
