"""Times the BINNING stage alone (development aid; safe for ablated builds whose sorted list is wrong: nothing renders it).
One forward with the radix tile sort builds the frame; then HS_TILE_SORT=<form> and only HS_STAGE_BIN is replayed.
usage: python scripts/time_binning.py [--form hier|radix] [--cfg c3|c4|c2] [--iters 20]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
ap = argparse.ArgumentParser()
ap.add_argument("--form", default="hier"); ap.add_argument("--cfg", default="c3"); ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
os.environ["HS_TILE_SORT"] = "radix"
from casualhdrsplat_amd import synthetic as S, GaussianRasterizer, _lib as L
from casualhdrsplat_amd.rasterizer import replay_forward
import helpers as Hh
P, W, H, deg, hdr, poses = {"c2": (100000, 800, 800, 0, False, 1), "c3": (1000000, 1920, 1080, 3, True, 1),
                            "c4": (1000000, 1920, 1080, 3, True, 8)}[a.cfg]
sc = S.make_scene(P, W, H, deg, seed=0, hdr=hdr)
cams = S.blur_poses(W, H, poses) if poses > 1 else None
rs, _, _ = Hh.settings_from_scene(sc, "cuda", cams, hdr, requires_grad=True)
leaves = [t.cuda().requires_grad_(True) for t in (sc.means3D, torch.zeros_like(sc.means3D), sc.opacities)]
cap = {"c2": 900000, "c3": 8500000, "c4": 68000000}[a.cfg]
out = GaussianRasterizer(rs, capacity=cap)(leaves[0], leaves[1], leaves[2], shs=sc.shs.cuda().requires_grad_(True),
                                           scales=sc.scales.cuda().requires_grad_(True), rotations=sc.rotations.cuda().requires_grad_(True))
torch.cuda.synchronize()
os.environ["HS_TILE_SORT"] = a.form
for _ in range(3):
    replay_forward(out[0], L.HS_STAGE_BIN)
torch.cuda.synchronize()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.iters)]
for e0, e1 in evs:
    e0.record(); replay_forward(out[0], L.HS_STAGE_BIN); e1.record()
torch.cuda.synchronize()
ts = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
print(f"binning[{a.form}] {a.cfg} lib={os.path.basename(L.LIB_PATH)}: median {ts[len(ts)//2]*1e3:.1f} us  min {ts[0]*1e3:.1f} us")
