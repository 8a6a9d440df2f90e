"""Runs K fwd+bwd steps of a BASELINE config on the GPU (development/profiling aid)."""
import sys, os, time, argparse
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from casualhdrsplat_amd import synthetic as S, GaussianRasterizer, inspect_state
import helpers as Hh

ap = argparse.ArgumentParser()
ap.add_argument("--P", type=int, default=1000000); ap.add_argument("--W", type=int, default=1920); ap.add_argument("--H", type=int, default=1080)
ap.add_argument("--deg", type=int, default=3); ap.add_argument("--hdr", type=int, default=1); ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--poses", type=int, default=1); ap.add_argument("--capacity", type=int, default=0)
# counter passes that must see render_bwd_kernel WITHOUT the CRF gradient's tail workgroups: after the
# first step only forwards are run, each followed by a replay of HS_BWD_RENDER alone (no tail, nothing else of the backward)
ap.add_argument("--render-bwd-alone", action="store_true")
a = ap.parse_args()
dev = "cuda"
sc = S.make_scene(a.P, a.W, a.H, a.deg, seed=0, hdr=bool(a.hdr))
cams = S.blur_poses(a.W, a.H, a.poses) if a.poses > 1 else None
rs, exposure, crf = Hh.settings_from_scene(sc, dev, cams, bool(a.hdr), requires_grad=True)
means3D = sc.means3D.to(dev).requires_grad_(True); means2D = torch.zeros_like(means3D, requires_grad=True)
opac = sc.opacities.to(dev).requires_grad_(True); shs = sc.shs.to(dev).requires_grad_(True)
scales = sc.scales.to(dev).requires_grad_(True); rots = sc.rotations.to(dev).requires_grad_(True)
dL = sc.dL_dimage.to(dev)
rast = GaussianRasterizer(rs, capacity=(a.capacity or None))
def step():
    out = rast(means3D, means2D, opac, shs=shs, scales=scales, rotations=rots)
    (out[0] * dL).sum().backward()
    return out
out = step(); torch.cuda.synchronize()
st = inspect_state(out[0])
nc = st["n_contrib"].to(torch.int64); rng = st["ranges"].to(torch.int64)
H, W = a.H, a.W
gx, gy = (W + 15) // 16, (H + 15) // 16
pad = torch.zeros(a.poses, gy * 16, gx * 16, dtype=torch.int64, device=dev); pad[:, :H, :W] = nc
tmax = pad.reshape(a.poses, gy, 16, gx, 16).amax(dim=(2, 4))
print(f"R={st['num_rendered']}  R'={int(tmax.sum())}  E={int(nc.sum())}  tiles={gx*gy}  avg list={float((rng[:,1]-rng[:,0]).float().mean()):.1f} max list={int((rng[:,1]-rng[:,0]).max())}")
print("color mean", float(out[0].mean()), "finite", bool(torch.isfinite(out[0]).all()), "grad finite", bool(torch.isfinite(means3D.grad).all()))
if a.render_bwd_alone:
    from casualhdrsplat_amd import _lib as L
    from casualhdrsplat_amd.rasterizer import replay_backward
    def step():
        o = rast(means3D, means2D, opac, shs=shs, scales=scales, rotations=rots)
        replay_backward(o[0], dL, L.HS_BWD_RENDER)
        return o
torch.cuda.synchronize(); t = time.time()
for _ in range(a.steps): step()
torch.cuda.synchronize(); dt = (time.time() - t) / a.steps
print(f"step {dt*1e3:.3f} ms -> {1/dt:.1f} img/s, {W*H*a.poses/dt/1e6:.1f} Mpix/s")
