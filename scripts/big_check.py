"""Large-scene sanity (development aid): 4M Gaussians at 4K, oracle-free invariants + timing."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as Hh
from casualhdrsplat_amd import GaussianRasterizer, inspect_state, synthetic as S
P, W, H = 4_000_000, 3840, 2160
sc = S.make_scene(P, W, H, 3, seed=1, hdr=True)
dev = "cuda"
rs, expo, crf = Hh.settings_from_scene(sc, dev, hdr=True, requires_grad=True)
leaves = [t.to(dev).requires_grad_(True) for t in (sc.means3D, torch.zeros(P, 3), sc.opacities, sc.shs, sc.scales, sc.rotations)]
out = GaussianRasterizer(rs)(leaves[0], leaves[1], leaves[2], shs=leaves[3], scales=leaves[4], rotations=leaves[5])
st = inspect_state(out[0])
R = st["num_rendered"]
keys = st["keys_sorted"][:R]
assert bool((keys[1:] >= keys[:-1]).all()) and R == int(st["tiles_touched"].to(torch.int64).sum())
rng = st["ranges"].to(torch.int64); assert int((rng[:, 1] - rng[:, 0]).sum()) == R
torch.autograd.backward(out[0], grad_tensors=sc.dL_dimage.to(dev))
assert all(bool(torch.isfinite(t.grad).all()) for t in leaves) and bool(torch.isfinite(crf.grad).all())
rast = GaussianRasterizer(rs, capacity=int(R * 1.1))
def step():
    for t in leaves: t.grad = None
    o = rast(leaves[0], leaves[1], leaves[2], shs=leaves[3], scales=leaves[4], rotations=leaves[5])
    torch.autograd.backward(o[0], grad_tensors=sc.dL_dimage.to(dev))
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f"P={P} {W}x{H} R={R}: {dt*1e3:.2f} ms/step, {W*H/dt/1e6:.0f} Mpix/s, peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
