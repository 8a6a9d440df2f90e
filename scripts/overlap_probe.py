"""Feasibility probe (development aid): how much of a memory-bound kernel hides under the binning chain when it runs on a
second stream?  Stand-in for a colour (SH) evaluation split off the forward's first kernel: a torch reduction that reads
the SH array (192 MB at c3) and writes 12 MB.  Prints binning alone, stand-in alone, serial sum, and both concurrently.
usage: python scripts/overlap_probe.py [--cfg c3|c4] [--iters 20]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
ap = argparse.ArgumentParser()
ap.add_argument("--cfg", default="c3"); ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
from casualhdrsplat_amd import synthetic as S, GaussianRasterizer, _lib as L
from casualhdrsplat_amd.rasterizer import replay_forward
import helpers as Hh
P, W, H, deg, hdr, poses = {"c3": (1000000, 1920, 1080, 3, True, 1), "c4": (1000000, 1920, 1080, 3, True, 8)}[a.cfg]
sc = S.make_scene(P, W, H, deg, seed=0, hdr=hdr)
cams = S.blur_poses(W, H, poses) if poses > 1 else None
rs, _, _ = Hh.settings_from_scene(sc, "cuda", cams, hdr, requires_grad=True)
leaves = [t.cuda().requires_grad_(True) for t in (sc.means3D, torch.zeros_like(sc.means3D), sc.opacities)]
cap = {"c3": 8500000, "c4": 68000000}[a.cfg]
shs = sc.shs.cuda()
out = GaussianRasterizer(rs, capacity=cap)(leaves[0], leaves[1], leaves[2], shs=shs.clone().requires_grad_(True),
                                           scales=sc.scales.cuda().requires_grad_(True), rotations=sc.rotations.cuda().requires_grad_(True))
torch.cuda.synchronize()
side = torch.cuda.Stream()
col = torch.empty(poses, P, 3, device="cuda")


def colour():
    for k in range(poses):
        torch.sum(shs, dim=1, out=col[k])


def binning():
    replay_forward(out[0], L.HS_STAGE_BIN)


def both():
    e = torch.cuda.Event(); e.record()
    with torch.cuda.stream(side):
        side.wait_event(e)
        colour()
        e2 = torch.cuda.Event(); e2.record()
    binning()
    torch.cuda.current_stream().wait_event(e2)


high = torch.cuda.Stream(priority=-1)


def both_prio():   # the binning chain on a high-priority stream, the stand-in on the default one
    e = torch.cuda.Event(); e.record()
    with torch.cuda.stream(high):
        high.wait_event(e)
        binning()
        e2 = torch.cuda.Event(); e2.record()
    colour()
    torch.cuda.current_stream().wait_event(e2)


def serial():
    colour(); binning()


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.iters)]
    for e0, e1 in evs:
        e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
    return ts[len(ts) // 2] * 1e3


for name, fn in (("binning", binning), ("colour stand-in", colour), ("serial", serial), ("concurrent", both), ("concurrent, binning at high priority", both_prio), ("binning", binning)):
    print(f"{a.cfg} {name:38s} {timed(fn):8.1f} us")
