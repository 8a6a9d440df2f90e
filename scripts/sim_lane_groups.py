"""Development aid (CPU, uses the oracle as the reference forward): how many trips the backward replay would need if the
lanes of a wave were split into independent groups, each owning a block of the half tile and walking its own list of
takers, instead of one list per half tile.  Same Gaussian density and footprint distribution as c3 on a 480x272 frame.
Reproduces the measured lane utilisation of the shipped kernel (0.38) and the trip ratio the built variant showed
(8 groups of 4x4 pixels: x0.70) -- profiles/README.md, "lane groups"."""
import sys, numpy as np, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from casualhdrsplat_amd import synthetic as S
from oracle import c_oracle as O
import helpers as Hh
W, H = 480, 272
P = int(1_000_000 * W * H / (1920 * 1080))
sc = S.make_scene(P, W, H, 3, seed=0, hdr=True)
cam = Hh.oracle_camera(O, sc)
f = O.forward(cam, sc.means3D.numpy(), sc.opacities.numpy(), shs=sc.shs.numpy(), scales=sc.scales.numpy(), rotations=sc.rotations.numpy())
xy, co, pl, rg = f["xy"], f["conic_opacity"], f["point_list"], f["ranges"]
gx, gy = (W + 15) // 16, (H + 15) // 16
KBS = (32, 48, 64, 128)
res = {}
def add(k, v): res[k] = res.get(k, 0) + v
partitions = {"8x4": (8, 4), "16x2": (16, 2), "4x8": (4, 8), "4x4": (4, 4), "8x8": (8, 8), "16x4": (16, 4), "8x2": (8,2)}
for t in range(gx * gy):
    ty, tx = divmod(t, gx)
    b, e = int(rg[t, 0]), int(rg[t, 1])
    if e <= b: continue
    ids = pl[b:e]
    px = (tx * 16 + np.arange(16))[None, :].repeat(16, 0).astype(np.float32)
    py = (ty * 16 + np.arange(16))[:, None].repeat(16, 1).astype(np.float32)
    inside = (px < W) & (py < H)
    T = np.ones((16, 16), np.float32); done = ~inside
    act = np.zeros((e - b, 16, 16), bool)
    for i, g in enumerate(ids):
        dx = xy[g, 0] - px; dy = xy[g, 1] - py
        A, B, C, o = co[g]
        power = -0.5 * (A * dx * dx + C * dy * dy) - B * dx * dy
        alpha = np.minimum(0.99, o * np.exp(power))
        ok = (power <= 0) & (alpha >= 1 / 255) & ~done
        tt = T * (1 - alpha)
        stop = ok & (tt < 1e-4)
        done |= stop
        ok &= ~stop
        T = np.where(ok, tt, T)
        act[i] = ok
        if done.all(): break
    n = i + 1
    act = act[:n]
    for w in range(2):                       # half tiles: rows 0..7 and 8..15
        a = act[:, 8 * w:8 * w + 8, :]       # [n, 8, 16]
        took = a.any(axis=(1, 2))
        add("trips_old", int(took.sum())); add("active_px", int(a.sum()))
        for name, (bw, bh) in partitions.items():
            blk = a.reshape(n, 8 // bh, bh, 16 // bw, bw).any(axis=(2, 4)).reshape(n, -1)   # [n, nblk]
            add("sum_" + name, int(blk.sum()))
            # per batch of KB entries (list positions), trips = max over blocks of takers in batch
            for KB in KBS:
                tr = 0
                # batches are aligned to the END of the processed range like the kernel (top batch partial)
                for s0 in range(0, n, KB):
                    tr += int(blk[s0:s0 + KB].sum(axis=0).max())
                add("trips_%d_" % KB + name, tr)
            add("trips_nobatch_" + name, int(blk.sum(axis=0).max()))
print("half-tile trips", res["trips_old"], "lane util", res["active_px"] / (128 * res["trips_old"]))
for name, (bw, bh) in partitions.items():
    nb = (16 // bw) * (8 // bh)
    print(f"{name}: blocks/wave {nb} sum takers {res['sum_'+name]} (x{res['sum_'+name]/res['trips_old']:.2f}) " + " ".join("KB%d x%.3f" % (KB, res['trips_%d_' % KB + name]/res['trips_old']) for KB in KBS) + f" nobatch x{res['trips_nobatch_'+name]/res['trips_old']:.3f}")
