"""Where the workgroups of the c2 render backward run, and for how long (development aid): per-workgroup clock stamps and
hardware ids from the STATS instantiation -> the block index -> (XCD, SE, CU) map of a fresh launch, per-CU busy time, the
longest workgroup."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench as B
from casualhdrsplat_amd.rasterizer import render_stats
cfg = B.CONFIGS["c2"]
dev = torch.device("cuda", 0)
step, state, mk, sc, dL, plist = B.build_step(cfg, 0, 1, dev)
out = step(); torch.cuda.synchronize()
R = B.derived_counts(out, cfg[1], cfg[2], cfg[5])[0]
state["rast"] = mk(int(R * 1.25) + 4096)
for p_ in plist: p_.grad = None
out = state["rast"]["allreduce"](*[plist[i] for i in (0, 1, 2)], shs=plist[3], scales=plist[4], rotations=plist[5])
render_stats(out[0], dL, timeline=True)
tl = render_stats(out[0], dL, timeline=True)["bwd_timeline"].numpy()
t0 = tl[:, 0].min()
s, e = (tl[:, 0] - t0) * 10e-3, (tl[:, 1] - t0) * 10e-3
xcc = (tl[:, 2] >> 32) & 0xF
hw = tl[:, 2] & 0xFFFFFFFF
cu = (hw >> 8) & 0xF; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
dur = e - s
n = len(s)
print(f"{n} workgroup slots; span {e.max():.1f} us; duration mean {dur.mean():.1f} median {np.median(dur):.1f} p90 {np.percentile(dur, 90):.1f} max {dur.max():.1f}")
print("start times: p50 %.1f p90 %.1f max %.1f us" % (np.percentile(s, 50), np.percentile(s, 90), s.max()))
print("block index -> (xcc, se, sh, cu) of the first 40 blocks:", [(int(b), int(xcc[b]), int(se[b]), int(sh[b]), int(cu[b])) for b in range(40)])
x0 = np.nonzero(xcc == xcc[0])[0][:48]
print("blocks of XCC", int(xcc[0]), "in launch order -> (se, sh, cu):", [(int(b), int(se[b]), int(sh[b]), int(cu[b])) for b in x0])
key = xcc * 1000 + se * 100 + sh * 20 + cu
busy = {k: dur[key == k].sum() for k in np.unique(key)}
vals = np.array(list(busy.values()))
print(f"{len(busy)} distinct CUs; summed workgroup time per CU: mean {vals.mean():.0f} min {vals.min():.0f} max {vals.max():.0f} us; workgroups per CU min {min((key == k).sum() for k in busy)} max {max((key == k).sum() for k in busy)}")
late = np.argsort(e)[-10:]
print("the ten last workgroups to finish: (block, start, duration)", [(int(b), round(float(s[b]), 1), round(float(dur[b]), 1)) for b in late])
