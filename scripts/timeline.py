"""Occupancy timeline of the render backward at c3 (development aid): per-workgroup start/end stamps from the STATS
instantiation -> resident workgroups over time, per-XCD finish times, duration spread."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench as B
from casualhdrsplat_amd.rasterizer import render_stats
cfg = B.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
dev = torch.device("cuda", 0)
step, state, mk, sc, dL, plist = B.build_step(cfg, 0, 1, dev)
out = step(); torch.cuda.synchronize()
R = B.derived_counts(out, cfg[1], cfg[2], cfg[5])[0]
state["rast"] = mk(int(R * 1.25) + 4096)
for p_ in plist: p_.grad = None
out = state["rast"]["allreduce"](*[plist[i] for i in (0, 1, 2)], shs=plist[3], scales=plist[4], rotations=plist[5])
render_stats(out[0], dL, timeline=True)
st = render_stats(out[0], dL, timeline=True)
tl = st["bwd_timeline"].numpy()
t0 = tl[:, 0].min()
s, e = (tl[:, 0] - t0) * 10e-3, (tl[:, 1] - t0) * 10e-3   # microseconds (100 MHz clock)
xcc = (tl[:, 2] >> 32) & 0xF
hw = tl[:, 2] & 0xFFFFFFFF
cu = (hw >> 8) & 0xF; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
dur = e - s
print(f"kernel span {e.max():.1f} us; WG duration mean {dur.mean():.1f} med {np.median(dur):.1f} p10 {np.percentile(dur,10):.1f} p90 {np.percentile(dur,90):.1f} max {dur.max():.1f}")
print("sum of WG durations / span =", round(dur.sum() / e.max(), 1), "resident WGs on average (slots: 256 CUs x 12 = 3072)")
for x in range(8):
    m = xcc == x
    print(f"XCC {x}: {m.sum()} WGs, work {dur[m].sum()/1e3:.2f} ms-WG, first start {s[m].min():.1f}, last end {e[m].max():.1f}, distinct (se,sh,cu) {len(set(zip(se[m], sh[m], cu[m])))}")
edges = np.linspace(0, e.max(), 23)
occ = [(np.minimum(e, b) - np.maximum(s, a)).clip(min=0).sum() / (b - a) for a, b in zip(edges[:-1], edges[1:])]
print("resident WGs per 1/22 of the span:", [int(o) for o in occ])
# tail: time from the moment the first XCD runs out of work to the end
ends = sorted(e[xcc == x].max() for x in range(8))
print("XCD finish times:", [round(float(v), 1) for v in ends], " spread", round(float(ends[-1] - ends[0]), 1))
order = np.argsort(s)
late = order[-800:]
print("durations of the last 800 WGs to start: mean", round(float(dur[late].mean()), 1), " vs all", round(float(dur.mean()), 1))
print("start time of WG #0,1000,...:", [round(float(s[order[i]]), 1) for i in range(0, len(s), 1000)])
