"""A/B aid: HIP-event time of the render forward / backward stages at a BASELINE config for the library named by
HS_LIB_PATH (default: the shipped one).  usage: HS_LIB_PATH=... python scripts/ab_render.py [--config c3] [--iters 20]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench as B
from casualhdrsplat_amd import _lib as L
from casualhdrsplat_amd.rasterizer import replay_backward, replay_forward, render_stats

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c3"); ap.add_argument("--iters", type=int, default=20); ap.add_argument("--stats", action="store_true")
a = ap.parse_args()
cfg = B.CONFIGS[a.config]
dev = torch.device("cuda", 0)
step, state, mk, sc, dL, plist = B.build_step(cfg, 0, 1, dev)
out = step(); torch.cuda.synchronize()
R = B.derived_counts(out, cfg[1], cfg[2], cfg[5])[0]
state["rast"] = mk(int(R * 1.25) + 4096)
for p_ in plist: p_.grad = None
out = state["rast"]["allreduce"](*[plist[i] for i in (0, 1, 2)], shs=plist[3], scales=plist[4], rotations=plist[5])
res = {"lib": os.path.basename(L.LIB_PATH)}
res["render_bwd_ms"], res["render_bwd_med"] = B.time_stage(lambda: replay_backward(out[0], dL, L.HS_BWD_RENDER), a.iters)
res["segsum_pre_bwd_ms"] = B.time_stage(lambda: replay_backward(out[0], dL, L.HS_BWD_PREPROCESS), a.iters)[0]
res["render_fwd_ms"], res["render_fwd_med"] = B.time_stage(lambda: replay_forward(out[0], L.HS_STAGE_RENDER), a.iters)
res["binning_ms"] = B.time_stage(lambda: replay_forward(out[0], L.HS_STAGE_BIN), a.iters)[0]
t = B.time_stage(step, a.iters)
res["step_ms"], res["step_med"] = t
if a.stats and hasattr(L.load(), "hs_render_stats"):
    st = render_stats(out[0], dL)
    res["bwd_trips"], res["bwd_empty"] = st["bwd_trips"], st["bwd_empty_trips"]
    res["bwd_lane_util"] = st["bwd_active_pixels"] / (128.0 * max(st["bwd_trips"], 1))
    res["fwd_trips"] = st["fwd_trips"]
print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in res.items()}))
