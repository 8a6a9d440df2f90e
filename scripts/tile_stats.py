"""Distribution of per-tile work at c3 (development aid)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from casualhdrsplat_amd import inspect_state
dev = torch.device("cuda", 0)
cfg = bench.CONFIGS["c3"]
step, state, make_rasterizer, sc, dL, plist = bench.build_step(cfg, 0, 1, dev)
out = step(); torch.cuda.synchronize()
st = inspect_state(out[0])
nc = st["n_contrib"].to(torch.int64)[0]
H, W = nc.shape
gx, gy = (W + 15) // 16, (H + 15) // 16
pad = torch.zeros(gy * 16, gx * 16, dtype=torch.int64, device=nc.device); pad[:H, :W] = nc
nproc = pad.reshape(gy, 16, gx, 16).amax(dim=(1, 3)).reshape(-1).float()
lens = (st["ranges"][:, 1] - st["ranges"][:, 0]).float()
for name, v in (("list length", lens), ("n_proc (max n_contrib)", nproc)):
    q = torch.quantile(v, torch.tensor([0.0, 0.05, 0.25, 0.5, 0.75, 0.95, 1.0], device=v.device))
    print(name, "mean %.0f std %.0f" % (v.mean(), v.std()), "quantiles", [int(x) for x in q])
