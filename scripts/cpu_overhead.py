"""Host-side enqueue time per step vs GPU time (development aid)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda", 0)
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
step, state, make_rasterizer, sc, dL, plist = bench.build_step(cfg, 0, 1, dev)
out = step(); torch.cuda.synchronize()
R = bench.derived_counts(out, cfg[1], cfg[2], cfg[5])[0]
state["rast"] = make_rasterizer(int(R * 1.25) + 4096)
for _ in range(5): step()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(20): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"enqueue {1e3*(t1-t0)/20:.3f} ms/step, total {1e3*(t2-t0)/20:.3f} ms/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
