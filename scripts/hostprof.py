"""Development aid (GPU box): host enqueue time per step against the GPU time, and the host profile (cProfile).
usage: python scripts/hostprof.py [config]"""
import os, sys, time, torch, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
step, state, make_rasterizer, sc, dL, plist = bench.build_step(cfg, 0, 1, dev)
out = step(); torch.cuda.synchronize()
R = bench.derived_counts(out, cfg[1], cfg[2], cfg[5])[0]
state["rast"] = make_rasterizer(int(R * 1.25) + 4096)
for _ in range(20): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/200:.3f} ms/step, total {1e3*(t2-t0)/200:.3f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
