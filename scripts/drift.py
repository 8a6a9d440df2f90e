"""Does the step time drift with the step count? (development aid)"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda", 0)
cfg = bench.CONFIGS["c3"]
step, state, make_rasterizer, sc, dL, plist = bench.build_step(cfg, 0, 1, dev)
out = step(); torch.cuda.synchronize()
R = bench.derived_counts(out, cfg[1], cfg[2], cfg[5])[0]
mode = sys.argv[1] if len(sys.argv) > 1 else "capacity"
if mode == "capacity":
    state["rast"] = make_rasterizer(int(R * 1.25) + 4096)
for blk in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"{mode} block {blk}: {1e3*(t1-t0)/10:.3f} ms/step  alloc {torch.cuda.memory_allocated()/2**20:.0f} MiB reserved {torch.cuda.memory_reserved()/2**20:.0f} MiB", flush=True)
