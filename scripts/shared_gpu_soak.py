"""Two (or more) processes running the c3 training step on ONE GPU at the same time: how often does the host move a process
to ticket order -- because waiting blocks of a radix pass had to help silent predecessors (hs_counters.reserved[4]), or,
before helping existed, because a pass gave up (hs_counters.overflow = 2 -> SortChainStalled) -- and what a step costs in
each mode; ticket order from the start (HS_SORT_TICKETS=1) for comparison.
usage: python scripts/shared_gpu_soak.py [--procs 2] [--steps 300] [--config c3]"""
import argparse, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys, json, time, warnings
sys.path.insert(0, os.environ["HS_ROOT"])
import torch, bench as B
from casualhdrsplat_amd import SortChainStalled, _lib as L
warnings.simplefilter("ignore")
cfg = B.CONFIGS[os.environ["HS_CFG"]]
dev = torch.device("cuda", 0)
step, state, mk, sc, dL, plist = B.build_step(cfg, 0, 1, dev)
out = step(); torch.cuda.synchronize()
R = B.derived_counts(out, cfg[1], cfg[2], cfg[5])[0]
state["rast"] = mk(int(R * 1.25) + 4096)
lib = L.load()
stalls = 0
t0 = time.time()
n = int(os.environ["HS_STEPS"])
keep = os.environ.get("HS_KEEP_BLOCKIDX") == "1"
for i in range(n):
    before = lib.hs_sort_tickets(-1)
    step()                      # (bench's step repeats a stalled step itself; count the switches)
    if lib.hs_sort_tickets(-1) != before:
        stalls += 1
        if keep:
            lib.hs_sort_tickets(0)   # keep provoking: back to blockIdx order
torch.cuda.synchronize()
print(json.dumps({"pid": os.getpid(), "steps": n, "switches_to_ticket_order": stalls, "tickets_at_end": lib.hs_sort_tickets(-1),
                  "ms_per_step": (time.time() - t0) / n * 1e3}))
"""

ap = argparse.ArgumentParser()
ap.add_argument("--procs", type=int, default=2); ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--config", default="c3")
a = ap.parse_args()
for label, extra in (("blockIdx order, switching on the first stall", {}),
                     ("blockIdx order kept (switched back after every stall)", {"HS_KEEP_BLOCKIDX": "1"}),
                     ("ticket order from the start", {"HS_SORT_TICKETS": "1"})):
    env = dict(os.environ, HS_ROOT=ROOT, HS_CFG=a.config, HS_STEPS=str(a.steps), **extra)
    ps = [subprocess.Popen([sys.executable, "-c", CHILD], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
          for _ in range(a.procs)]
    outs = [p.communicate(timeout=1800) for p in ps]
    res = []
    for p, (so, se) in zip(ps, outs):
        line = [l for l in so.splitlines() if l.startswith("{")]
        res.append(json.loads(line[-1]) if line else {"rc": p.returncode, "stderr": se[-400:]})
    print(json.dumps({"mode": label, "procs": a.procs, "results": res}))
