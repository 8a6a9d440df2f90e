"""In-situ leg of the captured-graph hazards: the multi-frame image-formation step of
examples/train_synthetic.py, eager and captured (graphs.GraphedStep), with the loss as a PLAIN .mean() -- the reduction
round 5 reported as reading 94.41 from the second replay on.  Neither hazard reproduces in isolation
(scripts/repro/graph_memset.hip, graph_torch_sum.py), so what is left is this step.  Prints the per-step losses of both
runs; identical histories = the hazard is gone (or never was the reduction's).
usage: python scripts/repro/graph_step_mean.py [steps]"""
import os, sys
os.environ["HS_EXAMPLE_PLAIN_MEAN"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
import train_synthetic as ex
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
hist = {}
for graph in (False, True):
    r = ex.run(steps=steps, quiet=True, graph=graph)
    hist[graph] = [h["loss"] for h in r["history"]]
    print("graph" if graph else "eager", " ".join(f"{v:.6f}" for v in hist[graph]), flush=True)
worst = max(abs(a - b) / max(abs(a), 1e-12) for a, b in zip(hist[False], hist[True]))
print("RESULT graph_step_mean: worst relative difference of the printed losses", f"{worst:.3g}",
      "(reproduced)" if worst > 1e-3 else "(not reproduced: the captured step prints the eager losses)")
