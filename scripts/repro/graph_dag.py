"""Shape of the DAG a captured formation step becomes (HS_GRAPH_DUMP -> hipGraphDebugDotPrint): nodes, edges, and the
nodes with more than one successor or predecessor -- a single-stream capture must be ONE chain.
usage: python scripts/repro/graph_dag.py"""
import os, re, sys, collections
os.environ["HS_GRAPH_DUMP"] = "/tmp/hs_graph.dot"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts", "repro"))
import importlib
src = open(os.path.join(ROOT, "scripts", "repro", "graph_bisect.py")).read().split("\nr = {}\n")[0]   # the definitions only
ns = {"__file__": os.path.join(ROOT, "scripts", "repro", "graph_bisect.py"), "__name__": "bisect_defs"}
exec(compile(src, "graph_bisect_defs", "exec"), ns)
src2 = open(os.path.join(ROOT, "scripts", "repro", "graph_bisect.py")).read()
defs = src2[src2.index("def formation_variant"):src2.index("\nf = {}\n")]
exec(compile(defs, "graph_bisect_defs2", "exec"), ns)
ns["formation_variant"]("formation step, one frame", frames=1)
dot = open("/tmp/hs_graph.dot").read()
edges = re.findall(r'"?([\w\.]+)"?\s*->\s*"?([\w\.]+)"?', dot)
succ, pred = collections.Counter(a for a, _ in edges), collections.Counter(b for _, b in edges)
nodes = set(a for a, _ in edges) | set(b for _, b in edges)
print("nodes", len(nodes), "edges", len(edges), "forks (>1 successor)", sum(v > 1 for v in succ.values()),
      "joins (>1 predecessor)", sum(v > 1 for v in pred.values()), "roots", sum(n not in pred for n in nodes))
labels = dict(re.findall(r'"?([\w\.]+)"?\s*\[[^\]]*label="([^"]*)"', dot))
for n, v in succ.items():
    if v > 1:
        print("FORK at", n, labels.get(n, "")[:120].replace("\n", " "), "->", [(b, labels.get(b, "")[:60].replace("\n", " ")) for a, b in edges if a == n][:4])
