"""Aggregate rocprofv3 counter_collection.csv per kernel name substring (development aid)."""
import csv, collections, glob, sys
pats = sys.argv[2:] or ["render_fwd", "render_bwd"]
for f in sorted(glob.glob(sys.argv[1])):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        for p in pats:
            if p in r["Kernel_Name"]:
                agg[p][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for p, d in agg.items():
        print(p, " ".join(f"{c}={sum(v)/len(v):.3e}" for c, v in sorted(d.items())))
