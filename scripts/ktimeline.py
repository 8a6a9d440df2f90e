"""Development aid: per-kernel timeline of the last step of the newest gpurun_out/kstats trace (scripts/kstats.sh)."""
import csv, glob, os, sys
f = max(glob.glob("gpurun_out/kstats/runc/*_kernel_trace.csv"), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "preprocess_fwd" in r["Kernel_Name"]]
seg = rows[idx[-2]:idx[-1]]
t0 = int(seg[0]["Start_Timestamp"])
for r in seg:
    n = r["Kernel_Name"]
    if "at::" in n: continue
    n = n.replace("hs::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:34]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))
