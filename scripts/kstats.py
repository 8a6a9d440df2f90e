"""Print a rocprofv3 kernel_stats.csv compactly (development aid)."""
import csv, glob, sys
f = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob("gpurun_out/*/*/*_kernel_stats.csv"))[-1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for r in list(csv.DictReader(open(f)))[:n]:
    name = r["Name"].replace("hs::(anonymous namespace)::", "").split("(")[0][:44]
    print(f"{name:44s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f} tot_ms={float(r['TotalDurationNs'])/1e6:8.2f} {float(r['Percentage']):5.1f}%")
