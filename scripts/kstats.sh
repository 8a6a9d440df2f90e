#!/bin/bash
# Runs ON the MI355X box: per-kernel time of scripts/step_c3.py (or "$@") -> prints the top of the stats table
ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=$ROOT/gpurun_out/kstats
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $ROOT/scripts/step_c3.py --steps 10 "$@" > $O/log.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$O/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:26]:
    n = r["Name"].replace("hs::(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:60]
    print(f"{n:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us  tot {float(r['TotalDurationNs'])/1e6:7.2f} ms")
PY
