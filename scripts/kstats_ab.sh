#!/bin/bash
# Runs ON the MI355X box: per-kernel averages of scripts/step_c3.py for two builds of the library, side by side.
# usage: bash scripts/kstats_ab.sh <libA.so> <libB.so> [step_c3.py args...]
ROOT=${GRAFT_REPO_ROOT:-$PWD}
A=$1; B=$2; shift 2
cd /tmp && export TMPDIR=/tmp
for lib in $A $B; do
  O=$ROOT/gpurun_out/kstats_$lib; rm -rf $O; mkdir -p $O
  HS_LIB_PATH=$ROOT/casualhdrsplat_amd/$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $ROOT/scripts/step_c3.py --steps 10 "$@" > $O/log.txt 2>&1
done
python3 - <<PY
import csv, glob
tabs = []
for lib in ("$A", "$B"):
    f = glob.glob("$ROOT/gpurun_out/kstats_%s/*/*kernel_stats.csv" % lib)[0]
    t = {}
    for r in csv.DictReader(open(f)):
        n = r["Name"].replace("hs::(anonymous namespace)::", "").replace("void ", "")
        if "at::native" in n or "rocclr" in n: n = n.split("<")[0]
        t[n[:70]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
    tabs.append(t)
names = sorted(set(tabs[0]) | set(tabs[1]), key=lambda n: -max(tabs[0].get(n, (0, 0))[1] * tabs[0].get(n, (0, 0))[0], tabs[1].get(n, (0, 0))[1] * tabs[1].get(n, (0, 0))[0]))
for n in names[:30]:
    a, b = tabs[0].get(n, (0, 0.0)), tabs[1].get(n, (0, 0.0))
    print(f"{n:70s} A {a[0]:4d} x {a[1]:8.1f} us   B {b[0]:4d} x {b[1]:8.1f} us")
PY
