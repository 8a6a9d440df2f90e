#!/bin/bash
# Runs ON the MI355X box: per-kernel averages (rocprofv3 --kernel-trace --stats) of scripts/step_c3.py for several builds of
# the library, side by side.   usage: bash scripts/kstats_many.sh "<lib1.so> <lib2.so> ..." [step_c3.py args...]
ROOT=${GRAFT_REPO_ROOT:-$PWD}
LIBS=$1; shift
cd /tmp && export TMPDIR=/tmp
for lib in $LIBS; do
  O=$ROOT/gpurun_out/kstats_$lib; rm -rf $O; mkdir -p $O
  HS_LIB_PATH=$ROOT/casualhdrsplat_amd/$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $ROOT/scripts/step_c3.py --steps 10 "$@" > $O/log.txt 2>&1
done
python3 - <<PY
import csv, glob
libs = "$LIBS".split()
tabs = []
for lib in libs:
    f = glob.glob("$ROOT/gpurun_out/kstats_%s/*/*kernel_stats.csv" % lib)[0]
    t = {}
    for r in csv.DictReader(open(f)):
        n = r["Name"].replace("hs::(anonymous namespace)::", "").replace("void ", "")
        if "at::native" in n or "rocclr" in n: n = n.split("<")[0]
        t[n[:60]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
    tabs.append(t)
names = sorted(set().union(*tabs), key=lambda n: -max(t.get(n, (0, 0))[0] * t.get(n, (0, 0))[1] for t in tabs))
print("%-60s" % "kernel (us per launch)", *["%14s" % l.replace("libhdrsplat", "").replace(".so", "")[-14:] for l in libs])
for n in names[:28]:
    print("%-60s" % n, *["%14.1f" % t.get(n, (0, 0.0))[1] for t in tabs])
PY
