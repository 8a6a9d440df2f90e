#!/bin/bash
# Runs ON the MI355X box: per-kernel averages of the BINNING stage alone (scripts/time_binning.py) for several builds.
# usage: bash scripts/kstats_bin.sh "<lib1.so> <lib2.so> ..." [time_binning.py args...]
ROOT=${GRAFT_REPO_ROOT:-$PWD}
LIBS=$1; shift
cd /tmp && export TMPDIR=/tmp
for lib in $LIBS; do
  O=$ROOT/gpurun_out/kbin_$lib; rm -rf $O; mkdir -p $O
  HS_LIB_PATH=$ROOT/casualhdrsplat_amd/$lib timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $ROOT/scripts/time_binning.py "$@" > $O/log.txt 2>&1
  tail -1 $O/log.txt
done
python3 - <<PY
import csv, glob
libs = "$LIBS".split()
tabs = []
for lib in libs:
    fs = glob.glob("$ROOT/gpurun_out/kbin_%s/*/*kernel_stats.csv" % lib)
    t = {}
    if fs:
        for r in csv.DictReader(open(fs[0])):
            n = r["Name"].replace("hs::(anonymous namespace)::", "").replace("void ", "")
            if "at::native" in n or "rocclr" in n: continue
            t[n[:56]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
    tabs.append(t)
names = sorted(set().union(*tabs), key=lambda n: -max(t.get(n, (0, 0))[0] * t.get(n, (0, 0))[1] for t in tabs))
print("%-56s" % "kernel (us per launch)", *["%12s" % l.replace("libhdrsplat", "").replace(".so", "")[-12:] for l in libs])
for n in names[:22]:
    print("%-56s" % n, *["%12.1f" % t.get(n, (0, 0.0))[1] for t in tabs])
PY
