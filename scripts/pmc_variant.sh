#!/bin/bash
# Runs ON the MI355X box: VALU instruction count and L2-fabric traffic of the render kernels for the library named by
# HS_LIB_PATH (A/B builds) -> one line per kernel.  usage: HS_LIB_PATH=... bash scripts/pmc_variant.sh <tag>
tag=${1:-cur}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=$ROOT/gpurun_out/pmc_variant_$tag
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/$n -- python3 $ROOT/scripts/step_c3.py --steps 2 --capacity 8500000 > $O/$n.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "render_" not in n: continue
        n = n.replace("hs::(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    c = {x: sum(v) / len(v) for x, v in agg[k].items()}
    hbm = (2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024
    busy = c.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (1024 * c.get("GRBM_GUI_ACTIVE", 1) / 8)
    print(f"$tag {k}: SQ_INSTS_VALU {c.get('SQ_INSTS_VALU', 0):.3e}  VALU-busy {busy:.3f}  L2-fabric bytes {hbm/1e6:.1f} MB")
PY
