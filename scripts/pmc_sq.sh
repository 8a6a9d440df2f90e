#!/bin/bash
# Runs ON the MI355X box: SQ counters of the render kernels for the library named by HS_LIB_PATH -> gpurun_out/pmc_sq_<tag>/
tag=${1:-cur}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=$ROOT/gpurun_out/pmc_sq_$tag
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_INST_LEVEL_LDS SQ_LDS_UNALIGNED_STALL SQ_LDS_ATOMIC_RETURN SQ_LDS_MEM_VIOLATIONS" "SQ_IFETCH SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/s$i -- python3 $ROOT/scripts/step_c3.py --steps 3 > $O/s$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/s*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "render_" not in n: continue
        n = n.replace("hs::(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$O/summary.csv", "w") as o:
    o.write("kernel,counter,avg_per_launch\n")
    for k in sorted(agg):
        for c in sorted(agg[k]):
            v = agg[k][c]
            o.write(f"{k},{c},{sum(v)/len(v):.4e}\n")
print(open("$O/summary.csv").read())
PY
