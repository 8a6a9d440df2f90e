#!/bin/bash
# Runs ON the MI355X box (gpurun): regenerates the raw material of profiles/ under gpurun_out/prof_final/.
# usage: bash scripts/collect_profiles.sh [pmc|bench|all] [c3|c2|c4]
#   pmc   -- rocprofv3 --pmc passes over scripts/step_c3.py at the given BASELINE config (FETCH_SIZE and WRITE_SIZE in
#            separate passes -- they do not fit the TCC counter slots together -- and, for c3, four passes of SQ counters)
#   bench -- the bench lines of c3 / c4 / c2, rocprofv3 --kernel-trace --stats of bench.py at c3, c2 and c4, the bare
#            two-rank plumbing run, the backward's timeline and the VALU-rate microbenchmark
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
what=${1:-all}
CFG=${2:-c3}
O=$ROOT/gpurun_out/prof_final
[ "$CFG" = c3 ] || O=$ROOT/gpurun_out/prof_final_$CFG
rm -rf $O   # counters are averaged over every file found: never mix runs
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
case $CFG in
  c3) STEP="python3 $ROOT/scripts/step_c3.py --steps 3 --capacity 8500000" ;;
  c2) STEP="python3 $ROOT/scripts/step_c3.py --steps 5 --P 100000 --W 800 --H 800 --deg 0 --hdr 0 --capacity 900000" ;;
  c4) STEP="python3 $ROOT/scripts/step_c3.py --steps 2 --poses 8 --capacity 68000000" ;;
esac
if [ "$what" = pmc ] || [ "$what" = all ]; then
  # HBM-side traffic: FETCH_SIZE and WRITE_SIZE in separate passes (they do not fit the TCC counter slots together)
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- $STEP > $O/pmc_$c.log 2>&1
  done
  if [ "$CFG" = c3 ]; then
  # render_bwd_kernel without the CRF gradient's tail workgroups (the tile replay alone): its launches are the LAST
  # `steps` ones of the kernel in these passes (the first step is a whole backward); fold_profiles.py takes those
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_alone_$c -- $STEP --render-bwd-alone > $O/pmc_alone_$c.log 2>&1
  done
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_sq$i -- $STEP > $O/pmc_sq$i.log 2>&1
  done
  fi
fi
if [ "$what" = bench ] || [ "$what" = all ]; then
  if [ "$CFG" = c3 ]; then
  python3 $ROOT/bench.py --steps 50 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err
  python3 $ROOT/bench.py --config c4 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err
  python3 $ROOT/bench.py --config c2 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
  fi
  # per-kernel totals of the bench command at this config (the driver's command at c3)
  extra=""; [ "$CFG" = c3 ] || extra="--config $CFG"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $ROOT/bench.py $extra --steps 10 --warmup 3 --no-cpu-baseline --no-extras --graph off > $O/stats.log 2>&1
  if [ "$CFG" = c3 ]; then
  # plumbing check of the bare multi-GPU invocation on a one-GPU box: bench.py starts its own two ranks (gloo, shared GPU)
  HS_BENCH_BACKEND=gloo timeout 900 python3 $ROOT/bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_2rank_gloo_one_gpu.json 2> $O/bench_2rank.err
  python3 $ROOT/scripts/timeline.py > $O/timeline.txt 2>&1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 $ROOT/scripts/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate > $O/valu_rate.txt 2>&1 || true
  fi
fi
ls $O
