#!/bin/bash
# Runs ON the MI355X box: A/B of two builds of the library inside ONE gpurun call (box-to-box spread is +-1 %).
# usage: bash scripts/ab_bench.sh <libA.so> <libB.so> [configs...]   -> alternating runs of bench.py --one-seed per config
ROOT=${GRAFT_REPO_ROOT:-$PWD}
A=$1; B=$2; shift 2
CFGS=${@:-c3}
for c in $CFGS; do
  for rep in 1 2; do
    for lib in $A $B; do
      steps=40; [ $c = c4 ] && steps=12; [ $c = c2 ] && steps=100
      HS_LIB_PATH=$ROOT/casualhdrsplat_amd/$lib timeout 200 python3 $ROOT/bench.py --config $c --steps $steps --warmup 5 --one-seed --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages_ms']
print('$c $lib', 'step %.4f' % d['ms_per_step'], 'pre+bin %.4f' % s['preprocess_fwd_and_binning'], 'bin %.4f' % s['binning'], 'rbwd %.4f' % s['render_bwd'], 'rfwd %.4f' % s['render_fwd'], 'pbwd %.4f' % s['segsum_and_preprocess_bwd'])"
    done
  done
done
