#!/bin/bash
# Runs ON the MI355X box: bench.py --one-seed for several builds of the library, two rounds, one config per call.
# usage: bash scripts/ab_many.sh <config> <lib1.so> <lib2.so> ...
ROOT=${GRAFT_REPO_ROOT:-$PWD}
c=$1; shift
steps=40; [ $c = c4 ] && steps=12; [ $c = c2 ] && steps=100
for rep in 1 2; do
  for lib in "$@"; do
    HS_LIB_PATH=$ROOT/casualhdrsplat_amd/$lib timeout 200 python3 $ROOT/bench.py --config $c --steps $steps --warmup 5 --one-seed --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages_ms']
print('$c %-24s' % '$lib', 'step %.4f' % d['ms_per_step'], 'pre+bin %.4f' % s['preprocess_fwd_and_binning'], 'bin %.4f' % s['binning'], 'rbwd %.4f' % s['render_bwd'], 'rfwd %.4f' % s['render_fwd'], 'pbwd %.4f' % s['segsum_and_preprocess_bwd'])"
  done
done
