#!/bin/bash
# Runs ON the MI355X box: bench.py --one-seed with and without an environment setting, alternating, one config per call.
# usage: bash scripts/ab_env.sh <config> <VAR=value> [rounds]
ROOT=${GRAFT_REPO_ROOT:-$PWD}
c=$1; kv=$2; rounds=${3:-3}
steps=40; [ $c = c4 ] && steps=12; [ $c = c2 ] && steps=100
for rep in $(seq $rounds); do
  for mode in base "$kv"; do
    if [ "$mode" = base ]; then pre=""; else pre="$kv"; fi
    env $pre timeout 200 python3 $ROOT/bench.py --config $c --steps $steps --warmup 5 --one-seed --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages_ms']
print('$c %-24s' % '$mode', 'step %.4f' % d['ms_per_step'], 'pre+bin %.4f' % s['preprocess_fwd_and_binning'], 'bin %.4f' % s['binning'], 'rbwd %.4f' % s['render_bwd'], 'rfwd %.4f' % s['render_fwd'], 'pbwd %.4f' % s['segsum_and_preprocess_bwd'])"
  done
done
