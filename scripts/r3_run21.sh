#!/bin/bash
set -u
for v in "" p12 p20 p24 "" p24; do
  s=${v:+_$v}
  HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat$s.so timeout 300 python scripts/ab_render.py --iters 30 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['lib'], 'c3 binning', d['binning_ms'], 'step', d['step_ms'], d['step_med'])"
done
for v in "" p24; do
  s=${v:+_$v}
  for c in c2 c4; do
  HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat$s.so timeout 300 python scripts/ab_render.py --iters 10 --config $c 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['lib'], '$c binning', d['binning_ms'], 'step', d['step_ms'], d['step_med'])"
  done
done
HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat_p24.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "golden or ldr_forward or 14400 or radix" 2>&1 | tail -2
