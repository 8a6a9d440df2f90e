#!/bin/bash
set -u
for v in "" fw7 fw6 "" fw7 fw6; do
  s=${v:+_$v}
  HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat$s.so timeout 300 python scripts/ab_render.py --iters 30 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['lib'], 'c3 fwd', d['render_fwd_ms'], d['render_fwd_med'], 'bwd', d['render_bwd_ms'], 'step', d['step_ms'], d['step_med'])"
done
for v in "" fw7; do
  s=${v:+_$v}
  for c in c2 c4; do
  HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat$s.so timeout 300 python scripts/ab_render.py --iters 10 --config $c 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['lib'], '$c fwd', d['render_fwd_ms'], 'step', d['step_ms'], d['step_med'])"
  done
done
