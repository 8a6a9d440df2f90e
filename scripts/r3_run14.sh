#!/bin/bash
set -u
for v in "" t4 t12 t16 k52w7 k48 ""; do
  s=${v:+_$v}
  HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat$s.so timeout 300 python scripts/ab_render.py --iters 30 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['lib'], 'c3 bwd', d['render_bwd_ms'], d['render_bwd_med'], 'fwd', d['render_fwd_ms'], 'step', d['step_ms'], d['step_med'])"
done
for v in "" k52w7; do
  s=${v:+_$v}
  for c in c2 c4; do
  HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat$s.so timeout 300 python scripts/ab_render.py --iters 10 --config $c 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['lib'], '$c bwd', d['render_bwd_ms'], 'step', d['step_ms'], d['step_med'])"
  done
done
HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat_k52w7.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "golden or determinism or replays or ldr_forward" 2>&1 | tail -2
