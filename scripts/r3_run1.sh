#!/bin/bash
# round-3 GPU batch 1: full GPU test suite (parity report on), c3 bench, binning tuning variants
set -u
O=gpurun_out/r3a; mkdir -p $O
HS_PARITY_REPORT=1 timeout 2400 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/tests.log 2>&1; echo "pytest rc $?"
grep -E "passed|failed" $O/tests.log | tail -3
grep -E "^FAILED|^ERROR|Error|assert " $O/tests.log | head -40
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench rc $?"
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/r3a/bench_c3.json'))
    print('c3', round(d['value'],1), 'img/s', round(d['ms_per_step'],4), 'ms', d['stages_ms'])
except Exception as e: print('bench parse failed', e)
PY
for v in "" d8 dl16 dl32 d8l16 d4l16 ""; do
  s=${v:+_$v}
  HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat$s.so timeout 300 python scripts/ab_render.py --iters 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['lib'], 'c3 binning', d['binning_ms'], 'segsum+pre_bwd', d['segsum_pre_bwd_ms'], 'bwd', d['render_bwd_ms'], 'fwd', d['render_fwd_ms'], 'step', d['step_ms'], d['step_med'])"
done
for v in "" d8l16 d4l16; do
  s=${v:+_$v}
  for c in c2 c4; do
  HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat$s.so timeout 300 python scripts/ab_render.py --iters 10 --config $c 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['lib'], '$c binning', d['binning_ms'], 'step', d['step_ms'], d['step_med'])"
  done
done
