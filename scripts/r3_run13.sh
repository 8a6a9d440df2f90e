#!/bin/bash
set -u
O=gpurun_out/r3m; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "radix or look_back or ticketed or ldr_forward or 14400 or c3 or c4 or fixed_capacity" > $O/tests.log 2>&1; echo "pytest rc $?"
grep -E "passed|failed" $O/tests.log | tail -3
grep -E "^FAILED|^ERROR|^E  " $O/tests.log | cut -c1-300 | head -20
for v in fwd1 "" fwd1 ""; do
  s=${v:+_$v}
  HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat$s.so timeout 300 python scripts/ab_render.py --iters 30 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['lib'], 'c3 binning', d['binning_ms'], 'step', d['step_ms'], d['step_med'])"
done
for v in fwd1 ""; do
  s=${v:+_$v}
  for c in c2 c4; do
  HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat$s.so timeout 300 python scripts/ab_render.py --iters 10 --config $c 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['lib'], '$c binning', d['binning_ms'], 'step', d['step_ms'], d['step_med'])"
  done
done
bash scripts/kstats.sh --capacity 8500000 2>&1 | grep -E "radix|emit|gather|ghist|tile_ranges|scan|preprocess|segsum|order|render|crf"
