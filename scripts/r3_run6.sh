#!/bin/bash
set -u
O=gpurun_out/r3f; mkdir -p $O
HS_PARITY_REPORT=1 timeout 2400 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/tests.log 2>&1; echo "pytest rc $?"
grep -E "passed|failed" $O/tests.log | tail -3
grep -E "^FAILED|^ERROR|^E  " $O/tests.log | head -40
for i in 1 2; do timeout 300 python scripts/ab_render.py --iters 20 --stats 2>/dev/null | tail -1; done
timeout 900 python bench.py --steps 50 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3f/bench_c3.json'))
print('c3', round(d['value'],1), 'img/s', round(d['ms_per_step'],4), 'ms', d['stages_ms'])
r=d['roofline']; print('frac', r['frac'], 'fwd_bwd', r['fwd_bwd']['frac'], 'whole', r['whole_step']['frac'], 'valu bwd', r['valu']['bwd'].get('frac'), r['valu']['bwd'].get('lane_utilisation'))
print(d['cpu_baseline'].get('value'), d['cpu_baseline'].get('c_oracle_single_thread',{}).get('value'))
PY
bash scripts/soak.sh 150 21 22 2>&1 | tail -4
