#!/bin/bash
# round-3 GPU batch 3: validate un-ticketed default + bounded wait, graph step, per-kernel stats
set -u
O=gpurun_out/r3c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "hip_graph or bare_multi or ticketed or radix or look_back or overflow or fixed_capacity" > $O/tests.log 2>&1; echo "pytest rc $?"
grep -E "passed|failed" $O/tests.log | tail -3
grep -E "^FAILED|^ERROR|^E  " $O/tests.log | head -30
HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat.so timeout 300 python scripts/ab_render.py --iters 20 2>/dev/null | tail -1
bash scripts/kstats.sh --capacity 8500000 2>&1 | grep -E "radix|emit|gather|ghist|tile_ranges|scan|preprocess|segsum|order" 
for c in c2 c3 c4; do
  timeout 600 python -X faulthandler bench.py --config $c --steps 50 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_$c.json 2> $O/bench_$c.err; echo "bench $c rc $?"; tail -3 $O/bench_$c.err
  timeout 600 python bench.py --config $c --steps 50 --warmup 5 --no-cpu-baseline --no-extras --graph off > $O/bench_${c}_eager.json 2> $O/bench_${c}_eager.err
done
python - <<'PY'
import json
for n in ("c2","c2_eager","c3","c3_eager","c4","c4_eager"):
    try:
        d=json.load(open(f'gpurun_out/r3c/bench_{n}.json'))
        print(n, round(d['value'],1), 'img/s', round(d['ms_per_step'],4), 'ms', d['config']['launch'][:12], d['stages_ms'])
    except Exception as e: print(n, 'parse failed', e)
PY
