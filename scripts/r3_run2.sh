#!/bin/bash
# round-3 GPU batch 2: ticket / bound A/B of the radix passes, per-kernel stats, graph step, preprocess staging
set -u
O=gpurun_out/r3b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "hip_graph or bare_multi or golden or ldr_forward or radix or 14400 or look_back" > $O/tests.log 2>&1; echo "pytest rc $?"
grep -E "passed|failed" $O/tests.log | tail -3
grep -E "^FAILED|^ERROR|^E  " $O/tests.log | head -30
for v in "" nt tl nb ntnb ""; do
  s=${v:+_$v}
  HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat$s.so timeout 300 python scripts/ab_render.py --iters 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['lib'], 'c3 binning', d['binning_ms'], 'segsum+pre_bwd', d['segsum_pre_bwd_ms'], 'bwd', d['render_bwd_ms'], 'fwd', d['render_fwd_ms'], 'step', d['step_ms'], d['step_med'])"
done
bash scripts/kstats.sh --capacity 8500000 2>&1 | tail -30
for c in c2 c3; do
  timeout 600 python bench.py --config $c --steps 50 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_$c.json 2> $O/bench_$c.err; echo "bench $c rc $?"; tail -2 $O/bench_$c.err
  timeout 600 python bench.py --config $c --steps 50 --warmup 5 --no-cpu-baseline --no-extras --graph off > $O/bench_${c}_eager.json 2> $O/bench_${c}_eager.err
done
python - <<'PY'
import json
for n in ("c2","c2_eager","c3","c3_eager"):
    try:
        d=json.load(open(f'gpurun_out/r3b/bench_{n}.json'))
        print(n, round(d['value'],1), 'img/s', round(d['ms_per_step'],4), 'ms', d['config']['launch'][:12], d['stages_ms'], 'sum', round(sum(v for k,v in d['stages_ms'].items() if k!='binning'),4))
    except Exception as e: print(n, 'parse failed', e)
PY
