#!/bin/bash
# round-3 GPU batch 12: two lane groups in the render FORWARD too -- parity, then A/B against the one-list forward
set -u
O=gpurun_out/r3l; mkdir -p $O
HS_PARITY_REPORT=1 timeout 2400 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/tests.log 2>&1; echo "pytest rc $?"
grep -E "passed|failed" $O/tests.log | tail -3
grep -E "^FAILED|^ERROR|^E  " $O/tests.log | cut -c1-300 | head -40
for v in fwd1 "" fwd1 ""; do
  s=${v:+_$v}
  HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat$s.so timeout 300 python scripts/ab_render.py --iters 20 --stats 2>/dev/null | tail -1
done
for v in fwd1 ""; do
  s=${v:+_$v}
  HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat$s.so timeout 300 python scripts/ab_render.py --iters 10 --config c4 2>/dev/null | tail -1
  HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat$s.so timeout 300 python scripts/ab_render.py --iters 10 --config c2 2>/dev/null | tail -1
done
for s in 41 42; do
HS_SWEEP_SEED=$s HS_SWEEP_CASES=300 timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k randomized_configurations -p no:cacheprovider > $O/soak_$s.log 2>&1
echo "seed $s: $(grep -E "^E  |passed|failed" $O/soak_$s.log | cut -c1-400 | head -3 | tr '\n' ' ')"
done
