#!/bin/bash
# round-3 GPU batch 4: two lane groups in the render backward -- parity, then A/B against the one-list kernel
set -u
O=gpurun_out/r3d; mkdir -p $O
HS_PARITY_REPORT=1 timeout 2400 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/tests.log 2>&1; echo "pytest rc $?"
grep -E "passed|failed" $O/tests.log | tail -3
grep -E "^FAILED|^ERROR|^E  " $O/tests.log | head -40
for v in onelist "" g96 onelist "" g96; do
  s=${v:+_$v}
  HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat$s.so timeout 300 python scripts/ab_render.py --iters 20 --stats 2>/dev/null | tail -1
done
for v in onelist "" g96; do
  s=${v:+_$v}
  HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat$s.so timeout 300 python scripts/ab_render.py --iters 10 --config c4 2>/dev/null | tail -1
  HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat$s.so timeout 300 python scripts/ab_render.py --iters 10 --config c2 2>/dev/null | tail -1
done
for v in b0 b2 "" b0 b2; do
  s=${v:+_$v}
  HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat$s.so timeout 300 python scripts/ab_render.py --iters 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['lib'], 'c3 binning', d['binning_ms'])"
done
