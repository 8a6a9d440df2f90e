#!/bin/bash
# round-3 GPU batch 5: two lane groups with plain-store planes -- parity subset, then A/B
set -u
O=gpurun_out/r3e; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "golden or ldr_forward or motion_blur or edge_cases or determinism or replays or 14400 or render_stats or inverse_depth or all_optional or randomized" > $O/tests.log 2>&1; echo "pytest rc $?"
grep -E "passed|failed" $O/tests.log | tail -3
grep -E "^FAILED|^ERROR|^E  " $O/tests.log | head -40
for v in onelist "" s96 s128 onelist ""; do
  s=${v:+_$v}
  HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat$s.so timeout 300 python scripts/ab_render.py --iters 20 --stats 2>/dev/null | tail -1
done
for v in onelist ""; do
  s=${v:+_$v}
  HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat$s.so timeout 300 python scripts/ab_render.py --iters 10 --config c4 2>/dev/null | tail -1
  HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat$s.so timeout 300 python scripts/ab_render.py --iters 10 --config c2 2>/dev/null | tail -1
done
HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat_gs.so timeout 300 python scripts/ab_render.py --iters 20 2>/dev/null | tail -1
HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat_gs.so bash scripts/kstats.sh --capacity 8500000 2>&1 | grep -E "render_fwd|gather_sorted"
