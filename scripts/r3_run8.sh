#!/bin/bash
set -u
O=gpurun_out/r3h; mkdir -p $O
python scripts/sweep_case.py 24 9 2>&1 | tail -30
for s in 22 24 25 26 27 28; do
HS_SWEEP_SEED=$s HS_SWEEP_CASES=200 timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k randomized_configurations -p no:cacheprovider > $O/soak_$s.log 2>&1
grep -E "^E  |passed|failed" $O/soak_$s.log | cut -c1-500 | head -6
done
