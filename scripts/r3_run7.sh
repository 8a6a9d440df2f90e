#!/bin/bash
set -u
O=gpurun_out/r3g; mkdir -p $O
for s in 22 23 24 25; do
HS_SWEEP_SEED=$s HS_SWEEP_CASES=150 timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k randomized_configurations -p no:cacheprovider > $O/soak_$s.log 2>&1
grep -E "^E  |passed|failed" $O/soak_$s.log | cut -c1-600 | head -12
done
