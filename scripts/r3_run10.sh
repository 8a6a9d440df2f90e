#!/bin/bash
set -u
O=gpurun_out/r3i; mkdir -p $O
for s in 22 24 29 30 31 32 33 34; do
HS_SWEEP_SEED=$s HS_SWEEP_CASES=300 timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k randomized_configurations -p no:cacheprovider > $O/soak_$s.log 2>&1
echo "seed $s: $(grep -E "^E  |passed|failed" $O/soak_$s.log | cut -c1-400 | head -3 | tr '\n' ' ')"
done
