#!/bin/bash
set -u
O=gpurun_out/r3j; mkdir -p $O
for s in 30 31 33 35 36 37 38 39; do
HS_SWEEP_SEED=$s HS_SWEEP_CASES=300 timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k randomized_configurations -p no:cacheprovider > $O/soak_$s.log 2>&1
echo "seed $s: $(grep -E "^E  |passed|failed" $O/soak_$s.log | cut -c1-400 | head -3 | tr '\n' ' ')"
done
python scripts/parity_table.py --big > $O/parity_table.log 2>&1; tail -40 $O/parity_table.log | cut -c1-300
python scripts/wild_c2.py 2>&1 | tail -6 | cut -c1-400
