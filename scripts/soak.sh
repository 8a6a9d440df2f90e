#!/bin/bash
# Soak of the randomized oracle sweep on the GPU box: bash scripts/soak.sh [cases-per-seed] [seed ...]
N=${1:-300}; shift
for s in ${@:-11 12 13}; do
  HS_SWEEP_SEED=$s HS_SWEEP_CASES=$N timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k randomized_configurations 2>&1 | tail -3 | sed "s/^/seed $s: /"
done
