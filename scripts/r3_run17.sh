#!/bin/bash
set -u
O=gpurun_out/r3q; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/tests.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed" $O/tests.log | tail -2
grep -E "^FAILED|^E  " $O/tests.log | cut -c1-300 | head
bash scripts/collect_profiles.sh all c3 > $O/collect_c3.log 2>&1; tail -2 $O/collect_c3.log
bash scripts/collect_profiles.sh all c2 > $O/collect_c2.log 2>&1; tail -1 $O/collect_c2.log
bash scripts/collect_profiles.sh all c4 > $O/collect_c4.log 2>&1; tail -1 $O/collect_c4.log
