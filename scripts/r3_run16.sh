#!/bin/bash
set -u
O=gpurun_out/r3p; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/tests.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed" $O/tests.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash scripts/collect_profiles.sh all c3 > $O/collect_c3.log 2>&1; tail -3 $O/collect_c3.log
bash scripts/collect_profiles.sh all c2 > $O/collect_c2.log 2>&1; tail -2 $O/collect_c2.log
bash scripts/collect_profiles.sh all c4 > $O/collect_c4.log 2>&1; tail -2 $O/collect_c4.log
du -sh gpurun_out/prof_final gpurun_out/prof_final_c2 gpurun_out/prof_final_c4
