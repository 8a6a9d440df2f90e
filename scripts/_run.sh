#!/bin/bash
cd $GRAFT_REPO_ROOT
bash scripts/collect_profiles.sh all c3 > gpurun_out/collect_c3.log 2>&1
bash scripts/collect_profiles.sh all c2 > gpurun_out/collect_c2.log 2>&1
bash scripts/collect_profiles.sh all c4 > gpurun_out/collect_c4.log 2>&1
tail -3 gpurun_out/collect_c*.log
