#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -q > gpurun_out/r03_gpu_suite_final.log 2>&1
echo "suite rc=$?"; grep -E "passed|failed" gpurun_out/r03_gpu_suite_final.log | tail -2
bash scripts/round_profiles.sh r03 > gpurun_out/r03_round_profiles.log 2>&1
ls gpurun_out/r03_* | wc -l
