#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 2700 python -m pytest tests -m gpu -q > gpurun_out/r03_gpu_suite2.log 2>&1
echo "suite rc=$?"; grep -E "passed|failed" gpurun_out/r03_gpu_suite2.log | tail -3
