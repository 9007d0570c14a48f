#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03b
export TMPDIR=/tmp
timeout 300 ./scripts/exp/variants/gemm_rs_probe > gpurun_out/r03b/probe.log 2>&1
echo "probe rc=$?"
cat gpurun_out/r03b/probe.log
timeout 900 python -m pytest tests/test_gpu_two_rank_engine.py tests/test_gpu_config5_fullsize.py -q -s > gpurun_out/r03b/new_tests.log 2>&1
echo "new tests rc=$?"
tail -25 gpurun_out/r03b/new_tests.log
