#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03c
export TMPDIR=/tmp
timeout 300 ./scripts/exp/variants/gemm_rs_probe > gpurun_out/r03c/probe.log 2>&1
echo "probe rc=$?"
cat gpurun_out/r03c/probe.log
timeout 600 python scripts/dbg/config5_dbg.py > gpurun_out/r03c/c5dbg.log 2>&1
echo "c5dbg rc=$?"
grep -v "^$" gpurun_out/r03c/c5dbg.log | tail -25
timeout 600 python scripts/dbg/config5_dbg.py 1000000 5000000 > gpurun_out/r03c/c5dbg_small.log 2>&1
grep -v "^$" gpurun_out/r03c/c5dbg_small.log | tail -14
timeout 900 python -m pytest tests/test_gpu_two_rank_engine.py -q -s -k relu > gpurun_out/r03c/two_rank.log 2>&1
echo "two rank rc=$?"
tail -12 gpurun_out/r03c/two_rank.log
