#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03m
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r03m/gpu_suite.log 2>&1
echo "suite rc=$?"
tail -8 gpurun_out/r03m/gpu_suite.log
for wl in zinc_b1024 molpcba_b1024 molhiv_b512; do
  timeout 600 python bench.py --workload $wl --cpu-steps 0 > gpurun_out/r03m/bench_$wl.log 2>&1
  tail -1 gpurun_out/r03m/bench_$wl.log | cut -c1-300
done
