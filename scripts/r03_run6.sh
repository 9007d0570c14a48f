#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03f
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_gemm_hs.py -x -q > gpurun_out/r03f/hs_tests.log 2>&1
echo "hs tests rc=$?"
tail -25 gpurun_out/r03f/hs_tests.log
for wl in molhiv_b512 zinc_b1024 molpcba_b1024; do
  timeout 600 python bench.py --workload $wl --cpu-steps 0 > gpurun_out/r03f/bench_$wl.log 2>&1
  tail -1 gpurun_out/r03f/bench_$wl.log | cut -c1-330
  PHC_MLP_NORM_FUSED=0 timeout 600 python bench.py --workload $wl --cpu-steps 0 > gpurun_out/r03f/bench_${wl}_nofuse.log 2>&1
  tail -1 gpurun_out/r03f/bench_${wl}_nofuse.log | cut -c1-330
done
timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_gemm_hs.py > gpurun_out/r03f/gpu_suite.log 2>&1
echo "suite rc=$?"
tail -12 gpurun_out/r03f/gpu_suite.log
