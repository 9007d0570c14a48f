#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03g
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_gemm_hs.py -x -q > gpurun_out/r03g/hs_tests.log 2>&1
echo "hs tests rc=$?"
tail -8 gpurun_out/r03g/hs_tests.log
for wl in zinc_b1024 molpcba_b1024; do
  timeout 600 python bench.py --workload $wl --cpu-steps 0 > gpurun_out/r03g/bench_$wl.log 2>&1
  tail -1 gpurun_out/r03g/bench_$wl.log | cut -c1-330
done
PHC_MLP_NORM_FUSED_MIN_WORK=0 timeout 600 python bench.py --workload molhiv_b512 --cpu-steps 0 > gpurun_out/r03g/bench_molhiv_fused.log 2>&1
tail -1 gpurun_out/r03g/bench_molhiv_fused.log | cut -c1-330
timeout 600 python bench.py --workload molhiv_b512 --cpu-steps 0 > gpurun_out/r03g/bench_molhiv.log 2>&1
tail -1 gpurun_out/r03g/bench_molhiv.log | cut -c1-330
timeout 2400 python -m pytest tests -m gpu -q --deselect tests/test_gpu_gemm_hs.py > gpurun_out/r03g/gpu_suite.log 2>&1
echo "suite rc=$?"
tail -12 gpurun_out/r03g/gpu_suite.log
