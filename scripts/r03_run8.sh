#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03h
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_gemm_hs.py -x -q > gpurun_out/r03h/hs_tests.log 2>&1
echo "hs tests rc=$?"
tail -3 gpurun_out/r03h/hs_tests.log
for wl in zinc_b1024 molpcba_b1024 molhiv_b512; do
  timeout 600 python bench.py --workload $wl --cpu-steps 0 > gpurun_out/r03h/bench_$wl.log 2>&1
  tail -1 gpurun_out/r03h/bench_$wl.log | cut -c1-330
done
PHC_GEMM_HS_ALL=1 PHC_MLP_NORM_FUSED_MIN_WORK=0 timeout 600 python bench.py --workload molhiv_b512 --cpu-steps 0 > gpurun_out/r03h/bench_molhiv_hsall.log 2>&1
tail -1 gpurun_out/r03h/bench_molhiv_hsall.log | cut -c1-330
timeout 600 python bench.py --workload synth_10m50m --steps 5 --warmup 2 > gpurun_out/r03h/bench_synth.log 2>&1
tail -1 gpurun_out/r03h/bench_synth.log | cut -c1-600
