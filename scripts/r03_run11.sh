#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03k
export TMPDIR=/tmp
timeout 300 python scripts/dbg/hs_phases.py 23600 256 256 > gpurun_out/r03k/phases_256.log 2>&1
grep -v "^$" gpurun_out/r03k/phases_256.log | tail -7
PHC_GEMM_HS_ALL=1 timeout 300 python scripts/dbg/hs_phases.py 13824 128 128 > gpurun_out/r03k/phases_128.log 2>&1
grep -v "^$" gpurun_out/r03k/phases_128.log | tail -7
timeout 900 python -m pytest tests/test_gpu_gemm_hs.py -x -q > gpurun_out/r03k/hs_tests.log 2>&1
echo "hs tests rc=$?"
tail -3 gpurun_out/r03k/hs_tests.log
for wl in zinc_b1024 molpcba_b1024; do
  timeout 600 python bench.py --workload $wl --cpu-steps 0 > gpurun_out/r03k/bench_$wl.log 2>&1
  tail -1 gpurun_out/r03k/bench_$wl.log | cut -c1-300
done
