#!/bin/bash
# round 3, first GPU call: the new parity tests, the whole GPU suite, baseline bench lines of this box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03a
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_two_rank_engine.py tests/test_gpu_config5_fullsize.py -x -q -s > gpurun_out/r03a/new_tests.log 2>&1
echo "new tests rc=$?" | tee -a gpurun_out/r03a/new_tests.log
timeout 2400 python -m pytest tests -m gpu -q --deselect tests/test_gpu_two_rank_engine.py --deselect tests/test_gpu_config5_fullsize.py > gpurun_out/r03a/gpu_suite.log 2>&1
echo "suite rc=$?" | tee -a gpurun_out/r03a/gpu_suite.log
tail -15 gpurun_out/r03a/gpu_suite.log
for wl in molhiv_b512 zinc_b1024 molpcba_b1024 molhiv_b512_softmax; do
  timeout 600 python bench.py --workload $wl --cpu-steps 0 > gpurun_out/r03a/bench_$wl.log 2>&1
  tail -1 gpurun_out/r03a/bench_$wl.log | cut -c1-400
done
tail -30 gpurun_out/r03a/new_tests.log
