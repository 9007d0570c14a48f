#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for wl in molhiv_pub_phc4_2x200 zinc_pub_phc5_4x200 zinc_pub_phc4_4x180 molhiv_b512_softmax; do
  python bench.py --workload $wl --cpu-steps 0 2>&1 | grep '^{"metric"' | tail -1 > gpurun_out/r03_bench_$wl.json
done
bash scripts/rocprof_trace.sh r03b 1800 > gpurun_out/r03_rocprof_trace.log 2>&1
python bench.py 2>&1 | tail -1 > gpurun_out/r03_bench_default.json
