#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03n
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_two_rank_engine.py tests/test_gpu_config5_fullsize.py -m gpu -q -k "data_parallel or two_rank or config5 or sharded or halo" > $O/dp_tests.log 2>&1
echo "dp tests rc=$?"; tail -5 $O/dp_tests.log
timeout 900 python bench.py > $O/bench_default.log 2>&1
echo "bench rc=$?"; tail -1 $O/bench_default.log | cut -c1-1500
PHC_BENCH_FORCE_SYNC=1 timeout 600 python bench.py --cpu-steps 0 > $O/bench_sync.log 2>&1
tail -1 $O/bench_sync.log | cut -c1-300
timeout 900 python bench.py --workload synth_10m50m --cpu-steps 0 > $O/bench_synth.log 2>&1
tail -1 $O/bench_synth.log | cut -c1-900
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_synth -- python3 $GRAFT_REPO_ROOT/bench.py --workload synth_10m50m --cpu-steps 0 --steps 5 --warmup 2 --min-time 0 > $GRAFT_REPO_ROOT/$O/prof_synth.log 2>&1)
find $O/prof_synth -name "*kernel_stats.csv" | head -1 | xargs -r head -25
find $O/prof_synth -name "*.db" -delete; find $O/prof_synth -name "*kernel_trace.csv" -delete
