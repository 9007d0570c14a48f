#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03p; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -k "softmax" > $O/softmax_tests.log 2>&1
echo "softmax tests rc=$?"; tail -4 $O/softmax_tests.log
for wl in molhiv_b512_softmax; do
  OUT=$GRAFT_REPO_ROOT/$O/prof_$wl
  rm -rf $OUT && mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --workload $wl --steps 50 --warmup 10 --cpu-steps 0 --min-time 0 --profile-steps 0 > $OUT/bench.log 2>&1
  python3 scripts/summarize_rocprof.py $(ls -t $OUT/*/*kernel_stats.csv | head -1) 60 70 > $OUT/summary.txt
  head -12 $OUT/summary.txt | cut -c1-200
  find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
done
timeout 600 python bench.py --workload molhiv_b512_softmax --cpu-steps 0 2>&1 | tail -1 | cut -c1-260
timeout 600 python bench.py --workload molhiv_b512 --cpu-steps 0 2>&1 | tail -1 | cut -c1-260
