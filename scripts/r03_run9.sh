#!/bin/bash
# rocprofv3 kernel stats for molpcba and zinc (eager mode would distort; graph mode kernels are still traced)
cd /tmp && export TMPDIR=/tmp
for wl in molpcba_b1024 zinc_b1024; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/r03i/prof_$wl
  rm -rf $OUT && mkdir -p $OUT
  cd $GRAFT_REPO_ROOT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --workload $wl --steps 30 --warmup 5 --cpu-steps 0 --min-time 0 > $OUT/bench.log 2>&1
  tail -1 $OUT/bench.log | cut -c1-200
  f=$(find $OUT -name "*kernel_stats.csv" | head -1)
  python3 scripts/summarize_rocprof.py $f 1 40 > $OUT/summary.txt
  head -50 $OUT/summary.txt
  find $OUT -name "*kernel_trace.csv" -size +5M -delete
  find $OUT -name "*.db" -delete
done
