#!/bin/bash
# kernel-stat profiles of the bench workloads (graph-mode replays are traced kernel by kernel)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for wl in zinc_b1024 molpcba_b1024 molhiv_b512 molhiv_b512_softmax; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/r03o/prof_$wl
  rm -rf $OUT && mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --workload $wl --steps 50 --warmup 10 --cpu-steps 0 --min-time 0 --profile-steps 0 > $OUT/bench.log 2>&1
  tail -1 $OUT/bench.log | cut -c1-160
  python3 scripts/summarize_rocprof.py $(ls -t $OUT/*/*kernel_stats.csv | head -1) 60 70 > $OUT/summary.txt
  head -60 $OUT/summary.txt | cut -c1-200
  find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
done
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03o/prof_synth_10m50m
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --workload synth_10m50m --cpu-steps 0 --steps 8 --warmup 2 --min-time 0 > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-160
python3 scripts/summarize_rocprof.py $(ls -t $OUT/*/*kernel_stats.csv | head -1) 10 40 > $OUT/summary.txt
head -40 $OUT/summary.txt | cut -c1-200
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
