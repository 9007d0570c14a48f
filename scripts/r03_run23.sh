#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03w; mkdir -p $O
for wl in ${WLS:-molhiv_pub_phc4_2x200 zinc_pub_phc5_4x200 zinc_pub_phc4_4x180}; do
  OUT=$GRAFT_REPO_ROOT/$O/prof_$wl
  rm -rf $OUT && mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --workload $wl --steps 50 --warmup 10 --cpu-steps 0 --min-time 0 --profile-steps 0 > $OUT/bench.log 2>&1
  grep '^{"metric"' $OUT/bench.log | cut -c1-200
  python3 scripts/summarize_rocprof.py $(ls -t $OUT/*/*kernel_stats.csv | head -1) 60 70 > $OUT/summary.txt
  head -${TOPN:-34} $OUT/summary.txt | cut -c1-160
  find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
done
