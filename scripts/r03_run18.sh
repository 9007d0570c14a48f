#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03r; mkdir -p $O
for x in 0 1 2 4 6; do
  export PHC_SM_EXP=$x
  OUT=$GRAFT_REPO_ROOT/$O/prof_exp$x
  rm -rf $OUT && mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --workload molhiv_b512_softmax --steps 50 --warmup 10 --cpu-steps 0 --min-time 0 --profile-steps 0 > $OUT/bench.log 2>&1
  python3 scripts/summarize_rocprof.py $(ls -t $OUT/*/*kernel_stats.csv | head -1) 60 70 > $OUT/summary.txt
  echo "exp=$x"; grep -E "softmax_agg|enc_direct" $OUT/summary.txt | cut -c1-200
  find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
done
