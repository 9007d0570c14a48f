#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03z; mkdir -p $O
for x in 0 1 2 16 4 8 15; do
  export PHC_HS_EXP=$x
  OUT=$GRAFT_REPO_ROOT/$O/prof_exp$x
  rm -rf $OUT && mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --workload zinc_b1024 --steps 30 --warmup 5 --cpu-steps 0 --min-time 0 --profile-steps 0 > $OUT/bench.log 2>&1
  python3 scripts/summarize_rocprof.py $(ls -t $OUT/*/*kernel_stats.csv | head -1) 35 70 > $OUT/summary.txt
  echo "exp=$x $(grep '^{"metric"' $OUT/bench.log | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")"; grep -E "gemm_hs16" $OUT/summary.txt | cut -c1-120
  find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
done
