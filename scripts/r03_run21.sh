#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03u; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_fullwidth.py -m gpu -q -k "g12 or pub" > $O/tests.log 2>&1
echo "tests rc=$?"; grep -E "passed|failed|Error|error" $O/tests.log | tail -30 | cut -c1-300
for wl in molhiv_pub_phc4_2x200 zinc_pub_phc5_4x200 zinc_pub_phc4_4x180; do
  OUT=$GRAFT_REPO_ROOT/$O/prof_$wl
  rm -rf $OUT && mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --workload $wl --steps 50 --warmup 10 --cpu-steps 0 --min-time 0 --profile-steps 0 > $OUT/bench.log 2>&1
  tail -1 $OUT/bench.log | cut -c1-200
  python3 scripts/summarize_rocprof.py $(ls -t $OUT/*/*kernel_stats.csv | head -1) 60 70 > $OUT/summary.txt
  head -30 $OUT/summary.txt | cut -c1-160
  find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
done
