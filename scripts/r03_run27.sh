#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03z2; mkdir -p $O
for x in 1 2 1 2; do
  export PHC_HS_RB=$x
  timeout 600 python bench.py --workload zinc_b1024 --cpu-steps 0 --profile-steps 0 2>&1 | grep '^{"metric"' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('rb=$x', d['ms_per_step'])"
done
export PHC_HS_RB=2
OUT=$GRAFT_REPO_ROOT/$O/prof_rb2
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --workload zinc_b1024 --steps 30 --warmup 5 --cpu-steps 0 --min-time 0 --profile-steps 0 > $OUT/bench.log 2>&1
python3 scripts/summarize_rocprof.py $(ls -t $OUT/*/*kernel_stats.csv | head -1) 35 70 > $OUT/summary.txt
grep -E "gemm_hs16" $OUT/summary.txt | cut -c1-120
timeout 900 python -m pytest tests/test_gpu_gemm_hs.py -q -x 2>&1 | tail -2
