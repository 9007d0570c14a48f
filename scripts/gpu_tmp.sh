#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 2700 python -m pytest tests -m gpu -q -x > gpurun_out/r03_gpu_suite8.log 2>&1
echo "suite rc=$?"; grep -E "passed|failed|^E  " gpurun_out/r03_gpu_suite8.log | tail -5 | cut -c1-250
for wl in molhiv_pub_phc4_2x200 zinc_pub_phc5_4x200 molhiv_b512; do
  timeout 600 python bench.py --workload $wl --cpu-steps 0 --profile-steps 0 2>&1 | grep '^{"metric"' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$wl', d['ms_per_step'])"
done
