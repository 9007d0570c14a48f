#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1 2 3; do for v in 0 1; do
  if [ $v = 1 ]; then export PHC_GEMM_HS_ALL=1; else unset PHC_GEMM_HS_ALL; fi
  timeout 600 python bench.py --workload molhiv_b512 --cpu-steps 0 --profile-steps 0 2>&1 | grep '^{"metric"' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('hs_all=$v', d['ms_per_step'])"
done; done
