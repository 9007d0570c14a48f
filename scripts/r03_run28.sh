#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03aa; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_model.py tests/test_gpu_dropout.py tests/test_gpu_ops.py -m gpu -q > $O/tests.log 2>&1
echo "tests rc=$?"; grep -E "passed|failed|^E  " $O/tests.log | tail -12 | cut -c1-300
for i in 1 2; do for wl in molhiv_b512 molhiv_b512_softmax; do
  timeout 600 python bench.py --workload $wl --cpu-steps 0 --profile-steps 0 2>&1 | grep '^{"metric"' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$wl', d['ms_per_step'])"
done; done
