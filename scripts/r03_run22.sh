#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03v; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_stat_accumulators.py tests/test_gpu_ops.py tests/test_gpu_fullwidth.py tests/test_gpu_model.py tests/test_gpu_dropout.py -m gpu -q -x > $O/tests.log 2>&1
echo "tests rc=$?"; grep -E "passed|failed|Error|error" $O/tests.log | tail -10 | cut -c1-300
for wl in molhiv_pub_phc4_2x200 zinc_pub_phc5_4x200 zinc_pub_phc4_4x180 molhiv_b512 zinc_b1024; do
  timeout 600 python bench.py --workload $wl --cpu-steps 0 2>&1 | grep '^{"metric"' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$wl', d['ms_per_step'])"
done
