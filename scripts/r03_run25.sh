#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03y; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_stat_accumulators.py tests/test_gpu_packet_capture.py -m gpu -q -x > $O/tests.log 2>&1
echo "tests rc=$?"; grep -E "passed|failed|Error|error" $O/tests.log | tail -10 | cut -c1-300
for i in 1 2 3; do
for v in 0 1; do
  if [ $v = 1 ]; then export PHC_CSR_PAIR_NO_FRONT=1; else unset PHC_CSR_PAIR_NO_FRONT; fi
  timeout 600 python bench.py --workload molhiv_b512 --cpu-steps 0 --profile-steps 0 2>&1 | grep '^{"metric"' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('nofront=$v', d['ms_per_step'])"
done; done
unset PHC_CSR_PAIR_NO_FRONT
for wl in zinc_b1024 molpcba_b1024 molhiv_b512_softmax; do
  timeout 600 python bench.py --workload $wl --cpu-steps 0 --profile-steps 0 2>&1 | grep '^{"metric"' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$wl', d['ms_per_step'])"
done
