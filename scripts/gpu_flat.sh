#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03flat; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullwidth.py tests/test_gpu_stat_accumulators.py -m gpu -q -x -k "not_a_multiple or g12 or pub or encode" > $O/tests.log 2>&1
echo "tests rc=$?"; grep -E "passed|failed|^E  " $O/tests.log | tail -12 | cut -c1-300
for wl in molhiv_pub_phc4_2x200 zinc_pub_phc5_4x200 zinc_pub_phc4_4x180; do
  for fl in 1 0; do
  PHC_FLAT_TABLES=$fl timeout 600 python bench.py --workload $wl --cpu-steps 0 --profile-steps 0 2>&1 | grep '^{"metric"' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$wl flat=$fl', d['ms_per_step'])"
  done
done
