cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_mlp_fwd_fused.py -x -q 2>&1 | tail -2
PYTHONPATH=. python scripts/dbg/mlp_all_wg_phases.py | grep -E "^arrive|^release|^end|blocks"
for i in 1 2 3 4; do
  for which in base r6c; do
    if [ $which = base ]; then unset PHC_LIB_PATH; else export PHC_LIB_PATH=$GRAFT_REPO_ROOT/scripts/exp/variants/libphc_hip_$which.so; fi
    python3 bench.py --cpu-steps 0 --profile-steps 0 --steps 600 --min-time 8 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$which', d['ms_per_step'])"
  done
done
