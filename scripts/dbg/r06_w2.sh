cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_mlp_bwd_fused.py -x -q 2>&1 | tail -15
for i in 1 2 3; do
for v in 1 0; do
PHC_MLP_BWD_FUSED_AGG=$v python3 bench.py --cpu-steps 0 --profile-steps 0 --steps 300 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('agg=$v', d['ms_per_step'])"
done; done
