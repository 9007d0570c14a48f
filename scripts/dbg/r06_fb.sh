cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_head_loss.py -x -q 2>&1 | tail -4
for i in 1 2 3 4 5 6; do
for v in 1 0; do
PHC_HEAD_FUSED_BWD=$v python3 bench.py --cpu-steps 0 --profile-steps 0 --steps 600 --min-time 8 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fb=$v', d['ms_per_step'])"
done; done
