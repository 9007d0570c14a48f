cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_step_fusions.py tests/test_gpu_head_chain.py -x -q 2>&1 | tail -3
for i in 1 2 3; do
for v in 0 1; do
HIP_FORCE_DEV_KERNARG=$v python3 bench.py --cpu-steps 0 --profile-steps 0 --steps 300 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('dev_kernarg=$v', d['ms_per_step'])"
done; done
