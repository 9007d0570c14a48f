cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_mlp_fwd_fused.py tests/test_gpu_ops.py tests/test_gpu_step_fusions.py -x -q -k "softmax or Softmax" 2>&1 | tail -4
for i in 1 2 3; do
for v in 1 0; do
PHC_SOFTMAX_INTO_MLP=$v python3 bench.py --workload molhiv_b512_softmax --cpu-steps 0 --profile-steps 0 --steps 300 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('softmax_into_mlp=$v', d['ms_per_step'])"
done; done
python3 bench.py --workload molhiv_pub_phc4_2x200 --cpu-steps 0 --profile-steps 0 --steps 300 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('2x200', d['ms_per_step'])"
bash scripts/dbg/prof_wl.sh molhiv_b512_softmax r06s 2>&1 | grep -E "period|mlp_fwd|softmax_agg"
