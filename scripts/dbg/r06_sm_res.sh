# softmax backward with the skips' shares inside the walk: A/B on one box (PHC_SKIP_SUM_HANDOFF=0 = the sum_n launch of before)
for w in molhiv_b512_softmax molhiv_pub_phc4_2x200; do
  for i in 1 2 3 4; do
    for h in 0 1; do
    PHC_SKIP_SUM_HANDOFF=$h python3 bench.py --workload $w --cpu-steps 0 --profile-steps 0 --steps 300 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w handoff=$h', d['ms_per_step'])"
    done
  done
done
