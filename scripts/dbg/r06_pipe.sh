cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
for v in 0 1; do
PHC_GRAPH_PIPELINE=$v python3 bench.py --cpu-steps 0 --profile-steps 0 --steps 300 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('pipeline=$v', d['ms_per_step'], d['config']['launch'][:120])"
done; done
