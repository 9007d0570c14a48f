# A/B two library builds on ONE box, alternating runs: scripts/ab.sh <variant-name> [rounds] [extra bench args]
# (the variant is built with `python -m phc_gnn_amd.build --variant <name> -D...`; the product library is "base")
V=$1; R=${2:-4}; shift; shift
for i in $(seq $R); do
  for which in base $V; do
    if [ $which = base ]; then unset PHC_LIB_PATH; else export PHC_LIB_PATH=$GRAFT_REPO_ROOT/scripts/exp/variants/libphc_hip_$V.so; fi
    python3 bench.py --cpu-steps 0 --profile-steps 0 --steps 300 "$@" 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$which', d['ms_per_step'])"
  done
done
