export PHC_LIB_PATH=$GRAFT_REPO_ROOT/scripts/exp/variants/libphc_hip_tailstamps.so
run() {  # workload front dhT
  echo "=== $1: front $2 dH target $3"
  PHC_TAIL_FRONT=$2 PHC_DH_TARGET3=$3 python3 scripts/dbg/tail_wg_timeline.py $1 2>&1 | grep -E "launch span|atom tables|bond tables|dH  "
  PHC_TAIL_FRONT=$2 PHC_DH_TARGET3=$3 python3 bench.py --workload $1 --cpu-steps 0 --profile-steps 0 --steps 200 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   ms/step', d['ms_per_step'])"
}
for cfg in "9999 512" "176 336" "160 352" "128 384" "192 320" "224 288" "96 416" "9999 512"; do run molhiv_b512 $cfg; done
for cfg in "9999 512" "64 448" "128 384"; do run zinc_b1024 $cfg; done
for cfg in "9999 512" "160 352" "224 288"; do run molhiv_pub_phc4_2x200 $cfg; done
