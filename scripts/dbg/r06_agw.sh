cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_step_fusions.py tests/test_gpu_fullwidth.py -x -q 2>&1 | tail -3
for wl in zinc_b1024 molpcba_b1024 zinc_pub_phc5_4x200; do
for rr in 4096 0; do
  for which in base r6c; do
    if [ $which = base ]; then unset PHC_LIB_PATH; else export PHC_LIB_PATH=$GRAFT_REPO_ROOT/scripts/exp/variants/libphc_hip_$which.so; fi
    PHC_BUCKET_ROUND_ROWS=$rr python3 bench.py --workload $wl --cpu-steps 0 --profile-steps 0 --steps 200 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$wl round=$rr $which', d['ms_per_step'], d['config']['launch'][60:110])"
  done
done
done
