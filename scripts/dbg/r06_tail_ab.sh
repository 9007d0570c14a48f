python3 -m pytest tests/test_gpu_fullwidth.py tests/test_gpu_ops.py -q -m gpu 2>&1 | tail -3
PHC_DEBUG_PLANS=1 PHC_LIB_PATH=$GRAFT_REPO_ROOT/scripts/exp/variants/libphc_hip_tailstamps.so python3 scripts/dbg/tail_wg_timeline.py zinc_pub_phc4_4x180 2>&1 | grep -E "phc tail|launch span|tables|dH  |job" | sort | uniq | head -40
for w in zinc_pub_phc4_4x180 molhiv_pub_phc4_2x200; do
  echo "== $w"; bash scripts/ab.sh prev 3 --workload $w
done
