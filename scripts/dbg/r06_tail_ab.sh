# the tail's slot shares: tests of the tail / step paths, the timeline, then A/B against the previous revision's library
python3 -m pytest tests/test_gpu_step_tail.py tests/test_gpu_bucket_independence.py tests/test_gpu_fullwidth.py -q -m gpu -x 2>&1 | tail -3
PHC_DEBUG_PLANS=1 PHC_LIB_PATH=$GRAFT_REPO_ROOT/scripts/exp/variants/libphc_hip_tailstamps.so python3 scripts/dbg/tail_wg_timeline.py molhiv_b512 2>&1 | grep -E "phc tail|launch span|tables|dH  " | sort | uniq | head -8
for w in molhiv_b512 molhiv_b32 molhiv_b512_conv molhiv_b512_softmax zinc_b1024; do
  echo "== $w"; bash scripts/ab.sh prev 3 --workload $w
done
