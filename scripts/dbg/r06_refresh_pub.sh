# the published shapes' lines + the tail timeline after the even-width table forms (same commands as scripts/round_profiles.sh / r06_call_final.sh)
cd $GRAFT_REPO_ROOT
for wl in molhiv_pub_phc4_2x200 zinc_pub_phc5_4x200 zinc_pub_phc4_4x180; do
  python bench.py --workload $wl --cpu-steps 0 2>&1 | grep '^{"metric"' | tail -1 > gpurun_out/r06_bench_$wl.json
done
for wl in molhiv_b512 molhiv_pub_phc4_2x200 zinc_pub_phc4_4x180 zinc_b1024; do
  PHC_DEBUG_PLANS=1 PHC_LIB_PATH=$GRAFT_REPO_ROOT/scripts/exp/variants/libphc_hip_tailstamps.so python scripts/dbg/tail_wg_timeline.py $wl 2>&1 | grep -v "amdgpu.ids" | awk '!seen[$0]++'
  echo
done > gpurun_out/r06_tail_wg_timeline.txt
