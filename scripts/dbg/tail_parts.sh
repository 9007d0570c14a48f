# which family of workgroups makes the first launch of phc_step_tail long (variant build -DPHC_EXP_TAIL_SKIP)
export PHC_LIB_PATH=$GRAFT_REPO_ROOT/scripts/exp/variants/libphc_hip_tailskip.so
for sk in none atom,bond dh; do
  export PHC_EXP_TAIL_SKIP=$sk
  bash scripts/rocprof_bench.sh ts_$sk 2>&1 | grep -i "tail_partial" | cut -c1-120 | sed "s/^/skip=$sk: /"
done
