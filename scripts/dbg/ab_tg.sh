for v in new pf44 new pf44 new pf44; do
  if [ $v = pf44 ]; then export PHC_LIB_PATH=$GRAFT_REPO_ROOT/scripts/exp/variants/libphc_hip_pf44.so; else unset PHC_LIB_PATH; fi
  python bench.py --cpu-steps 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['timing']['ms_per_step_min'], d['timing']['ms_per_step_max'])"
done
unset PHC_LIB_PATH
bash scripts/rocprof_bench.sh tg1 2>&1 | grep -i "tail_partial" | cut -c1-130 | sed "s/^/pf 8,6: /"
export PHC_LIB_PATH=$GRAFT_REPO_ROOT/scripts/exp/variants/libphc_hip_pf44.so
bash scripts/rocprof_bench.sh tg0 2>&1 | grep -i "tail_partial" | cut -c1-130 | sed "s/^/pf 4,4: /"
