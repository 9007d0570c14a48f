for cfg in "PHC_X=1" "PHC_BN_BWD_APPLY_FLAT=1" "PHC_X=1" "PHC_BN_BWD_APPLY_FLAT=1" "PHC_X=1" "PHC_BN_BWD_APPLY_FLAT=1"; do
  env $cfg python bench.py --cpu-steps 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'], d['timing']['ms_per_step_min'], d['timing']['ms_per_step_max'])"
done
