# A/B of the step tail: register-resident dH tiles inside the table launch (form 3, default), in their own launch (form 2), off
for wl in molhiv_b512 zinc_b1024; do
for cfg in "PHC_TAIL_DIRECT_FORM=3" "PHC_TAIL_DIRECT_FORM=2" "PHC_DH_NO_DIRECT=1" "PHC_TAIL_DIRECT_FORM=3" "PHC_TAIL_DIRECT_FORM=2" "PHC_DH_NO_DIRECT=1"; do
  env $cfg python bench.py --cpu-steps 0 --workload $wl 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl $cfg', d['ms_per_step'], d['timing']['ms_per_step_min'], d['timing']['ms_per_step_max'])"
done; done
