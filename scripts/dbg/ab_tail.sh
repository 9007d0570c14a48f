# A/B of the register-resident dH form (PHC_DH_NO_DIRECT=1 turns it off) on the bench workloads
for wl in molhiv_b512 zinc_b1024 molpcba_b1024; do
for cfg in "PHC_X=1" "PHC_DH_NO_DIRECT=1" "PHC_X=1" "PHC_DH_NO_DIRECT=1"; do
  env $cfg python bench.py --cpu-steps 0 --workload $wl 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl $cfg', d['ms_per_step'], d['timing']['ms_per_step_min'], d['timing']['ms_per_step_max'])"
done; done
