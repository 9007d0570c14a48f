for t in 0 256 2048 4096 0 256 2048 4096; do
  PHC_TG_TARGET=$t python bench.py --cpu-steps 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tg_target=$t', d['ms_per_step'], d['timing']['ms_per_step_min'], d['timing']['ms_per_step_max'])"
done
