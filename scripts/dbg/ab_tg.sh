for i in 1 2 3; do
  python bench.py --cpu-steps 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['ms_per_step'], d['timing']['ms_per_step_min'], d['timing']['ms_per_step_max'])"
done
bash scripts/rocprof_bench.sh tg1 2>&1 | grep -i "tail_partial\|tail_reduce" | cut -c1-130
