cd $GRAFT_REPO_ROOT
PYTHONPATH=. python scripts/dbg/mlp_fwd_agg_phases.py 2>&1 | grep -E "agg=True|since"
timeout 600 python -m pytest tests/test_gpu_mlp_fwd_fused.py -q -rf 2>&1 | grep -E "^FAILED|^E  |passed|failed" | head -30
for i in 1 2 3; do
  echo "bench: $(timeout 300 python3 bench.py --cpu-steps 0 --profile-steps 0 --steps 300 --min-time 3 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")"
done
