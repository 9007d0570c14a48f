cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_mlp_fwd_fused.py tests/test_gpu_mlp_bwd_fused.py -x -q 2>&1 | tail -3
PYTHONPATH=. python scripts/dbg/mlp_all_wg_phases.py | tail -16
bash scripts/ab.sh r6b 3
