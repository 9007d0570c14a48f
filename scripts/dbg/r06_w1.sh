cd $GRAFT_REPO_ROOT
PYTHONPATH=. python scripts/dbg/mlp_fwd_agg_phases.py 2>&1 | grep -E "agg=|since"
bash scripts/ab.sh r6base 3
