cd $GRAFT_REPO_ROOT
timeout 2600 python -m pytest tests -m gpu -q 2>&1 | grep -E "^FAILED|^ERROR|passed|failed" | head -60
