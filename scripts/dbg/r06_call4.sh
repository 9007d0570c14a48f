cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_head_loss.py -q -rA 2>&1 | grep -E "^PASSED|^FAILED|^E  " | head -40
