cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest "tests/test_gpu_fullwidth.py::test_fullwidth_plain_module_path" -q -x 2>&1 | grep -E "^E  |^FAILED|passed|failed" | head -12 | cut -c1-300
