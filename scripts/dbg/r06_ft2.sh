cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest "tests/test_gpu_fullwidth.py::test_fullwidth_plain_module_path" -q 2>&1 | grep -E "^FAILED|passed|failed" | head
timeout 900 python -m pytest tests/test_gpu_two_rank_engine.py -q -x 2>&1 | tail -40 | cut -c1-250
