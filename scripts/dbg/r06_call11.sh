cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
timeout 1500 python -m pytest tests/test_gpu_dispatch_surface.py -q -k "toggle_flip or library_switch" 2>&1 | grep -E "^FAILED|^E  |passed|failed" | head -30
