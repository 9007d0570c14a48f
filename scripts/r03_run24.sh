#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03x; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_fullwidth.py -m gpu -q -s -k "full_size" > $O/tests.log 2>&1
echo "tests rc=$?"; grep -E "passed|failed|gradient tensors vs" $O/tests.log | tail -40 | cut -c1-260
