#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03t; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_fullwidth.py -m gpu -q -k "g12" > $O/tests.log 2>&1
echo "tests rc=$?"; grep -E "passed|failed|Error|error" $O/tests.log | tail -30 | cut -c1-300
