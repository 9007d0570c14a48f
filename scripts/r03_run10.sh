#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03j
export TMPDIR=/tmp
timeout 300 python scripts/dbg/hs_phases.py 23600 256 256 > gpurun_out/r03j/phases_256.log 2>&1
grep -v "^$" gpurun_out/r03j/phases_256.log | tail -9
timeout 300 python scripts/dbg/hs_phases.py 13824 128 128 > gpurun_out/r03j/phases_128.log 2>&1
grep -v "^$" gpurun_out/r03j/phases_128.log | tail -8
