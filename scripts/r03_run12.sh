#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03l
export TMPDIR=/tmp
for wl in molhiv_b32 molhiv_b512 zinc_b1024; do
timeout 300 python scripts/dbg/fused_node_dbg.py $wl > gpurun_out/r03l/dbg_$wl.log 2>&1
grep trial gpurun_out/r03l/dbg_$wl.log | cut -c1-600
done
