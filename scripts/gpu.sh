#!/bin/bash
# Rebuild libphc_hip.so (stale objects only), then run the given command on an MI355X box through gpurun.
#   scripts/gpu.sh [--timeout S] -- '<command>'
set -e
cd "$(dirname "$0")/.."
python -m phc_gnn_amd.build > /tmp/phc_build.log 2>&1 || { grep -n "error" /tmp/phc_build.log | head -20; echo "BUILD FAILED"; exit 1; }
exec /usr/local/graft/bin/gpurun "$@"
