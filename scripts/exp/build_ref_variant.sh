#!/bin/bash
# Build the library of a git revision as an A/B variant: scripts/exp/build_ref_variant.sh <rev> <name>
#   -> scripts/exp/variants/libphc_hip_<name>.so (use with scripts/ab.sh <name>)
set -e
REV=${1:-HEAD}; NAME=${2:-old}
cd "$(dirname "$0")/../.."
T=/tmp/phc_variant_$NAME; rm -rf $T; mkdir -p $T scripts/exp/variants
git archive $REV phc_gnn_amd/csrc include | tar -x -C $T
objs=""
for s in $T/phc_gnn_amd/csrc/*.hip; do
  o=$T/$(basename $s .hip).o; objs="$objs $o"
  hipcc -c -x hip $s -o $o -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$T/include -I$T/phc_gnn_amd/csrc -fno-gpu-rdc -w &
  while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 1; done
done
wait
hipcc -shared -fPIC --offload-arch=gfx950 -o scripts/exp/variants/libphc_hip_$NAME.so $objs
ls -la scripts/exp/variants/libphc_hip_$NAME.so
