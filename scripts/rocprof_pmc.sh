# PMC passes (separate runs, no tracing domains besides kernel-trace) for HBM traffic of the engine kernels
set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$1_$c
  rm -rf $OUT && mkdir -p $OUT
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT -- python3 bench.py --steps 10 --warmup 3 --cpu-steps 0 --mode eager --profile-steps 0 --min-time 0 > $OUT/bench.log 2>&1
  ls $OUT/*/ | head
done
