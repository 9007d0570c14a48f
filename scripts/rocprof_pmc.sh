# PMC passes (separate runs, no tracing domains besides kernel-trace) for HBM traffic of the engine kernels of ONE workload
#   bash scripts/rocprof_pmc.sh <round> [workload]     -> gpurun_out/pmc_<round>_<workload>_{FETCH_SIZE,WRITE_SIZE}/
set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
WL=${2:-molhiv_b512}
for c in FETCH_SIZE WRITE_SIZE; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$1_${WL}_$c
  rm -rf $OUT && mkdir -p $OUT
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT -- python3 bench.py --workload $WL --steps ${PHC_PMC_STEPS:-10} --warmup ${PHC_PMC_WARMUP:-3} --cpu-steps 0 --mode eager --profile-steps 0 --min-time 0 > $OUT/bench.log 2>&1
  ls $OUT/*/ | head
done
