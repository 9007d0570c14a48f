# kernel trace of the replayed zinc / molpcba step with the tail's slot shares overridden (-DPHC_TAIL_KNOBS variant build:
#   python -m phc_gnn_amd.build --variant tailknobs -DPHC_TAIL_KNOBS): which launches move when the tables get few slots
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export PHC_LIB_PATH=$GRAFT_REPO_ROOT/scripts/exp/variants/libphc_hip_tailknobs.so
WL=${1:-zinc_b1024}
for sh in ${SHARES:-default 24,48,440 64,96,352}; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_shares_$(echo $sh | tr , _)
  rm -rf $OUT && mkdir -p $OUT
  if [ $sh = default ]; then unset PHC_TAIL_SHARES; else export PHC_TAIL_SHARES=$sh; fi
  rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --workload $WL --steps 30 --warmup 5 --cpu-steps 0 --profile-steps 0 --min-time 0 > $OUT/bench.log 2>&1
  echo "=== $WL shares $sh: $(grep '^{"metric"' $OUT/bench.log | tail -1 | python3 -c 'import json,sys; print(json.loads(sys.stdin.read())["ms_per_step"])')"
  T=$(find $OUT -name "*kernel_trace.csv" | head -1)
  python3 scripts/summarize_rocprof.py $T 25 ${TOP:-14} | tail -$((${TOP:-14} + 2))
  find $OUT -name "*.csv" -size +5M -delete
done
