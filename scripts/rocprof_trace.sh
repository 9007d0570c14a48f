# kernel trace of the graph-mode bench without the eager profile pass; prints the last replay in launch order
set -x
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_$1
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 30 --warmup 5 --cpu-steps 0 --profile-steps 0 --min-time 0 > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-200
T=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 scripts/trace_one_step.py $T ${2:-1600} > $OUT/last_step.txt
python3 scripts/summarize_rocprof.py $T 25 60 > $OUT/summary.txt          # the last 25 replayed steps only
find $OUT -name "*kernel_trace.csv" -size +20M -delete
tail -3 $OUT/last_step.txt
