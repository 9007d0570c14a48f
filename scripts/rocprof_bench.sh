# kernel-trace + stats of the default bench command; summaries are copied to profiles/ by hand
set -x
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$1
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 50 --warmup 10 --cpu-steps 0 --min-time 0 > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -45 {}'
ls -la $OUT/*/ | head
# keep only the small summaries
find $OUT -name "*kernel_trace.csv" -size +20M -delete
