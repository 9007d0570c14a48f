# kernel stats of one bench workload: bash scripts/dbg/prof_wl.sh <workload> <tag>
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$2
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --workload $1 --steps 50 --warmup 10 --cpu-steps 0 --min-time 0 --profile-steps 0 > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-160
python3 scripts/summarize_rocprof.py $(ls -t $OUT/*/*kernel_trace.csv | head -1) 40 60 | head -48      # the last 40 replayed steps
find $OUT -name "*kernel_trace.csv" -size +20M -delete
