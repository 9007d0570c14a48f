cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -iE "ICACHE|IFETCH|WAIT_INST|INST_LEVEL|SQ_WAVES\b|SQ_INSTS_VALU\b|SQ_WAIT_ANY|SQ_ACTIVE_INST" | sort -u | head -40
for P in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_IFETCH SQ_INSTS_VALU"; do
  tag=$(echo $P | cut -c1-14 | tr ' ' '_')
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmci_$tag
  rm -rf $OUT && mkdir -p $OUT
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT -- python3 bench.py --steps 6 --warmup 2 --cpu-steps 0 --mode eager --profile-steps 0 --min-time 0 > $OUT/bench.log 2>&1
  tail -1 $OUT/bench.log | cut -c1-100
  F=$(ls -t $OUT/*/*counter_collection.csv | head -1)
  python3 - "$F" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in sorted(agg, key=lambda k: -sum(agg[k].values()))[:12]:
    print(k, {c: round(v / max(cnt[(k, c)], 1)) for c, v in agg[k].items()})
PY
  find $OUT -name "*.csv" -size +5M -delete
done
