# MFMA-utilisation counters of the engine kernels of ONE workload: PMC passes in their own runs (kernel-trace only
# beside them), the program directly behind `--`.
#   bash scripts/rocprof_pmc_mfma.sh <round> [workload]   -> gpurun_out/pmcm_<round>_<workload>_<pass>/ + gpurun_out/<round>_pmc_mfma.{json,txt}
# pass A: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE   (busy cycles summed over the SIMDs / CUs; GUI_ACTIVE summed over 8 XCDs)
# pass B: SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES   (cross-check: MFMA ops issued against the algorithmic count)
set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=$1
WL=${2:-molhiv_b512}
declare -A PASS
PASS[A]="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
PASS[B]="SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
for p in A B; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcm_${R}_${WL}_$p
  rm -rf $OUT && mkdir -p $OUT
  rocprofv3 --kernel-trace --pmc ${PASS[$p]} --output-format csv -d $OUT -- python3 bench.py --workload $WL --steps 10 --warmup 3 --cpu-steps 0 --mode eager --profile-steps 0 --min-time 0 > $OUT/bench.log 2>&1
  tail -2 $OUT/bench.log
done
A=$(ls -t gpurun_out/pmcm_${R}_${WL}_A/*/*counter_collection.csv | head -1)
B=$(ls -t gpurun_out/pmcm_${R}_${WL}_B/*/*counter_collection.csv 2>/dev/null | head -1)
python3 scripts/summarize_pmc_mfma.py gpurun_out/${R}_pmc_mfma.json $WL $A $B >> gpurun_out/${R}_pmc_mfma.txt 2>&1
find gpurun_out/pmcm_${R}_${WL}_A gpurun_out/pmcm_${R}_${WL}_B -name "*.csv" -size +5M -delete
