# second call of the round-end collection: the bench lines that quote the PMC passes (profiles/r06_pmc_*.json must be in place), the
# per-workgroup phase clocks of the one-launch MLP kernels, the instruction-cache counters
cd $GRAFT_REPO_ROOT
{
echo "# every workgroup's phase clocks of the aggregating fused forward (last layer of a molhiv_b512 forward pass, eager; us since the first workgroup's entry)"
PYTHONPATH=. python scripts/dbg/mlp_all_wg_phases.py 2>&1 | grep -v "^W2026\|amdgpu.ids"
echo
echo "# ... and of the fused backward carrying the aggregation's dx (the last such launch of the backward pass)"
PYTHONPATH=. python scripts/dbg/mlp_all_wg_phases.py bwd 2>&1 | grep -v "^W2026\|amdgpu.ids"
} > gpurun_out/r06_mlp_all_wg_phases.txt
bash scripts/dbg/r06_phases_all.sh > /dev/null 2>&1
bash scripts/dbg/r06_icache.sh 2>&1 | grep -E "^void phc::mlp|^phc::|^void phc::tail|SQC|SQ_" | grep -v "Counter_Name\|Description\|Expression" > gpurun_out/r06_icache_counters.txt
python bench.py 2>&1 | tail -1 > gpurun_out/r06_bench_default.json
for wl in zinc_b1024 molpcba_b1024; do
  python bench.py --workload $wl --cpu-steps 8 --cpu-all-cores-max 0 2>&1 | grep '^{"metric"' | tail -1 > gpurun_out/r06_bench_$wl.json
done
python bench.py --workload synth_10m50m --steps 10 --warmup 2 2>&1 | tail -1 > gpurun_out/r06_bench_synth_10m50m_1gpu.json
python bench.py --workload synth_10m50m --layers 4 --steps 5 --warmup 2 2>&1 | tail -1 > gpurun_out/r06_bench_synth_10m50m_4layers_1gpu.json
cut -c1-200 gpurun_out/r06_bench_default.json
# every workgroup of the step tail's first launch against one clock (the -DPHC_TAIL_STAMPS variant: python -m phc_gnn_amd.build --variant tailstamps -DPHC_TAIL_STAMPS)
if [ -f scripts/exp/variants/libphc_hip_tailstamps.so ]; then
  for wl in molhiv_b512 molhiv_pub_phc4_2x200 zinc_b1024; do
    PHC_DEBUG_PLANS=1 PHC_LIB_PATH=$GRAFT_REPO_ROOT/scripts/exp/variants/libphc_hip_tailstamps.so python scripts/dbg/tail_wg_timeline.py $wl 2>&1 | grep -v "amdgpu.ids" | awk '!seen[$0]++'
    echo
  done > gpurun_out/r06_tail_wg_timeline.txt
fi
