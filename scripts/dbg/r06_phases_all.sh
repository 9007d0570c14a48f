# in-kernel phase clocks of the one-launch MLP kernels (profiles/r06_mlp_fused_phases.txt)
cd $GRAFT_REPO_ROOT
{
echo "# phc_mlp_norm_bwd_fused vs the three launches (scripts/dbg/mlp_bwd_phases.py; isolated launches, N = 12874, us; workgroup 0 has two 16-row steps, the last workgroup one)"
PYTHONPATH=. python scripts/dbg/mlp_bwd_phases.py 2>&1 | grep -E "launch|wg0|wgLast|diff"
echo "# phc_mlp_norm_fwd_fused without the aggregation vs plain GEMM1 + XF GEMM2 (scripts/dbg/mlp_fwd_phases.py)"
PYTHONPATH=. python scripts/dbg/mlp_fwd_phases.py 2>&1 | grep -E "launch|wg0|wgLast"
echo "# the aggregating form inside a molhiv_b512 forward pass, eager (scripts/dbg/mlp_fwd_agg_phases.py; last layer: pending norm + skip)"
PYTHONPATH=. python scripts/dbg/mlp_fwd_agg_phases.py 2>&1 | grep -E "agg=|since"
} > gpurun_out/r06_mlp_fused_phases.txt 2>&1
python bench.py 2>&1 | tail -1 > gpurun_out/r06_bench_default.json
python bench.py --workload synth_10m50m --steps 10 --warmup 2 2>&1 | tail -1 > gpurun_out/r06_bench_synth_10m50m_1gpu.json
cat gpurun_out/r06_mlp_fused_phases.txt
