# Real in-graph cost of each engine kernel: step time with the kernel's launches dropped (PHC_ABLATE) vs the full step.
# Results of ablated runs are numerically meaningless; only ms_per_step is read.
run() { PHC_ABLATE=$1 timeout 120 python3 bench.py --steps 100 --warmup 10 --cpu-steps 0 --profile-steps 0 2>/dev/null | tail -1 | python3 -c "
import sys, json
try:
    print(json.loads(sys.stdin.read())['ms_per_step'])
except Exception:
    print('nan')"; }
base=$(run none)
echo "baseline ms/step: $base"
for k in phm_build_H phm_gemm_fwd phm_gemm_dx phm_dH_partial phm_dH_reduce phm_contract_dW_dA bn_stats_partial bn_finalize bn_apply bn_bwd_partial bn_bwd_finalize bn_bwd_apply aggregate_enc_fwd aggregate_enc_dx table_grad table_grad_reduce aggregate_fwd aggregate_bwd gather_rows csr_histogram csr_scan csr_fill csr_canonicalize encode_fwd pack_edge_codes gated_pool_fwd gated_pool_bwd bce_masked_fwd bce_masked_bwd; do
  t=$(run $k)
  python3 -c "print(f'$k: {1e3*($base-float(\"$t\")):8.1f} us/step  (step $t ms)')"
done
base2=$(run none)
echo "baseline again: $base2"
