# in-graph cost of each launch family of the headline step (PHC_ABLATE drops the launches; results are invalid, only the time is read)
run() { PHC_ABLATE=$1 timeout 150 python3 bench.py --steps 100 --warmup 10 --cpu-steps 0 --profile-steps 0 --min-time 2 2>/dev/null | tail -1 | python3 -c "
import sys, json
try:
    print(json.loads(sys.stdin.read())['ms_per_step'])
except Exception:
    print('nan')"; }
base=$(run none)
echo "baseline ms/step: $base"
for k in phm_build_H phm_gemm_fwd phm_gemm_dx phm_gemm_ag_fwd phm_dH_partial phm_dH_reduce phm_contract_dW_dA bn_bwd_partial aggregate_enc_dx csr_scan csr_canonicalize bce_masked_fwd gated_pool_fwd gated_pool_bwd clip_adam downstream_fwd downstream_bwd; do
  t=$(run $k)
  python3 -c "print(f'$k: {1e3*($base-float(\"$t\")):8.1f} us/step  (step $t ms)')"
done
base2=$(run none)
echo "baseline again: $base2"
