# A/B of the wide aggregation-in-GEMM launch inside the replayed steps (alternating runs on one box)
#   A: PHC_GEMM_AGGREGATE_WIDE=0 (separate launches)  B: fused, norm fold off  C: fused + pending norm folded
for wl in ${WLS:-zinc_b1024 molpcba_b1024 zinc_pub_phc5_4x200 zinc_pub_phc4_4x180}; do
  for rep in 1 2; do
    for cfg in "0 1" "1 0" "1 1"; do
      set -- $cfg
      r=$(PHC_GEMM_AGGREGATE_WIDE=$1 PHC_GEMM_AGGREGATE_WIDE_NORM=$2 python bench.py --workload $wl --cpu-steps 0 --profile-steps 0 --min-time 3 2>/dev/null | grep '^{"metric"' | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
      echo "$wl wide=$1 norm=$2 ms_per_step=$r"
    done
  done
done
