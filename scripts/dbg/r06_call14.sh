cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_gemm_hs.py tests/test_gpu_model.py tests/test_gpu_bucket_independence.py -q -rf 2>&1 | grep -E "^FAILED|^E  |passed|failed" | head -20
for wl in zinc_b1024 molpcba_b1024 zinc_pub_phc5_4x200; do
for v in "PHC_BUCKET_ROUND_ROWS=4096" "PHC_BUCKET_ROUND_ROWS=0"; do
  echo "$wl $v: $(env $v timeout 400 python3 bench.py --workload $wl --cpu-steps 0 --profile-steps 0 --steps 200 --min-time 3 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['launch'][60:110])")"
done
done
