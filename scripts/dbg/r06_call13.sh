cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_auto_step.py -q -rf 2>&1 | grep -E "^FAILED|^E  |passed|failed" | head -30
for m in refloop eager; do
  echo "$m: $(python bench.py --mode $m --cpu-steps 0 --profile-steps 0 --steps 50 --min-time 3 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")"
done
echo "refloop PHC_AUTO_STEP=0: $(PHC_AUTO_STEP=0 python bench.py --mode refloop --cpu-steps 0 --profile-steps 0 --steps 50 --min-time 3 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")"
