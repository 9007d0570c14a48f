cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_clip_adam_fused.py tests/test_gpu_head_loss.py -x -q 2>&1 | tail -15
for i in 1 2; do
for v in "PHC_CLIP_ADAM_FUSED=0 PHC_HEAD_LOSS=0" "PHC_CLIP_ADAM_FUSED=1 PHC_HEAD_LOSS=0" "PHC_CLIP_ADAM_FUSED=0 PHC_HEAD_LOSS=1" "PHC_CLIP_ADAM_FUSED=1 PHC_HEAD_LOSS=1"; do
  echo "$v: $(env $v python3 bench.py --cpu-steps 0 --profile-steps 0 --steps 300 --min-time 3 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")"
done
done
