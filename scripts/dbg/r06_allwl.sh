cd $GRAFT_REPO_ROOT
for wl in molhiv_b512 molhiv_b32 molhiv_b512_softmax molhiv_b512_conv zinc_b1024 molpcba_b1024 molhiv_pub_phc4_2x200 zinc_pub_phc5_4x200 zinc_pub_phc4_4x180 molpcba_pub_phc4_7x512; do
  echo "$wl: $(timeout 400 python3 bench.py --workload $wl --cpu-steps 0 --profile-steps 0 --steps 200 --min-time 2 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")"
done
