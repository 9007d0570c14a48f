cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -i -E "mfma|SQ_BUSY|GRBM_GUI" | head -60 > gpurun_out/r06_counters_list.txt
rm -f gpurun_out/r06_pmc_mfma.json gpurun_out/r06_pmc_mfma.txt
bash scripts/rocprof_pmc_mfma.sh r06 molhiv_b512 > gpurun_out/r06_pmcm_molhiv.log 2>&1
for m in refloop eager; do
  python bench.py --mode $m --cpu-steps 0 --profile-steps 0 --steps 50 --min-time 3 2>&1 | tail -1 > gpurun_out/r06a_bench_molhiv_b512_$m.json
done
python bench.py --mode eager --optimizer torch --cpu-steps 0 --profile-steps 0 --steps 50 --min-time 3 2>&1 | tail -1 > gpurun_out/r06a_bench_molhiv_b512_eager_torchopt.json
python bench.py --cpu-steps 0 2>&1 | tail -1 > gpurun_out/r06a_bench_default.json
cat gpurun_out/r06_pmc_mfma.txt | head -40
for f in gpurun_out/r06a_bench_*.json; do python3 -c "
import json,sys
d=json.load(open('$f')); print('$f', d['ms_per_step'], d['value'])"; done
