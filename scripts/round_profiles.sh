# Round-end evidence in one call: kernel stats + launch sequence + PMC traffic of the headline bench, the other workloads' lines.
#   scripts/gpu.sh --timeout 2400 -- 'bash scripts/round_profiles.sh r02'
R=${1:-rXX}
bash scripts/rocprof_bench.sh ${R} > gpurun_out/${R}_rocprof_bench.log 2>&1
bash scripts/rocprof_trace.sh ${R}b 1800 > gpurun_out/${R}_rocprof_trace.log 2>&1
rm -f gpurun_out/${R}_pmc_traffic.json gpurun_out/${R}_pmc_traffic.txt
for wl in molhiv_b512 zinc_b1024 molpcba_b1024; do
  bash scripts/rocprof_pmc.sh ${R} $wl > gpurun_out/${R}_rocprof_pmc_$wl.log 2>&1
  F=$(ls -t gpurun_out/pmc_${R}_${wl}_FETCH_SIZE/*/*counter_collection.csv | head -1)
  W=$(ls -t gpurun_out/pmc_${R}_${wl}_WRITE_SIZE/*/*counter_collection.csv | head -1)
  python3 scripts/summarize_pmc.py $F $W gpurun_out/${R}_pmc_traffic.json $wl >> gpurun_out/${R}_pmc_traffic.txt 2>&1
  find gpurun_out/pmc_${R}_${wl}_FETCH_SIZE gpurun_out/pmc_${R}_${wl}_WRITE_SIZE -name "*.csv" -size +5M -delete
done
# MFMA-pipe utilisation counters (SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES; MOPS cross-check) of the same three workloads
rm -f gpurun_out/${R}_pmc_mfma.json gpurun_out/${R}_pmc_mfma.txt
for wl in molhiv_b512 zinc_b1024 molpcba_b1024; do
  bash scripts/rocprof_pmc_mfma.sh ${R} $wl > gpurun_out/${R}_rocprof_pmcm_$wl.log 2>&1
done
# HBM traffic of config 5 (one GPU, one layer): the one workload that IS HBM-bound
PHC_PMC_STEPS=4 bash scripts/rocprof_pmc.sh ${R} synth_10m50m > gpurun_out/${R}_rocprof_pmc_synth.log 2>&1
F=$(ls -t gpurun_out/pmc_${R}_synth_10m50m_FETCH_SIZE/*/*counter_collection.csv | head -1)
W=$(ls -t gpurun_out/pmc_${R}_synth_10m50m_WRITE_SIZE/*/*counter_collection.csv | head -1)
python3 scripts/summarize_pmc.py $F $W gpurun_out/${R}_pmc_traffic.json synth_10m50m >> gpurun_out/${R}_pmc_traffic.txt 2>&1
find gpurun_out/pmc_${R}_synth_10m50m_FETCH_SIZE gpurun_out/pmc_${R}_synth_10m50m_WRITE_SIZE -name "*.csv" -size +5M -delete
# what the reference's own loop gets: the loop body of train_hiv.py verbatim, and harness.train_step eager
for m in refloop eager; do
  python bench.py --mode $m --cpu-steps 0 --profile-steps 0 --steps 50 --min-time 3 2>&1 | tail -1 > gpurun_out/${R}_bench_molhiv_b512_$m.json
done
# (the BASELINE configurations carry their CPU denominator: a bounded oracle sample at 6 threads, bench.py cpu_baseline)
for wl in zinc_b1024 molpcba_b1024; do
  python bench.py --workload $wl --cpu-steps 8 --cpu-all-cores-max 0 2>&1 | grep '^{"metric"' | tail -1 > gpurun_out/${R}_bench_$wl.json
done
for wl in molhiv_b512_softmax molhiv_b512_conv molhiv_b32 molhiv_pub_phc4_2x200 zinc_pub_phc5_4x200 zinc_pub_phc4_4x180 molpcba_pub_phc4_7x512; do
  python bench.py --workload $wl --cpu-steps 0 2>&1 | grep '^{"metric"' | tail -1 > gpurun_out/${R}_bench_$wl.json
done
# kernel stats of the wide configurations and of config-5 at its full size (one GPU)
for wl in zinc_b1024 molpcba_b1024 synth_10m50m; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${R}_$wl
  rm -rf $OUT && mkdir -p $OUT
  if [ $wl = synth_10m50m ]; then ST=8; WU=2; else ST=50; WU=10; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --workload $wl --steps $ST --warmup $WU --cpu-steps 0 --min-time 0 --profile-steps 0 > $OUT/bench.log 2>&1
  # per REPLAYED step: the kernel trace restricted to the last steps (set-up launches excluded); config 5 is eager: whole-run stats
  if [ $wl = synth_10m50m ]; then
    # (eager: the timed steps delimited by the layer's last GEMM, the BA form of the dx GEMM: once per step)
    python3 scripts/summarize_rocprof.py $(ls -t $OUT/*/*kernel_trace.csv | head -1) $ST 60 "gemm_hs16_kernel<128, 2, 4, 1, 4, 8>" > gpurun_out/${R}_bench_${wl}_kernel_stats_summary.txt
  else
    python3 scripts/summarize_rocprof.py $(ls -t $OUT/*/*kernel_trace.csv | head -1) 40 60 > gpurun_out/${R}_bench_${wl}_kernel_stats_summary.txt
  fi
  find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
done
python bench.py --workload synth_10m50m --steps 10 --warmup 2 2>&1 | tail -1 > gpurun_out/${R}_bench_synth_10m50m_1gpu.json
python bench.py --workload synth_10m50m --layers 4 --steps 5 --warmup 2 2>&1 | tail -1 > gpurun_out/${R}_bench_synth_10m50m_4layers_1gpu.json
timeout 900 python scripts/microbench.py huge > gpurun_out/${R}_microbench_config5_1gpu.txt 2>&1
python bench.py 2>&1 | tail -1 > gpurun_out/${R}_bench_default.json
ls -la gpurun_out/${R}_* | head -30
