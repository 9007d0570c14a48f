for v in nomfma noload; do
  export PHC_LIB_PATH=$GRAFT_REPO_ROOT/scripts/exp/variants/libphc_hip_$v.so
  bash scripts/rocprof_bench.sh v_$v 2>&1 | grep -i "direct_kernel" | cut -c1-120 | sed "s/^/$v: /"
done
