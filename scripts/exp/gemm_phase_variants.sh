# Which phase of the streaming GEMM costs what INSIDE the captured step: builds variants of phm_linear.hip with one
# phase compiled out (results wrong, timing only) into build/variants/ and runs the bench with each (PHC_LIB_PATH).
set -e
cd $(dirname $0)/../..
mkdir -p build/variants
INC="-Iinclude -Iphc_gnn_amd/csrc"
for v in NO_HBUILD NO_MFMA NO_EPI NO_XLOAD; do
  /opt/rocm/bin/hipcc -c -x hip phc_gnn_amd/csrc/phm_linear.hip -o build/variants/phm_linear_$v.o -O3 -std=c++17 -fPIC --offload-arch=gfx950 $INC -DPHC_EXP_$v -fno-gpu-rdc -Wno-unused-value &
done
wait
for v in NO_HBUILD NO_MFMA NO_EPI NO_XLOAD; do
  objs=$(ls phc_gnn_amd/build/*.o | grep -v phm_linear.o)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build/variants/libphc_hip_$v.so $objs build/variants/phm_linear_$v.o
done
ls -la build/variants/*.so
