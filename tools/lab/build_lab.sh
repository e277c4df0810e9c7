# lab build of the library with in-kernel stamps (tools only; never shipped)
set -e
cd "$(dirname "$0")/../../vilco_amd/csrc"
mkdir -p ../../tools/lab/obj
for f in gemm attn norm conv eltwise nms decode optim loss qkvpre sync status defer; do
  extra=""; [ $f = nms ] && extra="-ffp-contract=off"
  [ $f = gemm ] && extra="-DVILCO_LAB $LABFLAGS"
  [ $f = attn ] && extra="-DVILCO_LAB_ATTN $LABFLAGS"
  if [ $f = gemm ] || [ $f = attn ] || [ ! -f ../../tools/lab/obj/$f.o ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I../../include -Wno-unused-value -Wno-comment $extra -c $f.hip -o ../../tools/lab/obj/$f.o &
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/lab/libvilco_lab.so ../../tools/lab/obj/*.o
