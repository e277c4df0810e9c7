#!/bin/bash
# builds libvilco variants (EXTRA=-DVILCO_GEMM_VARIANT=n) into tools/lab/ -- run HERE (CPU); then tools/lab/gemm_variants_run.sh on the GPU
set -e
cd "$(dirname "$0")/../.."
for v in "$@"; do
  rm -f vilco_amd/csrc/gemm.o
  make -C vilco_amd/csrc -j8 EXTRA="$(echo $v | sed "s/^p\(.*\)/-DVILCO_GEMM_PRIO=\1/; s/^\([0-9]\)/-DVILCO_GEMM_VARIANT=\1/")" > /dev/null
  cp vilco_amd/libvilco_hip.so tools/lab/libvilco_v$v.so
done
rm -f vilco_amd/csrc/gemm.o
make -C vilco_amd/csrc -j8 > /dev/null
