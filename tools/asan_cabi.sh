#!/bin/bash
# Host-side AddressSanitizer run of the C ABI: builds libvilco_hip_asan.so
# (host code instrumented, device code untouched) and runs the CPU tests that drive the entry points' argument validation,
# workspace / plan arithmetic and symbol table through it.  No GPU needed.  Output: profiles/r04_asan_cabi.txt
set -e
cd "$(dirname "$0")/.."
make -C vilco_amd/csrc asan -j4 > /dev/null
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
OUT=profiles/r04_asan_cabi.txt
{
  echo "# $(date -u +%F) host-side ASan: LD_PRELOAD=$(basename $RT) VILCO_HIP_LIB=vilco_amd/libvilco_hip_asan.so pytest tests/test_cabi_cpu.py"
  LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 VILCO_HIP_LIB=$PWD/vilco_amd/libvilco_hip_asan.so \
    python -m pytest tests/test_cabi_cpu.py -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|ERROR|AddressSanitizer|SUMMARY" || true
} | tee $OUT
