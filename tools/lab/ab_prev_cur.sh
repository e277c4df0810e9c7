# same-box A/B of two library builds: tools/lab/libvilco_prev.so (copy of an earlier build) vs the current one
for rep in 1 2 3; do
  VILCO_HIP_LIB=$PWD/tools/lab/libvilco_prev.so bash tools/ab_bench.sh "A=prev"
  bash tools/ab_bench.sh "A=cur"
done
