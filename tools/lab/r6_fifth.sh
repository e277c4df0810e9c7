mkdir -p gpurun_out
( for p in 5 6; do for s in 0 1 2; do for ch in 1 40; do timeout 200 tools/lab/nullstream_repro $s $p 3000 20000 $ch; done; done; done
  timeout 200 tools/lab/nullstream_repro 0 6 20000 2000 4 ) > gpurun_out/r06_nullstream_repro3.txt 2>&1
cat gpurun_out/r06_nullstream_repro3.txt
