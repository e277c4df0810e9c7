mkdir -p gpurun_out
( for p in 7 8 9; do for s in 0 1; do for ch in 1 40; do timeout 300 tools/lab/nullstream_repro $s $p 3000 20000 $ch; done; done; done
  timeout 300 tools/lab/nullstream_repro 0 8 20000 2000 4; timeout 300 tools/lab/nullstream_repro 0 9 20000 2000 4 ) > gpurun_out/r06_nullstream_repro4.txt 2>&1
cat gpurun_out/r06_nullstream_repro4.txt
bash tools/ab_bench.sh "VILCO_GRAPH_STREAMS=text,heads" "VILCO_GRAPH_STREAMS=text,heads,dw" "VILCO_GRAPH_STREAMS=text,heads" "VILCO_GRAPH_STREAMS=text,heads,dw" 2>&1 | tee gpurun_out/r06_ab_dwfork.txt
