mkdir -p gpurun_out
( for s in 0 1; do timeout 120 tools/lab/nullstream_repro $s 4 3000 20000 1; timeout 120 tools/lab/nullstream_repro $s 4 3000 20000 8; done
  timeout 200 tools/lab/nullstream_repro 0 3 3000 20000 200; timeout 200 tools/lab/nullstream_repro 0 4 20000 2000 4 ) > gpurun_out/r06_nullstream_repro2.txt 2>&1
cat gpurun_out/r06_nullstream_repro2.txt
timeout 2400 python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -k "mask_realisations" -s 2>&1 | tail -30 | tee gpurun_out/r06_parity_test.txt
