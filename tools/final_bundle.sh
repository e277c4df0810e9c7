# the whole round-end bundle + the GPU test suite in one gpurun call:  gpurun --timeout 3600 -- bash tools/final_bundle.sh
TAG=r06_z bash tools/run_final.sh > gpurun_out/r06_z_run_final.log 2>&1
python -m pytest tests -x -q -m gpu > gpurun_out/r06_z_gpu_tests.txt 2>&1
tail -3 gpurun_out/r06_z_gpu_tests.txt
