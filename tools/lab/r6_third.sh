mkdir -p gpurun_out
timeout 2400 python tools/diag/p_parity_decisions.py 4 > gpurun_out/r06_p_parity_decisions.txt 2>&1
tail -60 gpurun_out/r06_p_parity_decisions.txt
timeout 900 python tools/lab/dw_sweep.py > gpurun_out/r06_dw_sweep_pp.txt 2>&1
VILCO_GEMM_GL_SINGLE=1 timeout 900 python tools/lab/dw_sweep.py > gpurun_out/r06_dw_sweep_gl.txt 2>&1
cat gpurun_out/r06_dw_sweep_pp.txt gpurun_out/r06_dw_sweep_gl.txt
