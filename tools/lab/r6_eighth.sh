mkdir -p gpurun_out
bash tools/prof_gaps.sh; head -8 gpurun_out/step_gaps.txt
head -2 /tmp/ps/s_kernel_trace.csv
python3 tools/lab/occupancy_timeline.py /tmp/ps/s_kernel_trace.csv > gpurun_out/r06_occupancy_timeline.txt 2>&1; cat gpurun_out/r06_occupancy_timeline.txt
bash tools/lab/instep_gap.sh > gpurun_out/r06_instep_gap.txt 2>&1; cat gpurun_out/r06_instep_gap.txt
