#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_gpu_angle.py -x -q -m gpu 2>&1 | tail -12
rm -rf /tmp/p_angle
timeout 300 rocprofv3 --kernel-trace -d /tmp/p_angle -o r -- python tools/angle_time.py > gpurun_out/angle_time.txt 2> gpurun_out/angle_prof.err
cat gpurun_out/angle_time.txt
db=$(find /tmp/p_angle -name "*.db" | head -1)
python tools/rocpd_stats.py $db > gpurun_out/angle_kernel_stats.txt
head -16 gpurun_out/angle_kernel_stats.txt | cut -c1-150
