#!/bin/bash
# angle embedding passes at the headline size: per-kernel durations (rocprofv3 --stats) + the tool's own event timing
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf /tmp/p_angle
timeout 300 rocprofv3 --kernel-trace -d /tmp/p_angle -o r -- python tools/angle_time.py > gpurun_out/r6_angle_time.txt 2>&1
db=$(find /tmp/p_angle -name "*.db" | head -1)
python tools/rocpd_stats.py $db | grep -i "angle\|calls" | cut -c1-150 >> gpurun_out/r6_angle_time.txt
cat gpurun_out/r6_angle_time.txt
timeout 600 python -m pytest tests/test_gpu_angle.py -x -q 2>&1 | tail -2
