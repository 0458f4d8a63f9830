#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/p_ser
ALIGNN_BENCH_EAGER=1 ALIGNN_AMD_SIDE_STREAM=0 ALIGNN_AMD_LANES=0 timeout 600 rocprofv3 --kernel-trace -d /tmp/p_ser -o r -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --streamed-steps 0 --eager-steps 0 > gpurun_out/c14_ser_bench.json 2> gpurun_out/c14_ser.err
db=$(find /tmp/p_ser -name "*.db" | head -1)
python tools/rocpd_stats.py $db --grid > gpurun_out/c14_ser_by_grid.txt
python tools/rocpd_timeline.py $db 2 > gpurun_out/c14_ser_timeline.txt
head -4 gpurun_out/c14_ser_timeline.txt; head -22 gpurun_out/c14_ser_by_grid.txt | cut -c1-170
