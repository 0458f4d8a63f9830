#!/bin/bash
# kernel trace of the replayed default step only: timeline summary (who runs how long beside whom)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf /tmp/p_rt
env "$@" timeout 600 rocprofv3 --kernel-trace -d /tmp/p_rt -o r -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --streamed-steps 0 --eager-steps 0 --no-micro > gpurun_out/rt_bench.json 2> gpurun_out/rt.err
db=$(find /tmp/p_rt -name "*.db" | head -1)
python tools/rocpd_timeline.py $db 2 > gpurun_out/rt_timeline.txt
python tools/rocpd_stats.py $db --grid > gpurun_out/rt_by_grid.txt
cut -c1-150 gpurun_out/rt_timeline.txt | sed -n 2,9p; sed -n '/big kernels/,$p' gpurun_out/rt_timeline.txt | cut -c1-140
