#!/bin/bash
# kernel trace of force training (BASELINE configs[3]) through the whole-model C calls: eagerly launched steps
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/p_ff
ALIGNN_BENCH_EAGER=1 timeout 900 rocprofv3 --kernel-trace -d /tmp/p_ff -o r -- python bench.py --model alignn_ff --batch 16 --atoms 200 --steps 5 --warmup 2 --no-cpu-baseline --streamed-steps 0 --eager-steps 0 --no-micro --other-configs 0 > gpurun_out/prof_cfg4_ff_bench.json 2> gpurun_out/prof_cfg4_ff.err
db=$(find /tmp/p_ff -name "*.db" | head -1)
python tools/rocpd_stats.py $db > gpurun_out/prof_cfg4_ff_kernel_stats.txt
python tools/rocpd_stats.py $db --grid > gpurun_out/prof_cfg4_ff_kernel_stats_by_grid.txt
python tools/rocpd_timeline.py $db 2 > gpurun_out/prof_cfg4_ff_timeline.txt
python tools/rocpd_sequence.py $db 2 > gpurun_out/prof_cfg4_ff_sequence.txt
head -45 gpurun_out/prof_cfg4_ff_kernel_stats.txt | cut -c1-170
sed -n 1,12p gpurun_out/prof_cfg4_ff_timeline.txt
