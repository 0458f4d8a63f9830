#!/bin/bash
# round-4 profile set (copied into profiles/r04_*): kernel traces of the default bench command (timed steps = hipGraph
# replays), of eagerly launched steps on four streams (what a training loop gets: two C calls per step) and on ONE stream
# (durations add up to the step), per-step launch sequences, PMC traffic passes, and the force-training step
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
prof() { # name, env..., BARGS in the environment
  name=$1; shift
  rm -rf /tmp/p_$name
  env "$@" timeout 600 rocprofv3 --kernel-trace -d /tmp/p_$name -o r -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --streamed-steps 0 --eager-steps 0 --no-micro --other-configs 0 $BARGS > gpurun_out/prof_${name}_bench.json 2> gpurun_out/prof_${name}.err
  db=$(find /tmp/p_$name -name "*.db" | head -1)
  python tools/rocpd_stats.py $db > gpurun_out/prof_${name}_kernel_stats.txt
  python tools/rocpd_stats.py $db --grid > gpurun_out/prof_${name}_kernel_stats_by_grid.txt
  python tools/rocpd_timeline.py $db 2 > gpurun_out/prof_${name}_timeline.txt
  python tools/rocpd_sequence.py $db 2 > gpurun_out/prof_${name}_sequence.txt
  sed -n 2,4p gpurun_out/prof_${name}_timeline.txt
}
BARGS="" prof default A=1
BARGS="" prof eager ALIGNN_BENCH_EAGER=1
BARGS="" prof serialized ALIGNN_BENCH_EAGER=1 ALIGNN_AMD_SIDE_STREAM=0 ALIGNN_AMD_LANES=0 ALIGNN_AMD_FORK=0
BARGS="--model alignn_ff --batch 16 --atoms 200" prof cfg4_ff ALIGNN_AMD_SIDE_STREAM=0
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o r -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --streamed-steps 0 --eager-steps 0 --no-micro --other-configs 0 > /dev/null 2> gpurun_out/prof_pmc_$c.err
  db=$(find /tmp/pmc_$c -name "*.db" | head -1)
  python tools/rocpd_pmc.py $db 0 > gpurun_out/prof_pmc_$c.txt
  head -8 gpurun_out/prof_pmc_$c.txt
done
