#!/bin/bash
# round 6 working profile: kernel traces of the default bench command (replayed steps), of eager steps, and of the serialized one-stream
# form (durations add up), + the force-training step
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
for n in ${SET:-default serialized cfg4_ff}; do
  case $n in
    default) BARGS="" prof default A=1;;
    eager) BARGS="" prof eager ALIGNN_BENCH_EAGER=1;;
    serialized) BARGS="" prof serialized ALIGNN_BENCH_EAGER=1 ALIGNN_AMD_SIDE_STREAM=0 ALIGNN_AMD_LANES=0 ALIGNN_AMD_FORK=0;;
    cfg4_ff) BARGS="--model alignn_ff --batch 16 --atoms 200" prof cfg4_ff ALIGNN_BENCH_EAGER=1;;
    cfg4_ff_serialized) BARGS="--model alignn_ff --batch 16 --atoms 200" prof cfg4_ff_serialized ALIGNN_BENCH_EAGER=1 ALIGNN_AMD_SIDE_STREAM=0 ALIGNN_AMD_LANES=0 ALIGNN_AMD_FORK=0;;
  esac
done
