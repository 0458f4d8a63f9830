#!/bin/bash
# kernel sequences of one step of the C path: serialized (one stream, eager) and replayed (default)
export TMPDIR=/tmp
mkdir -p gpurun_out
seq() { # name, env...
  name=$1; shift
  rm -rf /tmp/p_$name
  env "$@" timeout 600 rocprofv3 --kernel-trace -d /tmp/p_$name -o r -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --streamed-steps 0 --eager-steps 0 --no-micro --other-configs 0 > gpurun_out/seq_${name}_bench.json 2> gpurun_out/seq_${name}.err
  db=$(find /tmp/p_$name -name "*.db" | head -1)
  python tools/rocpd_sequence.py $db 2 > gpurun_out/seq_${name}.txt
  python tools/rocpd_timeline.py $db 2 > gpurun_out/seq_${name}_timeline.txt
  python tools/rocpd_stats.py $db --grid > gpurun_out/seq_${name}_by_grid.txt
  head -1 gpurun_out/seq_${name}.txt; sed -n 2,4p gpurun_out/seq_${name}_timeline.txt
}
seq serialized ALIGNN_BENCH_EAGER=1 ALIGNN_AMD_SIDE_STREAM=0 ALIGNN_AMD_LANES=0 ALIGNN_AMD_FORK=0
seq replay A=1
seq eager ALIGNN_BENCH_EAGER=1
