#!/bin/bash
# round 4, call 2: stream configurations of the eagerly launched C path, host profile, replay timelines of both paths
export TMPDIR=/tmp
mkdir -p gpurun_out
run() { # label, env...
  label=$1; shift
  env "$@" ALIGNN_BENCH_EAGER=1 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-micro --streamed-steps 0 --eager-steps 0 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print('$label', 'eager ms', d['ms_per_step'], 'enqueue', d['host_enqueue_ms_per_step'])"
}
{
run "lanes1 fork1 side1" A=1
run "lanes0 fork0 side1" ALIGNN_AMD_LANES=0 ALIGNN_AMD_FORK=0
run "lanes0 fork0 side0" ALIGNN_AMD_LANES=0 ALIGNN_AMD_FORK=0 ALIGNN_AMD_SIDE_STREAM=0
run "lanes1 fork0 side1" ALIGNN_AMD_FORK=0
run "lanes0 fork1 side1" ALIGNN_AMD_LANES=0
run "ops path (python)" ALIGNN_AMD_CMODEL=0
} > gpurun_out/c2_streams.txt 2>&1
cat gpurun_out/c2_streams.txt
timeout 300 python tools/host_profile_c.py 64 > gpurun_out/c2_host_profile.txt 2>&1
head -40 gpurun_out/c2_host_profile.txt
prof() { # name, env...
  name=$1; shift
  rm -rf /tmp/p_$name
  env "$@" timeout 600 rocprofv3 --kernel-trace -d /tmp/p_$name -o r -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --streamed-steps 0 --eager-steps 0 --no-micro > gpurun_out/c2_${name}_bench.json 2> gpurun_out/c2_${name}.err
  db=$(find /tmp/p_$name -name "*.db" | head -1)
  python tools/rocpd_stats.py $db --grid > gpurun_out/c2_${name}_by_grid.txt
  python tools/rocpd_timeline.py $db 2 > gpurun_out/c2_${name}_timeline.txt
  head -4 gpurun_out/c2_${name}_timeline.txt | cut -c1-200
}
prof cmodel A=1
prof ops ALIGNN_AMD_CMODEL=0
