#!/bin/bash
# round 6: PMC traffic passes (FETCH_SIZE, WRITE_SIZE in separate runs) over the default bench command, and which projection variants a step times
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o r -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --streamed-steps 0 --eager-steps 0 --no-micro --other-configs 0 $BARGS > /dev/null 2> gpurun_out/prof_pmc_$c.err
  db=$(find /tmp/pmc_$c -name "*.db" | head -1)
  python tools/rocpd_pmc.py $db 0 > gpurun_out/prof_pmc${SUFFIX}_$c.txt
  head -16 gpurun_out/prof_pmc${SUFFIX}_$c.txt | cut -c1-200
done
