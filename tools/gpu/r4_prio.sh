#!/bin/bash
export TMPDIR=/tmp
run() { label=$1; shift
  env "$@" ALIGNN_BENCH_EAGER=1 timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-micro --streamed-steps 0 --eager-steps 0 --other-configs 0 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print('$label', 'eager ms', d['ms_per_step'], 'enqueue', d['host_enqueue_ms_per_step'])"
}
run "main prio 0 " A=1
run "main prio -1" ALIGNN_BENCH_MAIN_PRIORITY=-1
run "main prio 0 " A=1
run "main prio -1" ALIGNN_BENCH_MAIN_PRIORITY=-1
