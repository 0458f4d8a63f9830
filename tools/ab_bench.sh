#!/bin/bash
# A/B of one environment switch on ONE box (box-to-box differences exceed most effects): tools/ab_bench.sh VAR [bench args]
# runs bench.py with VAR=1, VAR=0, VAR=1, VAR=0 and prints replayed / eager ms per step of each.
var=$1; shift
for v in 1 0 1 0; do
  env $var=$v python bench.py --no-cpu-baseline --streamed-steps 0 --steps 30 "$@" 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); e=d['eager_launches'] or {}
print('$var=$v', 'replay', d['ms_per_step'], 'eager', e.get('ms_per_step'), 'enqueue', e.get('host_enqueue_ms_per_step'))"
done
