#!/bin/bash
# same-box A/B of the replayed / eagerly launched step: the tree of round 2's final commit (exported to ab_r02/, library built
# there) against the working tree.  usage: bash tools/gpu/ab_r02.sh
for i in 1 2 3; do
  for d in ab_r02 .; do
    (cd $d && python bench.py --no-cpu-baseline --streamed-steps 0 --steps 30 2>/dev/null) | python -c "
import json,sys
d=json.load(sys.stdin); e=d['eager_launches'] or {}
print('$d', 'replay', d['ms_per_step'], 'eager', e.get('ms_per_step'), 'enqueue', e.get('host_enqueue_ms_per_step'))"
  done
done
