#!/bin/bash
# same-box A/B of the replayed / eagerly launched step: the tree of round 2's final commit (exported to ab_r02/, library built
# there) against the working tree.  usage: bash tools/gpu/ab_r02.sh
# To make ab_r02/ (scratch, never committed; GIT_INDEX_FILE keeps the real index untouched):
#   mkdir ab_r02 && GIT_INDEX_FILE=/tmp/ab_r02.index git --work-tree=ab_r02 checkout 81cec8f -- . \
#     && (cd ab_r02 && rm -rf profiles tests/golden && python -c "from alignn_amd import build; build.build()")
for i in 1 2 3; do
  for d in ab_r02 .; do
    (cd $d && python bench.py --no-cpu-baseline --streamed-steps 0 --steps 30 2>/dev/null) | python -c "
import json,sys
d=json.load(sys.stdin); e=d['eager_launches'] or {}
print('$d', 'replay', d['ms_per_step'], 'eager', e.get('ms_per_step'), 'enqueue', e.get('host_enqueue_ms_per_step'))"
  done
done
