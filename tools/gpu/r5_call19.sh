#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
for V in 0 1 0 1; do
ALIGNN_AMD_TN_ON_T=$V timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --other-configs 0 --streamed-steps 0 --no-micro > gpurun_out/r5c19_$V.json 2> gpurun_out/r5c19_$V.err
V=$V python - <<'PY'
import json, os
f=os.environ['V']
d=json.load(open(f'gpurun_out/r5c19_{f}.json'))
print('TN_ON_T', f, 'headline', d['ms_per_step'], 'replay', d['replayed_steps']['ms_per_step'], 'eager', d['eager_launches']['ms_per_step'])
PY
done
