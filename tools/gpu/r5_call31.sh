#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
for V in 1 0 1 0 1 0; do
ALIGNN_AMD_LN_FWD_BOND=$V timeout 600 python bench.py --model alignn_ff --batch 16 --atoms 200 --steps 5 --warmup 2 --no-cpu-baseline --other-configs 0 --streamed-steps 0 --no-micro > gpurun_out/r5c31_ff.json 2> gpurun_out/r5c31_ff.err
V=$V python - <<'PY'
import json, os
d=json.load(open('gpurun_out/r5c31_ff.json'))
print('LN_FWD_BOND', os.environ['V'], 'cfg3', d['ms_per_step'], 'replay', d['replayed_steps']['ms_per_step'], 'eager', d['eager_launches']['ms_per_step'], 'host', d['eager_launches'].get('host_enqueue_ms_per_step'))
PY
done
