#!/bin/bash
# launch variants of the two reverse kernels with the LayerNorm inside, kernel level + whole step
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 600 python tools/ln_rev_time.py > gpurun_out/r5c12_ln_rev.txt 2>&1
cat gpurun_out/r5c12_ln_rev.txt | cut -c1-200
for V in 00 05 06 05 06; do
ALIGNN_AMD_LN_REV=$V timeout 600 python bench.py --model alignn_ff --batch 16 --atoms 200 --steps 5 --warmup 2 --no-cpu-baseline --other-configs 0 --streamed-steps 0 --no-micro > gpurun_out/r5c12_ff_$V.json 2> gpurun_out/r5c12_ff_$V.err
V=$V python - <<'PY'
import json, os
f=os.environ['V']
d=json.load(open(f'gpurun_out/r5c12_ff_{f}.json'))
print('LN_REV', f, 'cfg3', d['ms_per_step'], 'replay', d['replayed_steps']['ms_per_step'], 'eager', d['eager_launches']['ms_per_step'])
PY
done
