#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_cmodel_ff.py tests/test_gpu_cmodel.py -q -rf -s > gpurun_out/r5c18_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5c18_pytest.log
grep "gradients in place" gpurun_out/r5c18_pytest.log
grep -v "^    \|^E   " gpurun_out/r5c18_pytest.log | tail -5
grep "^E   " gpurun_out/r5c18_pytest.log | head -20
for G in 1 0 1 0; do
ALIGNN_AMD_GRAD_SINK=$G timeout 600 python bench.py --model alignn_ff --batch 16 --atoms 200 --steps 5 --warmup 2 --no-cpu-baseline --other-configs 0 --streamed-steps 0 --no-micro > gpurun_out/r5c18_ff_$G.json 2> gpurun_out/r5c18_ff_$G.err
G=$G python - <<'PY'
import json, os
f=os.environ['G']
d=json.load(open(f'gpurun_out/r5c18_ff_{f}.json'))
print('GRAD_SINK', f, 'cfg3', d['ms_per_step'], 'replay', d['replayed_steps']['ms_per_step'], 'eager', d['eager_launches']['ms_per_step'], 'host', d['eager_launches'].get('host_enqueue_ms_per_step'))
PY
done
