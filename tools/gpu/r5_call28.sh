#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -q -rf > gpurun_out/r5c28_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5c28_pytest.log
grep -v "^    \|^E   " gpurun_out/r5c28_pytest.log | tail -4
grep "^E   " gpurun_out/r5c28_pytest.log | head -20
timeout 300 python tools/ff_repro_check.py 48 c auto 2>&1 | grep "path=" | cut -c1-100
timeout 300 python tools/md_step.py 2>&1 | tail -2
for i in 1 2 3; do
timeout 600 python bench.py --model alignn_ff --batch 16 --atoms 200 --steps 5 --warmup 2 --no-cpu-baseline --other-configs 0 --streamed-steps 0 --no-micro > gpurun_out/r5c28_ff.json 2> gpurun_out/r5c28_ff.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5c28_ff.json'))
print('cfg3', d['ms_per_step'], 'replay', d['replayed_steps']['ms_per_step'], 'eager', d['eager_launches']['ms_per_step'])
PY
done
