#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_cmodel.py tests/test_gpu_cmodel_ff.py -q -rf > gpurun_out/r5c5_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5c5_pytest.log
grep -v "^    \|^E   " gpurun_out/r5c5_pytest.log | tail -6
grep "^E   " gpurun_out/r5c5_pytest.log | head -20
for r in 1 0 1 0; do
ALIGNN_AMD_GRAD_SINK=$r timeout 600 python bench.py --no-cpu-baseline --other-configs 0 --streamed-steps 0 --no-micro > gpurun_out/r5c5_bench_sink$r.json 2> gpurun_out/r5c5_bench_sink$r.err
python - <<PY
import json
d=json.load(open('gpurun_out/r5c5_bench_sink$r.json'))
print('sink=$r headline', d['ms_per_step'], 'replay', d['replayed_steps'], 'eager', d['eager_launches'])
PY
done
