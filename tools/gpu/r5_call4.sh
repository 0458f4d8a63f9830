#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_cmodel_ff.py tests/test_gpu_dual.py tests/test_gpu_round2.py tests/test_gpu_full_size.py -q -rf > gpurun_out/r5c4_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5c4_pytest.log
grep -v "^    \|^E   " gpurun_out/r5c4_pytest.log | tail -12
grep "^E   " gpurun_out/r5c4_pytest.log | head -20
for v in 0 2 3; do
ALIGNN_AMD_LN_STREAMS=$v timeout 600 python bench.py --model alignn_ff --batch 16 --atoms 200 --steps 5 --warmup 2 --no-cpu-baseline --other-configs 0 --streamed-steps 0 > gpurun_out/r5c4_ff_$v.json 2> gpurun_out/r5c4_ff_$v.err
python - <<PY
import json
d=json.load(open('gpurun_out/r5c4_ff_$v.json'))
print('LN_STREAMS=$v cfg3', d['ms_per_step'], 'replay', d['replayed_steps']['ms_per_step'], 'eager', d['eager_launches']['ms_per_step'], 'host', d['eager_launches']['host_enqueue_ms_per_step'], 'peak', d['peak_hbm_GB'])
PY
done
timeout 300 python tools/md_step.py > gpurun_out/r5c4_md.txt 2>&1; tail -3 gpurun_out/r5c4_md.txt
