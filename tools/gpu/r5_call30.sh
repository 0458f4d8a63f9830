#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_cmodel_ff.py tests/test_gpu_dual.py tests/test_gpu_round2.py tests/test_gpu_full_size.py -q -rf > gpurun_out/r5c30_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5c30_pytest.log
grep -v "^    \|^E   " gpurun_out/r5c30_pytest.log | tail -3
grep "^E   " gpurun_out/r5c30_pytest.log | head -20
{
for B in 16 48 64; do timeout 300 python tools/ff_repro_check.py $B c auto 2>&1 | grep "path="; done
timeout 300 python tools/ff_capture_check.py 48 2>&1 | grep "lanes auto\|forward only\|eval mode"
} > gpurun_out/r5c30_repro.txt 2>&1
grep -c " 0 tensors differ\|differing tensors: 0" gpurun_out/r5c30_repro.txt; grep -v " 0 tensors differ\|differing tensors: 0" gpurun_out/r5c30_repro.txt | head
for i in 1 2 3; do
timeout 600 python bench.py --model alignn_ff --batch 16 --atoms 200 --steps 5 --warmup 2 --no-cpu-baseline --other-configs 0 --streamed-steps 0 --no-micro > gpurun_out/r5c30_ff.json 2> gpurun_out/r5c30_ff.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5c30_ff.json'))
print('cfg3', d['ms_per_step'], 'replay', d['replayed_steps']['ms_per_step'], 'eager', d['eager_launches']['ms_per_step'], 'host', d['eager_launches'].get('host_enqueue_ms_per_step'))
PY
done
