#!/bin/bash
# the fixed build (norm.hip / dual.hip without SLP vectorisation), LayerNorm flavour back on helper streams: tests + reproducibility + timings
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_cmodel_ff.py tests/test_gpu_dual.py tests/test_gpu_round2.py tests/test_gpu_full_size.py tests/test_gpu_kernels.py -q -rf > gpurun_out/r5c10_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5c10_pytest.log
grep -v "^    \|^E   " gpurun_out/r5c10_pytest.log | tail -6
grep "^E   " gpurun_out/r5c10_pytest.log | head -20
{
for B in 48 64 96; do timeout 300 python tools/ff_repro_check.py $B c auto 2>&1 | grep "path="; done
FF=0 timeout 300 python tools/ff_repro_check.py 64 c auto 2>&1 | grep "path="
FF=0 timeout 300 python tools/ff_repro_check.py 64 ops auto 2>&1 | grep "path="
timeout 300 python tools/ff_repro_check.py 48 ops auto 2>&1 | grep "path="
timeout 300 python tools/ff_capture_check.py 48 2>&1 | grep "lanes auto\|forward only\|eval mode"
} > gpurun_out/r5c10_repro.txt 2>&1
cut -c1-150 gpurun_out/r5c10_repro.txt
timeout 600 python bench.py --model alignn_ff --batch 16 --atoms 200 --steps 5 --warmup 2 --no-cpu-baseline --other-configs 0 --streamed-steps 0 > gpurun_out/r5c10_ff.json 2> gpurun_out/r5c10_ff.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5c10_ff.json'))
print('cfg3', d['ms_per_step'], 'replay', d['replayed_steps'], 'eager', d['eager_launches'])
PY
timeout 300 python tools/md_step.py 2>&1 | tail -2
