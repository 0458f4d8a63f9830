#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_dual.py tests/test_gpu_cmodel_ff.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_full_size.py -q -rf -s > gpurun_out/r5c13_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5c13_pytest.log
grep "edge LayerNorm inside" gpurun_out/r5c13_pytest.log
grep -v "^    \|^E   " gpurun_out/r5c13_pytest.log | tail -6
grep "^E   " gpurun_out/r5c13_pytest.log | head -20
timeout 600 python tools/ln_rev_time.py > gpurun_out/r5c13_ln_rev.txt 2>&1
cat gpurun_out/r5c13_ln_rev.txt | cut -c1-200
{
for B in 48 64; do timeout 300 python tools/ff_repro_check.py $B c auto 2>&1 | grep "path="; done
timeout 300 python tools/ff_repro_check.py 48 ops auto 2>&1 | grep "path="
timeout 300 python tools/ff_capture_check.py 48 2>&1 | grep "lanes auto\|forward only\|eval mode"
} > gpurun_out/r5c13_repro.txt 2>&1
grep -c " 0 tensors differ\|differing tensors: 0" gpurun_out/r5c13_repro.txt; grep -v " 0 tensors differ\|differing tensors: 0" gpurun_out/r5c13_repro.txt | head
for V in 00 20; do
ALIGNN_AMD_LN_REV=$V timeout 600 python bench.py --model alignn_ff --batch 16 --atoms 200 --steps 5 --warmup 2 --no-cpu-baseline --other-configs 0 --streamed-steps 0 --no-micro > gpurun_out/r5c13_ff_$V.json 2> gpurun_out/r5c13_ff_$V.err
V=$V python - <<'PY'
import json, os
f=os.environ['V']
d=json.load(open(f'gpurun_out/r5c13_ff_{f}.json'))
print('LN_REV', f, 'cfg3', d['ms_per_step'], 'replay', d['replayed_steps']['ms_per_step'], 'eager', d['eager_launches']['ms_per_step'])
PY
done
