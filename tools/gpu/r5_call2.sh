#!/bin/bash
# round 5, call 2: the ALIGNNAtomWise C path - bit equality with the per-operator path, the force-field parity tests, cfg 3 and MD timings
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export ALIGNN_AMD_DEBUG=1
timeout 900 python -m pytest tests/test_gpu_cmodel_ff.py tests/test_gpu_cmodel.py tests/test_gpu_dual.py -q -rf > gpurun_out/r5c2_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5c2_pytest.log
grep -v "^    \|^E   " gpurun_out/r5c2_pytest.log | tail -40
grep "^E   " gpurun_out/r5c2_pytest.log | head -40
unset ALIGNN_AMD_DEBUG
timeout 600 python bench.py --model alignn_ff --batch 16 --atoms 200 --steps 5 --warmup 2 --no-cpu-baseline --other-configs 0 --streamed-steps 0 > gpurun_out/r5c2_ff.json 2> gpurun_out/r5c2_ff.err
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/r5c2_ff.json'))
    print('cfg3', d['ms_per_step'], d['step_launch'][:30], 'replay', d['replayed_steps'], 'eager', d['eager_launches'], 'peak', d['peak_hbm_GB'])
except Exception as e:
    print('bench failed', e); print(open('gpurun_out/r5c2_ff.err').read()[-1500:])
PY
timeout 300 python tools/md_step.py > gpurun_out/r5c2_md.txt 2>&1; tail -4 gpurun_out/r5c2_md.txt
