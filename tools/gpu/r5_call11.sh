#!/bin/bash
# edge LayerNorm inside the gate passes (csrc/convln.hip): parity + reproducibility + A/B timing against ALIGNN_AMD_LN_FUSED=0
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_cmodel_ff.py tests/test_gpu_dual.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_full_size.py -q -rf > gpurun_out/r5c11_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5c11_pytest.log
grep -v "^    \|^E   " gpurun_out/r5c11_pytest.log | tail -8
grep "^E   " gpurun_out/r5c11_pytest.log | head -20
{
for B in 48 64; do timeout 300 python tools/ff_repro_check.py $B c auto 2>&1 | grep "path="; done
FF=0 timeout 300 python tools/ff_repro_check.py 64 c auto 2>&1 | grep "path="
timeout 300 python tools/ff_repro_check.py 48 ops auto 2>&1 | grep "path="
timeout 300 python tools/ff_capture_check.py 48 2>&1 | grep "lanes auto\|forward only\|eval mode"
} > gpurun_out/r5c11_repro.txt 2>&1
cut -c1-150 gpurun_out/r5c11_repro.txt
for F in 1 0 1 0; do
ALIGNN_AMD_LN_FUSED=$F timeout 600 python bench.py --model alignn_ff --batch 16 --atoms 200 --steps 5 --warmup 2 --no-cpu-baseline --other-configs 0 --streamed-steps 0 > gpurun_out/r5c11_ff_$F.json 2> gpurun_out/r5c11_ff_$F.err
F=$F python - <<'PY'
import json, os
f=os.environ['F']
d=json.load(open(f'gpurun_out/r5c11_ff_{f}.json'))
print('LN_FUSED', f, 'cfg3', d['ms_per_step'], 'replay', d['replayed_steps']['ms_per_step'], 'eager', d['eager_launches']['ms_per_step'], 'kernels', d.get('kernels_per_step'))
PY
done
ALIGNN_AMD_LN_FUSED=1 timeout 300 python tools/md_step.py 2>&1 | tail -2
ALIGNN_AMD_LN_FUSED=0 timeout 300 python tools/md_step.py 2>&1 | tail -1
