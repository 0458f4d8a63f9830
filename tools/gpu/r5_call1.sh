#!/bin/bash
# round 5, call 1: the whole GPU suite on the tree with the ADVICE fixes + new parity tests, then baselines of this box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -rf > gpurun_out/r5c1_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5c1_pytest.log
grep -v "^    \|^E   " gpurun_out/r5c1_pytest.log | tail -25
timeout 900 python bench.py --no-cpu-baseline --other-configs 1 > gpurun_out/r5c1_bench.json 2> gpurun_out/r5c1_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5c1_bench.json'))
print('headline', d['ms_per_step'], d['step_launch'][:20], 'replay', d['replayed_steps'], 'eager', d['eager_launches'])
r=d['roofline']; print('roofline', r['achieved'], r['frac'], r['traffic'])
for k,v in d.get('other_configs',{}).items(): print(k, v.get('ms_per_step'), v.get('eager_launches'), v.get('replayed_steps'))
PY
timeout 300 python tools/md_step.py > gpurun_out/r5c1_md.txt 2>&1; tail -5 gpurun_out/r5c1_md.txt
