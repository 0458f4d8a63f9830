#!/bin/bash
# what the driver does at round end: GPU tests, smoke(), the default bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/final_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/final_pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; echo "bench rc=$?"; python -c "
import json;d=json.load(open('gpurun_out/final_bench.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'eager', d['eager_launches']['ms_per_step'], 'streamed', d['streamed_batches']['ms_per_step'])
print('roofline', {k: d['roofline'][k] for k in ('achieved','frac','ms_per_launch','traffic')})
print('in_step', {k: (round(v['ms_per_launch'], 4), v.get('frac')) for k, v in d['roofline']['in_step'].items() if isinstance(v, dict)})
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['seconds_per_step'])"
