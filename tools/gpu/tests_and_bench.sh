#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -rf > gpurun_out/tb_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/tb_pytest.log
grep -v "^    \|^E   " gpurun_out/tb_pytest.log | tail -15
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/tb_bench.json 2> gpurun_out/tb_bench.err; python -c "import json;d=json.load(open('gpurun_out/tb_bench.json'));print('replay', d['ms_per_step'], 'eager', d['eager_launches'], 'streamed', d['streamed_batches']['ms_per_step'])"
python -c "
import json;d=json.load(open('gpurun_out/tb_bench.json'));r=d['roofline']
print('roofline', r['achieved'], r['frac'], r['ms_per_launch']); print({k:(round(v['ms_per_launch'],4), v.get('frac')) for k,v in r['in_step'].items() if isinstance(v,dict)})"
ALIGNN_AMD_X6_PERSIST=0 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/tb_bench_onetile.json 2> gpurun_out/tb_bench_onetile.err; python -c "import json;d=json.load(open('gpurun_out/tb_bench_onetile.json'));print('one-tile kernels: replay', d['ms_per_step'], 'eager', d['eager_launches']['ms_per_step'])"
