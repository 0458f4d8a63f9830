#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_kernels.py tests/test_gpu_full_size.py -q -rf 2>&1 | grep -v "^    \|^E   " | tail -12
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/c13_bench.json 2> gpurun_out/c13_bench.err; python -c "import json;d=json.load(open('gpurun_out/c13_bench.json'));print('replay', d['ms_per_step'], 'eager', d['eager_launches'], 'streamed', d['streamed_batches']['ms_per_step'])"
