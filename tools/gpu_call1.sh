#!/bin/bash
# round 2, GPU call 1: full GPU test suite + the default bench line + the non-headline configs
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -rf --durations=8 > gpurun_out/c1_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c1_pytest.log
tail -40 gpurun_out/c1_pytest.log
timeout 600 python bench.py > gpurun_out/c1_bench_default.json 2> gpurun_out/c1_bench_default.err; echo "bench rc=$?"
tail -3 gpurun_out/c1_bench_default.err; cat gpurun_out/c1_bench_default.json
