#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_round3.py tests/test_gpu_stage.py tests/test_gpu_round2.py -q -rf -k "knn or neighbor or neighbour or graph or md or stage" > gpurun_out/r5c20_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5c20_pytest.log
grep -v "^    \|^E   " gpurun_out/r5c20_pytest.log | tail -5
grep "^E   " gpurun_out/r5c20_pytest.log | head -20
timeout 300 python tools/md_step.py 2>&1 | tail -3
timeout 300 python tools/md_step.py 2>&1 | tail -3
