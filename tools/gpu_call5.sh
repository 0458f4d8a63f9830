#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -rf > gpurun_out/c5_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c5_pytest.log
tail -15 gpurun_out/c5_pytest.log
