#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dual.py -q -x -rf 2>&1 | tail -40
