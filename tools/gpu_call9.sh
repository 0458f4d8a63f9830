#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
timeout 900 python -m pytest tests/test_gpu_ddp.py tests/test_gpu_round2.py tests/test_graph_builder_golden.py -m gpu -q -rf 2>&1 | grep -v "^    \|^E   " | tail -12
