#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_model.py -q -rf -k "training_loop" -s 2>&1 | grep -v "^    " | tail -25
