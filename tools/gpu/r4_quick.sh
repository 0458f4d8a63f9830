#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_cmodel.py tests/test_gpu_model.py -q 2>&1 | tail -8
timeout 300 python tools/infer_time.py 2>&1 | tail -3
