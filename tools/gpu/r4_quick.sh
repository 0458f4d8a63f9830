#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_cmodel.py -q 2>&1 | tail -8
