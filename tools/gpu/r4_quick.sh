#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_stage.py -x -q -m gpu 2>&1 | tail -5
timeout 300 python tools/stage_time.py 2>&1 | tail -4
