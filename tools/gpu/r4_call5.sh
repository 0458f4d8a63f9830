#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_radius_graph.py tests/test_gpu_stage.py tests/test_gpu_cmodel.py -q -m gpu 2>&1 | tail -30 > gpurun_out/c5_new.log
tail -4 gpurun_out/c5_new.log
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -40 > gpurun_out/c5_all.log
tail -6 gpurun_out/c5_all.log
