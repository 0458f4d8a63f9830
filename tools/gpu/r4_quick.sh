#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_stage.py tests/test_graph_builder_golden.py tests/test_radius_graph.py tests/test_gpu_round3.py -q -m gpu -k "stage or crystal or graphed or hip or radius or device" 2>&1 | tail -6
timeout 300 python tools/md_step.py 2>&1 | tail -4
