#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dual.py tests/test_gpu_full_size.py tests/test_gpu_round2.py -q -k "dual or dense or cfg4 or force or tangent or training_loop" 2>&1 | tail -12 > gpurun_out/ff_tests.log
tail -6 gpurun_out/ff_tests.log
for i in 1 2; do
timeout 400 python bench.py --model alignn_ff --batch 16 --atoms 200 --steps 10 --warmup 3 --no-cpu-baseline --no-micro --streamed-steps 0 --eager-steps 0 --other-configs 0 2>gpurun_out/ff_bench.err | python -c "
import json,sys
d=json.load(sys.stdin); print('cfg3 ff', d['ms_per_step'], d['value'], d['step_launch'][:20], d['loss'])"
done
tail -3 gpurun_out/ff_bench.err
