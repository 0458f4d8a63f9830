#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dual.py tests/test_gpu_full_size.py tests/test_gpu_round2.py tests/test_gpu_model.py -q -rf -k "dual or cfg4 or force or atomwise or ealignn or cutoff or forward_over" 2>&1 | tail -15
timeout 600 python bench.py --model alignn_ff --batch 16 --atoms 200 --no-cpu-baseline --streamed-steps 0 --steps 5 --warmup 2 > gpurun_out/c8_cfg4_ff.json 2> gpurun_out/c8_cfg4_ff.err; tail -2 gpurun_out/c8_cfg4_ff.err; python -c "import json;d=json.load(open('gpurun_out/c8_cfg4_ff.json'));print('cfg4 ff', d['ms_per_step'], d['value'], d['peak_hbm_GB'])"
cat gpurun_out/parity_full_cfg4.txt
