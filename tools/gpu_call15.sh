#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
for mt in 256 128 256 128; do
ALIGNN_AMD_X6_MIN_TILES=$mt timeout 300 python bench.py --no-cpu-baseline --streamed-steps 0 --steps 20 > gpurun_out/c15_$mt.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/c15_$mt.json'));print('min_tiles $mt replay', d['ms_per_step'], 'eager', d['eager_launches']['ms_per_step'])"
done
