#!/bin/bash
# MD step: does lane T pay at 35 k triplets?
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
for R in 131072 16384 131072 16384; do
echo "lane_min_rows $R"; ALIGNN_AMD_LANE_MIN_ROWS=$R timeout 300 python tools/md_step.py 2>&1 | tail -2
done
echo "side rows 1024 + lanes 16384"; ALIGNN_AMD_LANE_MIN_ROWS=16384 ALIGNN_AMD_LN_STREAMS=1 timeout 300 python tools/md_step.py 2>&1 | tail -2
