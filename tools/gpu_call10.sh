#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export ABL_F16=1 ABL_ONLY=warm,base,nosync
ABL_CHECK=1 ABL_ROUNDS=7 python tools/ablate_x6.py run 2>&1 | tee gpurun_out/c10_ablate.txt | tail -8
