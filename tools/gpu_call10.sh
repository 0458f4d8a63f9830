#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export ABL_TN=1 ABL_ONLY=warm,base,pipe ABL_CHECK=1
for i in 1 2 3; do python tools/ablate_x6.py run 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/c10_ablate_tn.txt
