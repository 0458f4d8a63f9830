#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
echo "== ablate base(persistent) vs nopersist"; ABL_F16=1 ABL_CHECK=1 ABL_CHECK_REPS=2 ABL_ROUNDS=5 ABL_ONLY=warm,base,nopersist timeout 120 python tools/ablate_x6.py run 2>&1 | grep -v amdgpu.ids | tee gpurun_out/x6p_ablate.txt
for st in 0 20000; do
echo "== family, persistent, stagger $st"; ALIGNN_AMD_X6_STAGGER=$st timeout 300 python tools/x6_family_check.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/x6p_family_$st.txt
done
echo "== family check, one-tile kernels"; ALIGNN_AMD_X6_PERSIST=0 timeout 300 python tools/x6_family_check.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/x6p_family_old.txt
echo "== odd size"; timeout 300 python tools/x6_family_check.py 140011 2>&1 | grep -v amdgpu.ids | tee gpurun_out/x6p_family_odd.txt
