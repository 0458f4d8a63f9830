#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
for BA in "1 200" "4 60" "8 60"; do
echo "== batch, atoms: $BA"
timeout 300 python tools/ln_rev_time.py $BA 2>&1 | grep -v amdgpu.ids | cut -c1-170
done
