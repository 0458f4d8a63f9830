#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
{
echo "== one stream (default)"
for B in 16 48 64; do timeout 300 python tools/ff_repro_check.py $B c auto 2>&1 | grep "path="; done
FF=0 timeout 300 python tools/ff_repro_check.py 64 c auto 2>&1 | grep "path="
echo "== side stream only (ALIGNN_AMD_LN_STREAMS=2)"
ALIGNN_AMD_LN_STREAMS=2 timeout 300 python tools/ff_repro_check.py 48 c auto 2>&1 | grep "path="
ALIGNN_AMD_LN_STREAMS=2 timeout 300 python tools/ff_repro_check.py 64 c auto 2>&1 | grep "path="
FF=0 ALIGNN_AMD_LN_STREAMS=2 timeout 300 python tools/ff_repro_check.py 64 c auto 2>&1 | grep "path="
echo "== per-operator path, side stream + lanes as before round 5 (energy only)"
FF=0 timeout 300 python tools/ff_repro_check.py 64 ops auto 2>&1 | grep "path="
FF=1 timeout 300 python tools/ff_repro_check.py 64 ops auto 2>&1 | grep "path="
} > gpurun_out/r5c3_repro.txt 2>&1
cut -c1-200 gpurun_out/r5c3_repro.txt
timeout 300 python tools/ff_capture_check.py 48 2>&1 | grep "lanes auto\|forward only\|eval mode" | cut -c1-160
