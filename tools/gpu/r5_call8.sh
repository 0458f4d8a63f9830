#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
{
for v in 9 6; do
echo "== norm.hip variant $v (9: -O3 -fno-slp-vectorize = no packed fp32 instructions; 6: -O1), helper streams ON (ALIGNN_AMD_LN_STREAMS=3)"
ALIGNN_AMD_LIB_PATH=$PWD/tools/_lib_v$v.so ALIGNN_AMD_LN_STREAMS=3 timeout 300 python tools/ff_repro_check.py 48 c auto 2>&1 | grep "path="
ALIGNN_AMD_LIB_PATH=$PWD/tools/_lib_v$v.so ALIGNN_AMD_LN_STREAMS=3 timeout 300 python tools/ff_repro_check.py 64 c auto 2>&1 | grep "path="
ALIGNN_AMD_LIB_PATH=$PWD/tools/_lib_v$v.so ALIGNN_AMD_LN_STREAMS=3 FF=0 timeout 300 python tools/ff_repro_check.py 64 c auto 2>&1 | grep "path="
ALIGNN_AMD_LIB_PATH=$PWD/tools/_lib_v$v.so ALIGNN_AMD_LN_STREAMS=3 timeout 300 python tools/ff_capture_check.py 48 2>&1 | grep "lanes auto\|forward only\|eval mode" | cut -c1-150
done
} > gpurun_out/r5c8_variants2.txt 2>&1
cut -c1-170 gpurun_out/r5c8_variants2.txt
