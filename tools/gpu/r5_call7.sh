#!/bin/bash
# the LayerNorm reverse kernel with its row's stores drained before the next row (-DLN_DRAIN_STORES) vs as shipped, helper streams ON
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
{
echo "== shipped kernel, ALIGNN_AMD_LN_STREAMS=3"
ALIGNN_AMD_LN_STREAMS=3 timeout 300 python tools/ff_repro_check.py 48 c auto 2>&1 | grep "path="
FF=0 ALIGNN_AMD_LN_STREAMS=3 timeout 300 python tools/ff_repro_check.py 64 c auto 2>&1 | grep "path="
echo "== s_waitcnt vmcnt(0) after every row's stores, ALIGNN_AMD_LN_STREAMS=3"
ALIGNN_AMD_LIB_PATH=$PWD/tools/_libalignn_hip_drain.so ALIGNN_AMD_LN_STREAMS=3 timeout 300 python tools/ff_repro_check.py 48 c auto 2>&1 | grep "path="
ALIGNN_AMD_LIB_PATH=$PWD/tools/_libalignn_hip_drain.so FF=0 ALIGNN_AMD_LN_STREAMS=3 timeout 300 python tools/ff_repro_check.py 64 c auto 2>&1 | grep "path="
} > gpurun_out/r5c7_drain.txt 2>&1
cut -c1-200 gpurun_out/r5c7_drain.txt
