#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
echo "== library: shipped"
for B in 64 96; do timeout 300 python tools/ff_repro_check.py $B c auto 2>&1 | grep "run" | cut -c1-200; done
for only in norm dual convln; do
  echo "== SLP only in $only.hip"
  for B in 64 96; do ALIGNN_AMD_LIB_PATH=$PWD/tools/_libalignn_hip_slp_$only.so timeout 300 python tools/ff_repro_check.py $B c auto 2>&1 | grep "run" | cut -c1-200; done
done
echo "== SLP in all three, helper streams off (ALIGNN_AMD_LN_STREAMS=0)"
ALIGNN_AMD_LN_STREAMS=0 ALIGNN_AMD_LIB_PATH=$PWD/tools/_libalignn_hip_slp.so timeout 300 python tools/ff_repro_check.py 96 c auto 2>&1 | grep "run" | cut -c1-200
