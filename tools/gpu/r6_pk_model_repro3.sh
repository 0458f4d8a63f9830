#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
L=$PWD/tools/_libalignn_hip_slp_norm.so
run() { echo "== $*"; env "$@" ALIGNN_AMD_LIB_PATH=$L timeout 300 python tools/ff_repro_check.py 96 c $LANES 2>&1 | grep "run" | cut -c1-160; }
LANES=auto run A=1
LANES=0 run A=1
LANES=auto run ALIGNN_AMD_FORK=0
LANES=0 run ALIGNN_AMD_FORK=0
LANES=auto run ALIGNN_AMD_LN_STREAMS=1
LANES=auto run ALIGNN_AMD_LN_STREAMS=2
LANES=auto run ALIGNN_AMD_SIDE_STREAM=0
LANES=auto run AMD_SERIALIZE_KERNEL=3
LANES=auto run ALIGNN_AMD_DW_FUSED=0
