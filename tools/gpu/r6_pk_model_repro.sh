#!/bin/bash
# does the round-5 fault (run-to-run different force-training steps with the SLP-vectorised LayerNorm kernels) still reproduce
# at model level?  shipped library vs a build of norm.hip / dual.hip / convln.hip WITHOUT -fno-slp-vectorize
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
for lib in "" tools/_libalignn_hip_slp.so; do
  echo "== library: ${lib:-shipped}"
  for B in 48 64 96; do
    for path in c ops; do
      ALIGNN_AMD_LIB_PATH=${lib:+$PWD/$lib} timeout 300 python tools/ff_repro_check.py $B $path auto 2>&1 | grep "run" | cut -c1-220
    done
  done
  FF=0 ALIGNN_AMD_LIB_PATH=${lib:+$PWD/$lib} timeout 300 python tools/ff_repro_check.py 64 ops 1 2>&1 | grep "run" | cut -c1-220
done
