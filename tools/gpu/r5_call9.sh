#!/bin/bash
# packed fp32 (SLP-vectorised) code off: whole library / norm.hip + dual.hip only / as shipped - headline and force training, helper streams ON for the LayerNorm flavour
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
run() { # tag, lib
  tag=$1; lib=$2
  ALIGNN_AMD_LIB_PATH=$lib timeout 600 python bench.py --no-cpu-baseline --other-configs 0 --streamed-steps 0 --no-micro > gpurun_out/r5c9_$tag.json 2> gpurun_out/r5c9_$tag.err
  ALIGNN_AMD_LIB_PATH=$lib ALIGNN_AMD_LN_STREAMS=3 timeout 600 python bench.py --model alignn_ff --batch 16 --atoms 200 --steps 5 --warmup 2 --no-cpu-baseline --other-configs 0 --streamed-steps 0 --no-micro > gpurun_out/r5c9_${tag}_ff3.json 2> gpurun_out/r5c9_${tag}_ff3.err
  ALIGNN_AMD_LIB_PATH=$lib ALIGNN_AMD_LN_STREAMS=0 timeout 600 python bench.py --model alignn_ff --batch 16 --atoms 200 --steps 5 --warmup 2 --no-cpu-baseline --other-configs 0 --streamed-steps 0 --no-micro > gpurun_out/r5c9_${tag}_ff0.json 2> gpurun_out/r5c9_${tag}_ff0.err
  python - <<PY
import json
d=json.load(open('gpurun_out/r5c9_$tag.json')); f3=json.load(open('gpurun_out/r5c9_${tag}_ff3.json')); f0=json.load(open('gpurun_out/r5c9_${tag}_ff0.json'))
print('$tag: headline eager', d['eager_launches']['ms_per_step'], 'replay', d['replayed_steps']['ms_per_step'], '| cfg3 streams on: eager', f3['eager_launches']['ms_per_step'], 'replay', f3['replayed_steps']['ms_per_step'], '| cfg3 one stream: eager', f0['eager_launches']['ms_per_step'], 'replay', f0['replayed_steps']['ms_per_step'])
PY
}
run shipped $PWD/alignn_amd/libalignn_hip.so
run noslp_ln $PWD/tools/_lib_noslp_ln.so
run noslp_all $PWD/tools/_lib_noslp_all.so
run shipped2 $PWD/alignn_amd/libalignn_hip.so
for lib in noslp_ln noslp_all; do
ALIGNN_AMD_LIB_PATH=$PWD/tools/_lib_$lib.so ALIGNN_AMD_LN_STREAMS=3 timeout 300 python tools/ff_repro_check.py 64 c auto 2>&1 | grep "path=" | cut -c1-150
done
