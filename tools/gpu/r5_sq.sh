#!/bin/bash
# SQ counter pass over ONE serialized, eagerly launched training step of the headline workload (every kernel as a step launches it)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/sq5_$i
  ALIGNN_BENCH_EAGER=1 ALIGNN_AMD_SIDE_STREAM=0 ALIGNN_AMD_LANES=0 ALIGNN_AMD_FORK=0 timeout 600 rocprofv3 --kernel-trace --pmc $set -d /tmp/sq5_$i -o r -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --streamed-steps 0 --eager-steps 0 --no-micro --other-configs 0 > /dev/null 2> gpurun_out/sq5_$i.err
  db=$(find /tmp/sq5_$i -name "*.db" | head -1)
  if [ -n "$db" ]; then python tools/rocpd_sq.py $db 100 > gpurun_out/r5_sq_pass$i.txt; else tail -5 gpurun_out/sq5_$i.err; fi
  cut -c1-190 gpurun_out/r5_sq_pass$i.txt | head -40
done
