#!/bin/bash
# SQ counter passes (wave cycles, wait buckets, MFMA busy, LDS) over the T-row split-product kernels
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*\|GRBM_[A-Z_]*\|TCP_[A-Z0-9_]*\|TA_[A-Z0-9_]*" | sort -u > gpurun_out/sq_counter_names.txt
wc -l gpurun_out/sq_counter_names.txt
TARGET=${TARGET:-tools/x6_family_once.py}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAVES SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"; do
  i=$((i+1))
  rm -rf /tmp/sq_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/sq_$i -o r -- python $TARGET > /dev/null 2> gpurun_out/sq_$i.err
  db=$(find /tmp/sq_$i -name "*.db" | head -1)
  if [ -n "$db" ]; then python tools/rocpd_raw.py $db gemm_ > gpurun_out/sq_pass$i.txt; else tail -5 gpurun_out/sq_$i.err; fi
  cat gpurun_out/sq_pass$i.txt | cut -c1-140
done
