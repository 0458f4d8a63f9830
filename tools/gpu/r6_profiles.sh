#!/bin/bash
# round-6 profile set (copied into profiles/r05_*) on the FINAL library: kernel traces of the default bench command (timed steps =
# hipGraph replays), of eagerly launched steps on four streams (two C calls per step) and on ONE stream (durations add up to the
# step), per-step launch sequences, PMC traffic passes, one SQ counter pass over a serialized step, the force-training step
# (BASELINE configs[3]) through the whole-model C calls, run-to-run reproducibility of the headline step
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
prof() { # name, env..., BARGS in the environment
  name=$1; shift
  rm -rf /tmp/p_$name
  env "$@" timeout 600 rocprofv3 --kernel-trace -d /tmp/p_$name -o r -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --streamed-steps 0 --eager-steps 0 --no-micro --other-configs 0 $BARGS > gpurun_out/prof_${name}_bench.json 2> gpurun_out/prof_${name}.err
  db=$(find /tmp/p_$name -name "*.db" | head -1)
  python tools/rocpd_stats.py $db > gpurun_out/prof_${name}_kernel_stats.txt
  python tools/rocpd_stats.py $db --grid > gpurun_out/prof_${name}_kernel_stats_by_grid.txt
  python tools/rocpd_timeline.py $db 2 > gpurun_out/prof_${name}_timeline.txt
  python tools/rocpd_sequence.py $db 2 > gpurun_out/prof_${name}_sequence.txt
  sed -n 2,4p gpurun_out/prof_${name}_timeline.txt
}
BARGS="" prof default A=1
BARGS="" prof eager ALIGNN_BENCH_EAGER=1
BARGS="" prof serialized ALIGNN_BENCH_EAGER=1 ALIGNN_AMD_SIDE_STREAM=0 ALIGNN_AMD_LANES=0 ALIGNN_AMD_FORK=0
BARGS="--model alignn_ff --batch 16 --atoms 200" prof cfg4_ff ALIGNN_BENCH_EAGER=1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o r -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --streamed-steps 0 --eager-steps 0 --no-micro --other-configs 0 > /dev/null 2> gpurun_out/prof_pmc_$c.err
  db=$(find /tmp/pmc_$c -name "*.db" | head -1)
  python tools/rocpd_pmc.py $db 0 > gpurun_out/prof_pmc_$c.txt
  head -6 gpurun_out/prof_pmc_$c.txt | cut -c1-200
done
rm -rf /tmp/sq5
ALIGNN_BENCH_EAGER=1 ALIGNN_AMD_SIDE_STREAM=0 ALIGNN_AMD_LANES=0 ALIGNN_AMD_FORK=0 timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d /tmp/sq5 -o r -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --streamed-steps 0 --eager-steps 0 --no-micro --other-configs 0 > /dev/null 2> gpurun_out/prof_sq.err
db=$(find /tmp/sq5 -name "*.db" | head -1)
python tools/rocpd_sq.py $db 100 > gpurun_out/prof_sq_step.txt; head -14 gpurun_out/prof_sq_step.txt | cut -c1-190
rm -rf /tmp/sq5f
ALIGNN_BENCH_EAGER=1 timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d /tmp/sq5f -o r -- python bench.py --model alignn_ff --batch 16 --atoms 200 --steps 2 --warmup 1 --no-cpu-baseline --streamed-steps 0 --eager-steps 0 --no-micro --other-configs 0 > /dev/null 2> gpurun_out/prof_sq_ff.err
db=$(find /tmp/sq5f -name "*.db" | head -1)
python tools/rocpd_sq.py $db 150 > gpurun_out/prof_sq_cfg4_ff.txt; head -8 gpurun_out/prof_sq_cfg4_ff.txt | cut -c1-190
{ timeout 300 python tools/bn_repro_check.py 64; timeout 300 python tools/bn_repro_check.py 128; } 2>&1 | grep "run\|C calls" > gpurun_out/prof_bn_repro.txt; cat gpurun_out/prof_bn_repro.txt | cut -c1-160
timeout 300 python tools/md_step.py 2>&1 | tail -3 > gpurun_out/prof_md_step.txt; cat gpurun_out/prof_md_step.txt
timeout 300 python tools/infer_time.py 2>&1 | tail -3 > gpurun_out/prof_infer.txt; cat gpurun_out/prof_infer.txt | cut -c1-200
