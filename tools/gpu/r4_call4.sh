#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
ALIGNN_BENCH_EAGER_AFTER=1 timeout 600 python bench.py --no-cpu-baseline --no-micro > gpurun_out/c4_bench.json 2> gpurun_out/c4_bench.err
python - <<'PY'
import json
o = json.load(open("gpurun_out/c4_bench.json"))
print("replay", o["ms_per_step"], "eager", o["eager_launches"], "streamed", o["streamed_batches"] and o["streamed_batches"]["ms_per_step"], "enq", o["host_enqueue_ms_per_step"], "peak", o["peak_hbm_GB"])
PY
ALIGNN_AMD_DEBUG_TIME=1 ALIGNN_BENCH_EAGER=1 timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-micro --streamed-steps 0 --eager-steps 0 2>&1 | grep "alignn_model" | tail -4
