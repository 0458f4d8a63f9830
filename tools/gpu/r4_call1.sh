#!/bin/bash
# round 4, first GPU call: the whole-model C entry points - bit equality with the per-operator path, then bench lines
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_cmodel.py -x -q 2>&1 | tail -40 > gpurun_out/c1_cmodel.log
timeout 600 python -m pytest tests/test_gpu_round3.py -q -k "composite_entry or registry_hits" 2>&1 | tail -25 > gpurun_out/c1_round3.log
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/c1_bench.json 2> gpurun_out/c1_bench.err
ALIGNN_AMD_CMODEL=0 timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-micro > gpurun_out/c1_bench_ops.json 2> gpurun_out/c1_bench_ops.err
tail -3 gpurun_out/c1_cmodel.log; tail -3 gpurun_out/c1_round3.log
python - <<'PY'
import json
for f in ("c1_bench", "c1_bench_ops"):
    try:
        o = json.load(open(f"gpurun_out/{f}.json"))
        print(f, o["ms_per_step"], "eager", o["eager_launches"], "streamed", (o["streamed_batches"] or {}).get("ms_per_step"), "enq", o["host_enqueue_ms_per_step"], "peak", o["peak_hbm_GB"])
    except Exception as e:
        print(f, "failed", e)
PY
