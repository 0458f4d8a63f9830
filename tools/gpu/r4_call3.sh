#!/bin/bash
# round 4, call 3: staging kernel tests, cmodel tests again, host profile, default bench line (with other configs)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_stage.py tests/test_gpu_cmodel.py -q 2>&1 | tail -30 > gpurun_out/c3_tests.log
tail -5 gpurun_out/c3_tests.log
timeout 300 python tools/host_profile_c.py 64 > gpurun_out/c3_host_profile.txt 2>&1
head -6 gpurun_out/c3_host_profile.txt
timeout 900 python bench.py > gpurun_out/c3_bench.json 2> gpurun_out/c3_bench.err
python - <<'PY'
import json
o = json.load(open("gpurun_out/c3_bench.json"))
print("replay", o["ms_per_step"], "eager", o["eager_launches"], "streamed", o["streamed_batches"] and o["streamed_batches"]["ms_per_step"], "enq", o["host_enqueue_ms_per_step"], "peak", o["peak_hbm_GB"])
print("calib", o["host_calibration"])
print("roofline frac", o["roofline"]["frac"], "cpu", o["cpu_baseline"] and o["cpu_baseline"]["value"])
for k, v in (o.get("other_configs") or {}).items():
    print(k, v.get("ms_per_step"), v.get("value"), (v.get("roofline") or {}).get("frac"), (v.get("cpu_baseline") or {}).get("value"), v.get("error"), v.get("wall_s_incl_start_up"))
PY
tail -5 gpurun_out/c3_bench.err
