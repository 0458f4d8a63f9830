#!/bin/bash
# end-of-round verification: build check, smoke(), the whole GPU suite, the default bench command
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -3
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -12 > gpurun_out/verify_tests.log
tail -4 gpurun_out/verify_tests.log
timeout 900 python bench.py > gpurun_out/verify_bench.json 2> gpurun_out/verify_bench.err
python - <<'PY'
import json
o = json.load(open("gpurun_out/verify_bench.json"))
print("replay", o["ms_per_step"], "eager", o["eager_launches"], "streamed", o["streamed_batches"] and o["streamed_batches"]["ms_per_step"], "enq", o["host_enqueue_ms_per_step"], "peak", o["peak_hbm_GB"])
print("roofline frac", o["roofline"]["frac"], "cpu", o["cpu_baseline"] and o["cpu_baseline"]["value"], "calib", o["host_calibration"])
for k, v in (o.get("other_configs") or {}).items():
    print(k, v.get("ms_per_step"), v.get("value"), (v.get("roofline") or {}).get("frac"), (v.get("cpu_baseline") or {}).get("value"), v.get("error"))
PY
