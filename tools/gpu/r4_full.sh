#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15
timeout 600 python bench.py --no-cpu-baseline --other-configs 0 > gpurun_out/bench_angle.json 2> gpurun_out/bench_angle.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_angle.json'))
print({k:d[k] for k in ('value','ms_per_step') if k in d}, d.get('eager_launches'), d.get('replayed_steps'))
PY
