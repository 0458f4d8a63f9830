#!/bin/bash
# A/B of one environment switch inside the bench step, one box, alternating: VAR=name VALS="a b c" [BARGS=...] [ROUNDS=2]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
for i in $(seq 1 ${ROUNDS:-2}); do
  for v in $VALS; do
    env $VAR=$v timeout 400 python bench.py --no-cpu-baseline --no-micro --other-configs 0 --streamed-steps 0 --steps 20 $BARGS > gpurun_out/r6_ab_${VAR}_${v}_$i.json 2> gpurun_out/r6_ab_${VAR}_${v}_$i.err
    python - <<PY
import json
d=json.load(open("gpurun_out/r6_ab_${VAR}_${v}_$i.json"))
print("$VAR=$v run $i:", d["ms_per_step"], "replay", (d.get("replayed_steps") or {}).get("ms_per_step"), "eager", (d.get("eager_launches") or {}).get("ms_per_step"), "frac", d["roofline"].get("frac"))
PY
  done
done
