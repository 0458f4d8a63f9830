#!/bin/bash
# A/B of the fused input-gradient + weight-gradient pass inside the force-training step (BASELINE configs[3]), same box, alternating
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
for i in 1 2; do
  for v in 1 0; do
    ALIGNN_AMD_DW_FUSED=$v timeout 500 python bench.py --model alignn_ff --batch 16 --atoms 200 --no-cpu-baseline --no-micro --other-configs 0 --streamed-steps 0 --steps 10 > gpurun_out/r6_ab_dwff_${v}_$i.json 2> gpurun_out/r6_ab_dwff_${v}_$i.err
    python - <<PY
import json
d=json.load(open("gpurun_out/r6_ab_dwff_${v}_$i.json"))
print("DW_FUSED=$v run $i:", d["ms_per_step"], "replay", (d.get("replayed_steps") or {}).get("ms_per_step"), "eager", (d.get("eager_launches") or {}).get("ms_per_step"))
r=d.get("roofline",{}).get("in_step") or {}
print("   ", {k:(v["launches"], v["ms_per_launch"]) for k,v in r.items() if isinstance(v,dict) and "launches" in v})
PY
  done
done
