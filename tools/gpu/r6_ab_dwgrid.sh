#!/bin/bash
# A/B of the fused pass's grid (compute units it leaves to the other streams) inside the headline step and the force-training step
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
for i in 1 2; do
  for v in ${GRIDS:-256 248 240 224}; do
    ALIGNN_AMD_DW_GRID=$v timeout 400 python bench.py --no-cpu-baseline --no-micro --other-configs 0 --streamed-steps 0 --steps 20 $BARGS > gpurun_out/r6_ab_dwgrid_${v}_$i.json 2> gpurun_out/r6_ab_dwgrid_${v}_$i.err
    python - <<PY
import json
d=json.load(open("gpurun_out/r6_ab_dwgrid_${v}_$i.json"))
print("DW_GRID=$v run $i:", d["ms_per_step"], "replay", (d.get("replayed_steps") or {}).get("ms_per_step"), "eager", (d.get("eager_launches") or {}).get("ms_per_step"), "roofline.frac", d["roofline"].get("frac"), {k: v["ms_per_launch"] for k, v in d["roofline"]["in_step"].items() if isinstance(v, dict) and "launches" in v})
PY
  done
done
