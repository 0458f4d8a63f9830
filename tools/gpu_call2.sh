#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_graph_builder_golden.py tests/test_gpu_full_size.py -m gpu -q -rf -x > gpurun_out/c2_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c2_pytest.log
tail -30 gpurun_out/c2_pytest.log
for lanes in 1 0; do
  ALIGNN_AMD_LANES=$lanes timeout 300 python bench.py --no-cpu-baseline --streamed-steps 4 > gpurun_out/c2_bench_lanes$lanes.json 2> gpurun_out/c2_bench_lanes$lanes.err; echo "bench lanes=$lanes rc=$?"
  tail -4 gpurun_out/c2_bench_lanes$lanes.err
  python - <<PY
import json
d=json.load(open("gpurun_out/c2_bench_lanes$lanes.json"))
print("lanes=$lanes", d["ms_per_step"], "eager", d["eager_launches"], "streamed", d["streamed_batches"]["ms_per_step"], "in_step", d["roofline"]["in_step"]["ms_per_launch"])
PY
done
