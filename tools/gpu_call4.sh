#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --streamed-steps 0 --eager-steps 5 > gpurun_out/c4_$name.json 2> gpurun_out/c4_$name.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/c4_$name.json"))
    print("$name: replay", d["ms_per_step"], "eager", d["eager_launches"]["ms_per_step"], "enq", d["eager_launches"]["host_enqueue_ms_per_step"], "in_step", d["roofline"]["in_step"]["ms_per_launch"])
except Exception as e:
    print("$name FAILED", e); print(open("gpurun_out/c4_$name.err").read()[-1500:])
PY
}
run lanes0 ALIGNN_AMD_LANES=0
run lanes1 ALIGNN_AMD_LANES=1
run lanes1_Tlow1 ALIGNN_AMD_LANES=1 ALIGNN_AMD_LANE_PRIORITY=0 ALIGNN_AMD_SIDE_PRIORITY=0
run lanes1_mainhigh ALIGNN_AMD_LANES=1 ALIGNN_BENCH_MAIN_PRIORITY=-1
run lanes1_mainhigh_Tlow ALIGNN_AMD_LANES=1 ALIGNN_BENCH_MAIN_PRIORITY=-1 ALIGNN_AMD_LANE_PRIORITY=0 ALIGNN_AMD_SIDE_PRIORITY=0
run lanes0_mainhigh ALIGNN_AMD_LANES=0 ALIGNN_BENCH_MAIN_PRIORITY=-1
timeout 600 python -m pytest tests/test_gpu_full_size.py -m gpu -q 2>&1 | tail -3
