#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_graph_builder_golden.py tests/test_gpu_full_size.py -m gpu -q -rf > gpurun_out/c3_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c3_pytest.log
tail -12 gpurun_out/c3_pytest.log
for lanes in 1 0; do
  rm -rf /tmp/prof_l$lanes
  ALIGNN_AMD_LANES=$lanes timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_l$lanes -o r -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --streamed-steps 0 --eager-steps 0 > gpurun_out/c3_prof_lanes$lanes.json 2> gpurun_out/c3_prof_lanes$lanes.err
  db=$(find /tmp/prof_l$lanes -name "*.db" | head -1)
  python tools/rocpd_timeline.py $db 2 > gpurun_out/c3_timeline_lanes$lanes.txt 2>&1
  head -40 gpurun_out/c3_timeline_lanes$lanes.txt
done
