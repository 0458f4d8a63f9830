#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf /tmp/p_mb
timeout 600 rocprofv3 --kernel-trace -d /tmp/p_mb -o r -- python tools/microbatch_probe.py ${PARTS:-2} 16 200 5 > gpurun_out/r6_microbatch.txt 2>&1
db=$(find /tmp/p_mb -name "*.db" | head -1)
python tools/rocpd_timeline.py $db 1 > gpurun_out/r6_microbatch_timeline.txt
grep "parts=" gpurun_out/r6_microbatch.txt
sed -n 2,12p gpurun_out/r6_microbatch_timeline.txt | cut -c1-160
grep -A14 "big kernels" gpurun_out/r6_microbatch_timeline.txt | cut -c1-150
