#!/bin/bash
export TMPDIR=/tmp ALIGNN_AMD_DEBUG=1
mkdir -p gpurun_out
timeout 300 python - > gpurun_out/dbg.log 2>&1 <<'PY'
import torch
from alignn_amd import ALIGNN, ALIGNNConfig, GraphBatch, cmodel
from alignn_amd.synthetic import make_batch
raw = make_batch(16, 60, seed0=77)
b = GraphBatch.from_raw(raw, device="cuda")
t = torch.randn(16).cuda()
torch.manual_seed(0)
m = ALIGNN(ALIGNNConfig(name="alignn")).cuda().train()
try:
    torch.nn.functional.l1_loss(m(b), t).backward()
    torch.cuda.synchronize()
    print("ok")
except Exception as e:
    print("ERR", e)
PY
tail -20 gpurun_out/dbg.log
timeout 600 python -m pytest tests/test_gpu_cmodel.py -q 2>&1 | tail -40 > gpurun_out/c1_cmodel.log
tail -15 gpurun_out/c1_cmodel.log
