"""Inference throughput of the headline model (eval(), no_grad, 64 crystals x 60 atoms): BatchNorm-folded gate pass
(alignn_egc_gate_infer) vs the training-capable kernels in eval mode."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alignn_amd import ALIGNN, ALIGNNConfig, GraphBatch, ops
from alignn_amd.synthetic import make_batch

dev = "cuda"
batch = GraphBatch.from_raw(make_batch(64, 60), device=dev)
torch.manual_seed(0)
model = ALIGNN(ALIGNNConfig(name="alignn")).to(dev).eval()

def t(n=10):
    with torch.no_grad():
        for _ in range(3): out = model(batch)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): out = model(batch)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, out
tf, of = t()
from alignn_amd import cmodel
cmodel.ENABLED = False
tp, op_ = t()
print(f"one C call {tf*1e3:.2f} ms ({64/tf:.0f} graphs/s) vs per-operator launches {tp*1e3:.2f} ms; bit-identical: {bool(torch.equal(of, op_))}")
ops.INFER_FUSED = False
ts, os_ = t()
print(f"inference, 64 x 60 atoms: folded {tf*1e3:.2f} ms ({64/tf:.0f} graphs/s), unfolded eval {ts*1e3:.2f} ms ({64/ts:.0f} graphs/s), "
      f"max rel diff {float((of-os_).abs().max()/os_.abs().max()):.2e}")
