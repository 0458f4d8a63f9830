"""Count which split-product scheme every projection of one training step takes (f16x3 needs a tracked max|A|;
a projection that silently lost its bound falls back to bf16x6 - correct, but 1.5x slower at T rows)."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from alignn_amd import ALIGNN, ALIGNNConfig, ops
from alignn_amd.graph import GraphBatch
from alignn_amd.synthetic import make_batch

dev = torch.device("cuda", 0)
raw = make_batch(64, 60, seed0=1234)
batch = GraphBatch.from_raw(raw, device=dev)
model = ALIGNN(ALIGNNConfig(name="alignn")).to(dev).train()
target = torch.randn(64, device=dev)
calls = collections.Counter()
for name in ("gemm_nt_f16x3", "gemm_nt_x6", "gemm_nt", "gemm_nn"):
    f = getattr(ops, name)

    def wrap(*a, _f=f, _n=name, **k):
        A = a[0]
        calls[(_n, A.shape[0], A.shape[1])] += 1
        return _f(*a, **k)

    setattr(ops, name, wrap)
tn = ops.gemm_tn


def tn_wrap(g, a, g_amax=None, a_amax=None):
    calls[("gemm_tn " + ("f16x3" if (g_amax is not None and a_amax is not None) else "no-amax"), g.shape[0], g.shape[1], a.shape[1])] += 1
    return tn(g, a, g_amax, a_amax)


ops.gemm_tn = tn_wrap
for _ in range(2):
    calls.clear()
    torch.nn.functional.l1_loss(model(batch), target).backward()
torch.cuda.synchronize()
for k, v in sorted(calls.items(), key=lambda kv: (kv[0][0], -kv[0][1])):
    print(v, k)
