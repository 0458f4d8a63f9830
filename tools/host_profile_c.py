"""Where the HOST time of an eagerly launched training step goes with the whole-model C entry points (alignn_amd/cmodel.py):
enqueue time of forward / loss / backward / optimizer per step, then a cProfile of 20 steps.  usage: python tools/host_profile_c.py [B]"""
import cProfile
import pstats
import sys
import time

import torch

sys.path.insert(0, ".")
from alignn_amd import ALIGNN, ALIGNNConfig, GraphBatch  # noqa: E402
from alignn_amd.optim import FlatAdamW, group_decay  # noqa: E402
from alignn_amd.synthetic import make_batch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
batch = GraphBatch.from_raw(make_batch(B, 60), device=dev)
target = torch.randn(B, device=dev)
torch.manual_seed(0)
model = ALIGNN(ALIGNNConfig(name="alignn")).to(dev).train()
opt = FlatAdamW(group_decay(model), lr=1e-3, weight_decay=1e-2, module=model)
params = list(model.parameters())


def step(t):
    t0 = time.perf_counter()
    for p in params:
        p.grad = None
    t1 = time.perf_counter()
    pred = model(batch)
    t2 = time.perf_counter()
    loss = torch.nn.functional.l1_loss(pred, target)
    t3 = time.perf_counter()
    loss.backward()
    t4 = time.perf_counter()
    opt.step()
    t5 = time.perf_counter()
    if t is not None:
        for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            t[i] += d


for _ in range(5):
    step(None)
torch.cuda.synchronize()
acc = [0.0] * 5
n = 30
from alignn_amd import cmodel  # noqa: E402
cmodel.TIMING = {}
t0 = time.perf_counter()
for _ in range(n):
    step(acc)
enq = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f"B={B}: wall {wall / n * 1e3:.2f} ms/step, host enqueue {enq / n * 1e3:.2f} ms/step")
print("  inside: C forward call %.3f ms, C backward call %.3f ms, Function.backward in all %.3f ms" % tuple(
    cmodel.TIMING.get(k, 0.0) / n * 1e3 for k in ("cfwd", "cbwd", "bwd_py")))
cmodel.TIMING = None
print("  zero_grad %.3f  forward %.3f  loss %.3f  backward %.3f  optimizer %.3f  (ms)" % tuple(a / n * 1e3 for a in acc))
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step(None)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
