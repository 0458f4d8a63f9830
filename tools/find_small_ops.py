"""Which Python lines launch the small torch kernels (fills, copies, cats, elementwise) of one training step?
torch.profiler with stacks over one eagerly launched step of the bench workload.  usage: python tools/find_small_ops.py"""
import collections, os, sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alignn_amd import ALIGNN, ALIGNNConfig, GraphBatch  # noqa: E402
from alignn_amd.ddp import FlatGradSync  # noqa: E402
from alignn_amd.synthetic import make_batch  # noqa: E402

dev = torch.device("cuda", 0)
raw = make_batch(64, 60, seed0=1234)
batch = GraphBatch.from_raw(raw, device=dev)
torch.manual_seed(0)
model = ALIGNN(ALIGNNConfig(name="alignn")).to(dev).train()
target = torch.randn(64).to(dev)
sync = FlatGradSync(model.parameters())
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, fused=True)


def step():
    sync.zero_grad()
    loss = torch.nn.functional.l1_loss(model(batch), target)
    loss.backward()
    sync.sync()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
by = collections.Counter()
dur = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith("aten::") and ev.name.split("::")[1] in (
            "zero_", "fill_", "zeros", "zeros_like", "copy_", "clone", "contiguous", "cat", "empty_like", "add_", "add", "mul", "div",
            "sum", "mean", "abs", "sub", "neg", "to", "_to_copy", "ones_like", "full", "index_select", "stack", "select_backward", "sgn",
            "l1_loss", "expand", "mul_"):
        frames = [f for f in (ev.stack or []) if root in f and "find_small_ops" not in f]
        where = frames[0].split(root + "/")[-1] if frames else ("<autograd/optimizer>" if not ev.stack else ev.stack[0][-60:])
        cuda_us = sum(k.duration for k in ev.kernels) if ev.kernels else 0
        if ev.kernels:
            by[(ev.name, where)] += len(ev.kernels)
            dur[(ev.name, where)] += cuda_us
print(f"{'op':22s} {'kernels':>7s} {'GPU us':>8s}  python frame")
for k, n in sorted(by.items(), key=lambda kv: -dur[kv[0]])[:60]:
    print(f"{k[0]:22s} {n:7d} {dur[k]:8.1f}  {k[1]}")
print("total kernels from these ops:", sum(by.values()), " GPU us:", round(sum(dur.values()), 1))
