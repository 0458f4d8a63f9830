"""Which host-side torch operations put copies (hipMemcpy D2D = __amd_rocclr_copyBuffer, elementwise copy kernels, cat) into
an eagerly launched training step?  torch.profiler over one step of bench.py's loop, grouped by op / shapes / Python frame."""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alignn_amd import ALIGNN, ALIGNNConfig, GraphBatch
from alignn_amd.optim import FlatAdamW, group_decay
from alignn_amd.synthetic import make_batch
from torch.profiler import profile, ProfilerActivity
dev = "cuda"
B = int(os.environ.get("B", "64"))
batch = GraphBatch.from_raw(make_batch(B, 60), device=dev)
torch.manual_seed(0)
model = ALIGNN(ALIGNNConfig(name="alignn")).to(dev).train()
opt = FlatAdamW(group_decay(model), lr=1e-3, weight_decay=1e-2, module=model, average_gradients=True)
target = torch.randn(B, device=dev)
def step():
    opt.zero_grad()
    torch.nn.functional.l1_loss(model(batch), target).backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
groups = collections.Counter()
names = collections.Counter()
for e in prof.events():
    names[e.name] += 1
    if e.name in ("aten::copy_", "aten::clone", "aten::cat", "aten::contiguous", "aten::add_", "aten::zero_", "aten::fill_", "aten::mul", "aten::to", "aten::_to_copy", "aten::index", "aten::div_"):
        frames = [f for f in (e.stack or []) if "alignn_amd" in f or "bench" in f or "tools/" in f][:2]
        groups[(e.name, str(e.input_shapes)[:80], " <- ".join(frames)[:200])] += 1
for (n, shp, st), c in sorted(groups.items(), key=lambda kv: -kv[1])[:60]:
    print(f"{c:4d} {n:16s} {shp:80s} {st}")
print("--- device-side names")
for n, c in sorted(names.items(), key=lambda kv: -kv[1])[:70]:
    print(f"{c:5d} {n[:150]}")
