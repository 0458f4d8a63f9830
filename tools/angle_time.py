"""The bond-angle embedding at the headline size (T = 676 200 rows): csrc/angle.hip (recomputing passes) forward + backward,
timed with events; errors against float64 printed once.  `rocprofv3 --kernel-trace --stats -- python tools/angle_time.py`
gives the per-pass durations."""
import sys

import torch

sys.path.insert(0, ".")
from alignn_amd.alignn import MLPLayer, RBFExpansion  # noqa: E402
from alignn_amd.angle import AngleEmbedding  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 676_200
dev = "cuda"
torch.manual_seed(0)
rbf = RBFExpansion(vmin=-1.0, vmax=1.0, bins=40).to(dev)
l1, l2 = MLPLayer(40, 64).to(dev).train(), MLPLayer(64, 256).to(dev).train()
h = torch.rand(T, device=dev) * 2 - 1
gz = torch.randn(T, 256, device=dev)
emb = AngleEmbedding(rbf.centers, rbf.gamma, (l1, l2))


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


print("forward  %.3f ms" % timed(lambda: emb.forward(h)))
emb.forward(h)
print("backward %.3f ms" % timed(lambda: emb.backward(gz)))
