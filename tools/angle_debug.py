import sys
import torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from test_gpu_angle import _modules
from alignn_amd.angle import AngleEmbedding
dev = "cuda"
rbf, l1, l2 = _modules(0, 40)
g = torch.Generator().manual_seed(0)
h = torch.rand(1000, generator=g) * 2 - 1
rbf, l1, l2 = rbf.to(dev), l1.to(dev).train(), l2.to(dev).train()
emb = AngleEmbedding(rbf.centers, rbf.gamma, (l1, l2))
z, amax = emb.forward(h.to(dev))
torch.cuda.synchronize()
print("scal", emb.scal[:12].tolist())
print("shift nan", bool(emb.scal[16:80].isnan().any()))
print("stat1 nan", bool(emb.stat1.isnan().any()), emb.stat1[:4].tolist(), emb.stat1[64:68].tolist())
print("stat2 nan", bool(emb.stat2.isnan().any()), emb.stat2[:4].tolist(), emb.stat2[256:260].tolist())
print("z nan frac", float(z.isnan().float().mean()), float(amax))
print("rm", l1.layer[1].running_mean[:4].tolist(), l2.layer[1].running_var[:4].tolist())
gz = (torch.randn(1000, 256, generator=g) * (torch.rand(256, generator=g) + 0.1) + 0.05).to(dev)
print("z nan before bwd", float(z.isnan().float().mean()))
grads = emb.backward(gz)
torch.cuda.synchronize()
print("z nan after bwd", float(z.isnan().float().mean()))
for gg in grads:
    print([bool(t.isnan().any()) for t in gg], [float(t.abs().max()) for t in gg])
print("scal", emb.scal[:12].tolist())
