"""What staging a FRESH batch costs by route (64 crystals x 60 atoms, inputs already on the device):
  from_coo with the caller's explicit line graph (what ALIGNN.forward does with the reference's (g, lg) DGL pair), from_coo
  deriving L(g) itself, and the packed-buffer loader.  Host time (enqueue) and wall time per batch."""
import sys
import time

import torch

sys.path.insert(0, ".")
from alignn_amd import GraphBatch, loader  # noqa: E402
from alignn_amd.synthetic import make_batch  # noqa: E402

dev = "cuda"
raw = make_batch(64, 60)
t = torch.from_numpy
u, v, lu, lv = (t(a).to(dev) for a in (raw.u, raw.v, raw.lg_u, raw.lg_v))
af, r, h = t(raw.atom_features).to(dev), t(raw.r).to(dev), t(raw.h).to(dev)
bnn = t(raw.batch_num_nodes)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    return host * 1e3, (time.perf_counter() - t0) / n * 1e3


print("explicit (g, lg):   host %.2f ms, wall %.2f ms" % timeit(lambda: GraphBatch.from_coo(u, v, raw.num_nodes, bnn, lu, lv, af, r, h, device=dev)))
print("L(g) derived:       host %.2f ms, wall %.2f ms" % timeit(lambda: GraphBatch.from_coo(u, v, raw.num_nodes, bnn, atom_features=af, r=r, device=dev, build_line_graph=True)))
p = loader.pack_raw(raw)
print("packed loader:      host %.2f ms, wall %.2f ms (incl. the 2.4 MB H2D copy)" % timeit(lambda: loader.stage(p, dev)))
