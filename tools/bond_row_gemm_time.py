"""Bond-row (E = 50 712 rows at the benchmark batch) split-product projections: one thin generation of 397 128-row tiles on
256 CUs.  Times every epilogue variant a step launches at that height; run once per ALIGNN_AMD_X6_RM1_BELOW setting
(256 = default: 128-row tiles; 512: 793 64-row tiles)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alignn_amd import GraphBatch, ops
from alignn_amd.synthetic import make_batch
dev = "cuda"
b = GraphBatch.from_raw(make_batch(64, 60), device=dev)
g = b.g
E, N, H = g.n_edges, g.n_nodes, 256
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
y = torch.randn(E, H, device=dev); am = ops.absmax(y)
w = torch.randn(H, H, device=dev) / 16; wh = ops.split_f16x2(w); bias = torch.randn(H, device=dev)
wcat = torch.randn(4 * H, H, device=dev) / 16; wch = ops.split_f16x2(wcat); bcat = torch.randn(4 * H, device=dev)
wct = ops.split_f16x2(wcat.t().contiguous())
P = torch.randn(N, 4 * H, device=dev); GP = torch.randn(E, 4 * H, device=dev); amg = ops.absmax(GP)
res = torch.randn(E, H, device=dev)
print("RM1_BELOW =", os.environ.get("ALIGNN_AMD_X6_RM1_BELOW", "256"), " E =", E)
print("plain           E x 256 x 256   %.1f us" % t(lambda: ops.gemm_nt_f16x3(y, am, wh, bias)))
print("plain           E x 1024 x 256  %.1f us" % t(lambda: ops.gemm_nt_f16x3(y, am, wch, bcat)))
print("addend          E x 256 x 1024  %.1f us" % t(lambda: ops.gemm_nt_f16x3(GP, amg, wct, None, addend=res)))
print("gather + stats  E x 256 x 256   %.1f us" % t(lambda: ops.gemm_nt_f16x3_gather(y, am, wh, bias, P, g.src, g.dst, want_stats=True)))
