"""How much of the gather epilogue's cost is WHERE the gathered rows come from?  The T-row edge-gate projection + u_add_v
(alignn_gemm_nt_f16x3_gather) timed with the real line-graph indices of the benchmark batch, with all rows gathering row 0
(every gather an L1 hit), and with a random permutation (every gather a miss), next to the plain projection."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alignn_amd import GraphBatch, ops
from alignn_amd.synthetic import make_batch
dev = "cuda"
b = GraphBatch.from_raw(make_batch(64, 60), device=dev)
lg = b.lg
T, E, H = lg.n_edges, lg.n_nodes, 256
y = torch.randn(T, H, device=dev); P = torch.randn(E, 4 * H, device=dev)
w = torch.randn(H, H, device=dev) / 16; bias = torch.randn(H, device=dev)
wh, am = ops.split_f16x2(w), ops.absmax(y)
out = torch.empty(T, H, device=dev)
def t(fn, k=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(k): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / k * 1e3
z = torch.zeros(T, dtype=torch.int32, device=dev)
rp = torch.randint(0, E, (T,), dtype=torch.int32, device=dev)
print("T", T, "E", E)
print("plain projection            %.1f us" % t(lambda: ops.gemm_nt_f16x3(y, am, wh, bias, out=out)))
print("gather, line-graph indices  %.1f us" % t(lambda: ops.gemm_nt_f16x3_gather(y, am, wh, bias, P, lg.src, lg.dst, out=out)))
print("gather + statistics         %.1f us" % t(lambda: ops.gemm_nt_f16x3_gather(y, am, wh, bias, P, lg.src, lg.dst, out=out, want_stats=True)))
print("gather, every row -> row 0  %.1f us" % t(lambda: ops.gemm_nt_f16x3_gather(y, am, wh, bias, P, z, z, out=out)))
print("gather, src real, dst -> 0  %.1f us" % t(lambda: ops.gemm_nt_f16x3_gather(y, am, wh, bias, P, lg.src, z, out=out)))
print("gather, random rows         %.1f us" % t(lambda: ops.gemm_nt_f16x3_gather(y, am, wh, bias, P, rp, rp, out=out)))
bd2 = ops.segment_ordered_bd(P, lg, H)
print("segment-ordered copy of Bd  %.1f us" % t(lambda: ops.segment_ordered_bd(P, lg, H)))
print("gather, Bd from the table   %.1f us" % t(lambda: ops.gemm_nt_f16x3_gather(y, am, wh, bias, P, lg.src, lg.dst, out=out, bd2=bd2, rank=lg.seg_rank)))
ref = ops.gemm_nt_f16x3_gather(y, am, wh, bias, P, lg.src, lg.dst)
print("same bits:", torch.equal(ref, ops.gemm_nt_f16x3_gather(y, am, wh, bias, P, lg.src, lg.dst, bd2=bd2, rank=lg.seg_rank)))
