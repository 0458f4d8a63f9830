"""Run the T-row f16x3 projections as a training step launches them (plain, gather+stats, bnred+addend) and the weight
gradient a few times - a minimal target for rocprofv3 --pmc passes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alignn_amd import ops
from alignn_amd import GraphBatch
from alignn_amd.synthetic import make_batch
lg = GraphBatch.from_raw(make_batch(64, 60), device="cuda").lg  # the benchmark batch's line graph: real gather indices
T, E, H = lg.n_edges, lg.n_nodes, 256
g = torch.Generator().manual_seed(0)
a = torch.randn(T, H, generator=g).cuda(); w = (torch.randn(H, H, generator=g) / 16).cuda(); b = torch.randn(H, generator=g).cuda()
res = torch.randn(T, H, generator=g).cuda(); xn = torch.randn(T, H, generator=g).cuda()
stat = torch.stack([xn.mean(0), torch.rsqrt(xn.var(0, unbiased=False) + 1e-5), torch.ones(H).cuda(), torch.zeros(H).cuda()]).contiguous()
P = torch.randn(E, 4 * H, generator=g).cuda()
out = torch.empty(T, H, device="cuda")
wsh, wst, am, am2 = ops.split_f16x2(w), ops.split_f16x2(w, transpose=True), ops.absmax(a), ops.absmax(res)
for _ in range(4):
    ops.gemm_nt_f16x3(a, am, wsh, b, out=out)  # plain (persistent)
    bd2 = ops.segment_ordered_bd(P, lg, H)
    ops.gemm_nt_f16x3_gather(a, am, wsh, b, P, lg.src, lg.dst, out=out, want_stats=True, bd2=bd2, rank=lg.seg_rank)  # as a step launches it
    ops.gemm_nt_f16x3_bnred(a, am, wst, xn, stat, None, None, out=out)  # BatchNorm-backward sums (persistent)
    ops.gemm_nt_f16x3_bnred(a, am, wst, xn, stat, None, res, out=out)  # ... + residual addend (one-tile kernel)
    ops.gemm_tn(a, res, g_amax=am, a_amax=am2)
torch.cuda.synchronize()
