"""Per-kernel times of the fused RBF + first-embedding-layer passes (csrc/rbf_mlp.hip) at T rows, next to the unfused chain
(rbf_fwd -> fp32 GEMM -> statistics -> normalise; backward: apply_sum, weight-gradient GEMM)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alignn_amd import _lib, ops
from alignn_amd._lib import check, ptr, stream
lib = _lib.load()
dev = "cuda"
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 676200
bins, F = (int(sys.argv[2]) if len(sys.argv) > 2 else 40), 64
d = torch.rand(rows, device=dev) * 2 - 1
centers = torch.linspace(-1, 1, bins, device=dev)
W = torch.randn(F, bins, device=dev) / 6; Wt = W.t().contiguous(); b = torch.randn(F, device=dev)
gamma, beta = torch.ones(F, device=dev), torch.zeros(F, device=dev)
rm, rv = torch.zeros(F, device=dev), torch.ones(F, device=dev)
slabs = lib.alignn_rbf_mlp_slabs(rows)
partial = torch.empty(slabs * (3 * F + 1), device=dev); stat = torch.empty(4, F, device=dev)
y = torch.empty(rows, F, device=dev); gy = torch.randn(rows, F, device=dev); gpre = torch.empty(rows, F, device=dev)
part2 = torch.empty(slabs, 2, F, device=dev); red = torch.zeros(2, F, device=dev); gbp = torch.empty(slabs, F, device=dev)
wpart = torch.empty(slabs, F * bins, device=dev)
def t(fn, k=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(k): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / k * 1e3
g = 19.5
print(f"rows {rows} bins {bins} F {F} slabs {slabs}")
print("stats      %.1f us" % t(lambda: check(lib.alignn_rbf_mlp_stats(ptr(d), ptr(centers), g, ptr(Wt), ptr(b), rows, bins, F, ptr(partial), stream()), "s")))
check(lib.alignn_bn_finalize_welford(ptr(partial), slabs, rows, F, ptr(gamma), ptr(beta), 1e-5, 0.1, ptr(rm), ptr(rv), ptr(stat), stream()), "f")
print("fwd        %.1f us" % t(lambda: check(lib.alignn_rbf_mlp_fwd(ptr(d), ptr(centers), g, ptr(Wt), ptr(b), rows, bins, F, ptr(stat), ptr(y), None, stream()), "f")))
print("bwd_reduce %.1f us" % t(lambda: check(lib.alignn_rbf_mlp_bwd_reduce(ptr(d), ptr(centers), g, ptr(Wt), ptr(b), rows, bins, F, ptr(stat), ptr(gy), ptr(part2), stream()), "r")))
print("bwd_apply  %.1f us" % t(lambda: check(lib.alignn_rbf_mlp_bwd_apply(ptr(d), ptr(centers), g, ptr(Wt), ptr(b), rows, bins, F, ptr(stat), ptr(gy), ptr(red), 0, ptr(gpre), ptr(gbp), None, stream()), "a")))
print("wgrad      %.1f us" % t(lambda: check(lib.alignn_rbf_mlp_wgrad(ptr(d), ptr(centers), g, ptr(gpre), rows, bins, F, ptr(wpart), stream()), "w")))
# the unfused chain
r = torch.empty(rows, bins, device=dev)
print("unfused: rbf_fwd %.1f us" % t(lambda: check(lib.alignn_rbf_fwd(ptr(d), ptr(centers), g, ptr(r), rows, bins, stream()), "rbf")))
print("unfused: gemm_nt %.1f us" % t(lambda: ops.gemm_nt(r, W, b, out=y)))
print("unfused: gemm_tn %.1f us" % t(lambda: ops.gemm_tn(gpre, r)))
