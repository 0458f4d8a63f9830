"""The pipelined form of the gather projection (ALIGNN_AMD_X6PP=1) against the shipped one: run once per setting - the second
run compares outputs and column-sum slabs bit for bit - and time both."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alignn_amd import GraphBatch, ops
from alignn_amd.synthetic import make_batch
dev = "cuda"
mode = os.environ.get("ALIGNN_AMD_X6PP", "0")
b = GraphBatch.from_raw(make_batch(int(os.environ.get("B", "64")), 60), device=dev)
lg = b.lg
T, E, H = lg.n_edges, lg.n_nodes, 256
g = torch.Generator(device=dev).manual_seed(0)
y = torch.randn(T, H, device=dev, generator=g); P = torch.randn(E, 4 * H, device=dev, generator=g)
w = torch.randn(H, H, device=dev, generator=g) / 16; bias = torch.randn(H, device=dev, generator=g)
wh, am = ops.split_f16x2(w), ops.absmax(y)
bd2 = ops.segment_ordered_bd(P, lg, H)
def t(fn, k=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(k): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / k * 1e3
out = torch.empty(T, H, device=dev)
f_stats = lambda: ops.gemm_nt_f16x3_gather(y, am, wh, bias, P, lg.src, lg.dst, out=out, want_stats=True, bd2=bd2, rank=lg.seg_rank)
f_plain = lambda: ops.gemm_nt_f16x3_gather(y, am, wh, bias, P, lg.src, lg.dst, out=out, bd2=bd2, rank=lg.seg_rank)
print(f"X6PP={mode}  T={T}: gather + statistics {t(f_stats):.1f} us, gather {t(f_plain):.1f} us")
o, part, tiles = f_stats(); torch.cuda.synchronize()
torch.save({"out": o.cpu(), "part": part[:tiles].cpu()}, f"/tmp/x6pp_{mode}.pt")
o2 = f_plain().clone(); torch.cuda.synchronize()
print("  stats variant == plain variant:", torch.equal(o.cpu(), o2.cpu()))
if os.path.exists("/tmp/x6pp_0.pt") and os.path.exists("/tmp/x6pp_1.pt"):
    a, c = torch.load("/tmp/x6pp_0.pt"), torch.load("/tmp/x6pp_1.pt")
    print("  pipelined == shipped: out", torch.equal(a["out"], c["out"]), " column sums", torch.equal(a["part"], c["part"]),
          " max |diff|", float((a["out"] - c["out"]).abs().max()))
    ref = (y.double() @ w.double().t() + bias.double() + P[lg.src.long(), :H].double() + P[lg.dst.long(), H:2 * H].double())
    print("  vs float64: shipped %.2e, pipelined %.2e (relative to max)" % (float((a["out"].to(dev).double() - ref).abs().max() / ref.abs().max()), float((c["out"].to(dev).double() - ref).abs().max() / ref.abs().max())))
