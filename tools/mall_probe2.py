"""Second look at the last-level cache with the library's own streaming kernel (alignn_bn_silu_fwd: float4 per lane, the
nontemporal form above 128 MB, the cached form below): consumer Y = silu(BN(X)) timed (a) right after a producer wrote X
and (b) after a 1.4 GB pass over unrelated memory evicted it, for several sizes of X."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alignn_amd import _lib
from alignn_amd._lib import check, ptr
lib = _lib.load()
dev = "cuda"
F = 256
stat = torch.stack([torch.zeros(F), torch.ones(F), torch.ones(F), torch.zeros(F)]).to(dev).contiguous()
st = torch.cuda.current_stream().cuda_stream
junk = torch.randn(350_000_000, device=dev)  # 1.4 GB
def bn(x, y):
    check(lib.alignn_bn_silu_fwd(ptr(x), F, 0, 0, ptr(stat), ptr(y), F, x.shape[0], F, 0, st), "bn_silu_fwd")
def timed(fn, pre, n=10):
    tot = 0.0
    for _ in range(n):
        pre()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        tot += s.elapsed_time(e)
    return tot / n * 1e3
for mb in (32, 64, 100, 128, 173, 256):
    rows = mb * (1 << 20) // (F * 4)
    a = torch.randn(rows, F, device=dev); x = torch.empty_like(a); y = torch.empty_like(a)
    warm = timed(lambda: bn(x, y), lambda: bn(a, x))
    cold = timed(lambda: bn(x, y), lambda: (bn(a, x), junk.add_(1.0)))
    print(f"{mb:4d} MB: consumer right after its producer {warm:7.1f} us ({2 * mb / 1024 / warm * 1e6 / 1e3:5.2f} TB/s), after eviction {cold:7.1f} us ({2 * mb / 1024 / cold * 1e6 / 1e3:5.2f} TB/s)")
    del a, x, y
