"""Host-side cost of the pieces an eagerly launched step is made of (us each): a C-ABI kernel launch through ctypes, the
same launch behind hipGraph replay, torch.empty, a torch elementwise op, an autograd.Function apply."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alignn_amd import _lib, ops
lib = _lib.load()
dev = "cuda"
a = torch.randn(64, 256, device=dev); out = torch.empty(256, device=dev)
def t(fn, k=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    dt = (time.perf_counter() - t0) / k * 1e6
    torch.cuda.synchronize(); return dt
st = _lib.stream()
print("ctypes launch (slab_sum, tiny):      %.2f us" % t(lambda: lib.alignn_slab_sum(a.data_ptr(), 64, 256, out.data_ptr(), st)))
print("  + ptr()/stream() lookups:          %.2f us" % t(lambda: lib.alignn_slab_sum(_lib.ptr(a), 64, 256, _lib.ptr(out), _lib.stream())))
print("ctypes call, no launch (slabs query): %.2f us" % t(lambda: lib.alignn_col_stats_slabs(1000)))
print("torch.empty(1024, 256):              %.2f us" % t(lambda: torch.empty(1024, 256, device=dev)))
print("torch add (tiny):                    %.2f us" % t(lambda: a + 1.0))
print("torch.cuda.Event record:             %.2f us" % t(lambda: torch.cuda.Event().record()))
class F(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x): return x
    @staticmethod
    def backward(ctx, g): return g
x = torch.randn(4, device=dev, requires_grad=True)
print("autograd.Function.apply (no-op):     %.2f us" % t(lambda: F.apply(x)))
