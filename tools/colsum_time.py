import sys, torch
sys.path.insert(0, ".")
from alignn_amd import ops
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for rows, F in ((676200, 64), (676200, 256), (50712, 256), (50712, 1024), (3840, 256)):
    x = torch.randn(rows, F, device="cuda")
    ref = x.double().sum(0)
    out = ops.col_sum(x)
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    print(f"col_sum {rows} x {F}: {t(lambda: ops.col_sum(x)):7.1f} us   rel err {err:.1e}")
