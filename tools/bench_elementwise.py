"""Timing of the element-wise BatchNorm kernels at E- and T-row sizes (HIP events), with and without amax tracking."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alignn_amd import ops
from tools.bench_kernels import timeit
for rows in (50712, 676200):
    x = torch.randn(rows, 256, device="cuda"); res = torch.randn(rows, 256, device="cuda")
    stat = torch.randn(4, 256, device="cuda"); gamma = torch.randn(256, device="cuda"); red = torch.randn(2, 256, device="cuda")
    out = torch.empty_like(x)
    for f16 in (True, False):
        ops.F16X3 = f16
        t1 = timeit(lambda: ops._bn_silu_fwd(x, res, stat))
        am = ops.new_amax(x) if f16 else None
        t2 = timeit(lambda: ops._bn_silu_bwd_apply(x, res, stat, gamma, red, False, out, am))
        t3 = timeit(lambda: ops._bn_silu_bwd_reduce(x, res, stat))
        gb = rows * 256 * 4 / 1e9
        print(f"rows={rows} amax={f16}: bn_silu_fwd {t1*1e3:7.1f} us ({3*gb/t1/1e-3/1e3:5.2f} TB/s) | bwd_apply {t2*1e3:7.1f} us | bwd_reduce(+finalize) {t3*1e3:7.1f} us")
