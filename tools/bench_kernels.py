"""Micro-benchmarks of individual C-ABI kernels at BASELINE config-2 sizes (HIP-event timing).
usage (GPU box): python tools/bench_kernels.py [gemm|conv|all]"""

import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alignn_amd import ops  # noqa: E402

DEV = "cuda"
T, E, N, H = 676200, 50712, 3840, 256


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def gemms():
    for (M, Nn, K) in [(T, 256, 256), (E, 1024, 256), (E, 256, 1024), (E, 256, 256), (T, 256, 64), (N, 1024, 256), (N, 256, 1024), (N, 256, 256)]:
        a = torch.randn(M, K, device=DEV)
        w = torch.randn(Nn, K, device=DEV) / K**0.5
        b = torch.randn(Nn, device=DEV)
        add = torch.randn(M, Nn, device=DEV)
        out = torch.empty(M, Nn, device=DEV)
        fl = 2.0 * M * Nn * K
        by = 4.0 * (M * K + M * Nn)
        t32 = timeit(lambda: ops.gemm_nt(a, w, b, out=out))
        ws = ops.split_bf16x3(w)
        t6 = timeit(lambda: ops.gemm_nt_x6(a, ws, b, out=out))
        t6a = timeit(lambda: ops.gemm_nt_x6(a, ws, b, add, out=out))
        tsp = timeit(lambda: ops.split_bf16x3(w))
        am = ops.absmax(a)
        wh = ops.split_f16x2(w)
        th = timeit(lambda: ops.gemm_nt_f16x3(a, am, wh, b, out=out))
        tha = timeit(lambda: ops.gemm_nt_f16x3(a, am, wh, b, add, out=out))
        tam = timeit(lambda: ops.absmax(a))
        print(f"NT M={M} N={Nn} K={K}: fp32 {t32*1e3:8.1f} us ({fl/t32/1e9:6.1f} TF) | x6 {t6*1e3:8.1f} us ({fl/t6/1e9:6.1f} TF-eq, "
              f"{by/t6/1e6:6.0f} GB/s) | x6+addend {t6a*1e3:8.1f} us | split {tsp*1e3:6.1f} us | f16x3 {th*1e3:8.1f} us "
              f"({by/th/1e6:6.0f} GB/s) +addend {tha*1e3:8.1f} us | absmax {tam*1e3:6.1f} us")
    for (M, Nn, K) in [(T, 256, 256), (E, 1024, 256), (E, 256, 256), (N, 1024, 256), (T, 256, 64), (T, 64, 40), (E, 64, 80)]:
        g = torch.randn(M, Nn, device=DEV)
        a = torch.randn(M, K, device=DEV)
        t = timeit(lambda: ops.gemm_tn(g, a))
        gm, am = ops.absmax(g), ops.absmax(a)
        th = timeit(lambda: ops.gemm_tn(g, a, gm, am))
        print(f"TN M={M} N={Nn} K={K}: {t*1e3:8.1f} us ({2.0*M*Nn*K/t/1e9:6.1f} TF) | f16x3 {th*1e3:8.1f} us")


if __name__ == "__main__":
    gemms()
