"""BatchNorm-backward reductions in the epilogue of the input-gradient projection (alignn_gemm_nt_f16x3_bnred) against
the separate reduction kernel: same numbers, timing of both.  usage: python tools/bnred_bench.py [rows]"""
import sys

import torch

sys.path.insert(0, ".")
from alignn_amd import ops  # noqa: E402


def t(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def main(T=676200, H=256):
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    gm = torch.randn(T, H, generator=g).to(dev)
    w = (torch.randn(H, H, generator=g) / 16).to(dev)
    res = torch.randn(T, H, generator=g).to(dev)
    xn = (torch.randn(T, H, generator=g) * 1.3 + 0.2).to(dev)
    gamma, beta = (1 + 0.1 * torch.randn(H, generator=g)).to(dev), (0.1 * torch.randn(H, generator=g)).to(dev)
    mean, var = xn.mean(0), xn.var(0, unbiased=False)
    rstd = torch.rsqrt(var + 1e-5)
    stat = torch.stack([mean, rstd, gamma * rstd, beta]).contiguous()
    amax = ops.absmax(gm)
    ws = ops.split_f16x2(w, transpose=True)
    for addend in (res, None):
        ref = ops.gemm_nt_f16x3(gm, amax, ws, None, addend)
        red_ref = ops._bn_silu_bwd_reduce(ref, xn, stat)
        out, red = ops.gemm_nt_f16x3_bnred(gm, amax, ws, xn, stat, None, addend)
        torch.cuda.synchronize()
        scale = red_ref.abs().max(1, keepdim=True).values
        print(f"addend={'yes' if addend is not None else 'no'}: product equal {torch.equal(out, ref)}, "
              f"reduction error {float(((red - red_ref).abs() / scale).max()):.2e} of the row maximum")
        # fp64 reference of the reductions
        z = (xn.double() - mean.double()) * stat[2].double() + beta.double()
        sg = torch.sigmoid(z)
        gz = ref.double() * (sg * (1 + z * (1 - sg)))
        r64 = torch.stack([gz.sum(0), (gz * (xn.double() - mean.double()) * rstd.double()).sum(0)])
        s64 = r64.abs().max(1, keepdim=True).values
        print(f"   vs float64: fused {float(((red.double() - r64).abs() / s64).max()):.2e}, separate kernel "
              f"{float(((red_ref.double() - r64).abs() / s64).max()):.2e}")
        buf = torch.empty_like(ref)
        t_plain = t(lambda: ops.gemm_nt_f16x3(gm, amax, ws, None, addend, out=buf))
        t_red = t(lambda: ops._bn_silu_bwd_reduce(ref, xn, stat))
        t_fused = t(lambda: ops.gemm_nt_f16x3_bnred(gm, amax, ws, xn, stat, None, addend, out=buf))
        print(f"   projection {t_plain:.0f} us + separate reduction {t_red:.0f} us = {t_plain + t_red:.0f} us;  fused (incl. slab finalize) {t_fused:.0f} us")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 676200)
