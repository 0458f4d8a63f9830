"""The fused input-gradient + weight-gradient pass (csrc/gemm_dw.hip) against the two launches it replaces (same bits for the
input gradient) and against float64, with its timing next to theirs.
usage: python tools/dw_check.py [rows ...]      (default: 4096+17, 65 536+1, 676 200)"""
import os, sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alignn_amd import ops  # noqa: E402


def t1(fn, n=10):
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


def rel(x, ref):
    return float((x.double() - ref).abs().max() / ref.abs().max())


def check(T, H=256, seed=0, timing=True):
    g = torch.Generator().manual_seed(seed)
    # element (r, c) patterns that a transposed / permuted operand would not reproduce
    gm = (torch.randn(T, H, generator=g) * (1 + torch.arange(H) / 64.0)).cuda()
    y = (torch.randn(T, H, generator=g) + 0.3 * torch.sin(torch.arange(H) * 0.37)).cuda()
    w = (torch.randn(H, H, generator=g) / 16).cuda()
    res = torch.randn(T, H, generator=g).cuda()
    xn = (torch.randn(T, H, generator=g) * 1.3 + 0.2).cuda()
    gamma, beta = (1 + 0.1 * torch.randn(H, generator=g)).cuda(), (0.1 * torch.randn(H, generator=g)).cuda()
    mean, var = xn.mean(0), xn.var(0, unbiased=False)
    rstd = torch.rsqrt(var + 1e-5)
    stat = torch.stack([mean, rstd, gamma * rstd, beta]).contiguous()
    g_amax, y_amax = ops.absmax(gm), ops.absmax(y)
    wt = ops.split_f16x2(w, True)
    dw64 = gm.double().t() @ y.double()
    z = (xn.double() - mean.double()) * stat[2].double() + beta.double()
    sg = torch.sigmoid(z)
    ds = sg * (1 + z * (1 - sg))
    xhat = (xn.double() - mean.double()) * rstd.double()
    del z, sg
    worst = 0.0
    ok = True
    for addend in (None, res):
        for bn in (False, True):
            name = ("bnred" if bn else "plain") + (" + addend" if addend is not None else "")
            c64 = gm.double() @ w.double() + (addend.double() if addend is not None else 0)
            out, dW, red = ops.gemm_dgrad_wgrad(gm, g_amax, y, y_amax, wt, addend, xn if bn else None, stat if bn else None)
            # the two launches
            if bn:
                o2, red2 = ops.gemm_nt_f16x3_bnred(gm, g_amax, wt, xn, stat, None, addend)
            else:
                o2, red2 = ops.gemm_nt_f16x3(gm, g_amax, wt, None, addend), None
            dW2 = ops.gemm_tn(gm, y, g_amax, y_amax)
            torch.cuda.synchronize()
            same = bool(torch.equal(out, o2))
            e = [rel(out, c64), rel(dW, dw64), rel(dW2, dw64)]
            if bn:
                gz = c64 * ds
                e += [rel(red[0], gz.sum(0)), rel(red[1], (gz * xhat).sum(0)), rel(red2[0], gz.sum(0)), rel(red2[1], (gz * xhat).sum(0))]
                del gz
            # run-to-run reproducibility of everything the kernel writes
            out_b, dW_b, red_b = ops.gemm_dgrad_wgrad(gm, g_amax, y, y_amax, wt, addend, xn if bn else None, stat if bn else None)
            rep = torch.equal(out, out_b) and torch.equal(dW, dW_b) and (red is None or torch.equal(red, red_b))
            line = f"T={T:7d} {name:16s} g_y vs f64 {e[0]:.1e} (== two-launch bits: {same})  dW {e[1]:.1e} (two-launch {e[2]:.1e})"
            if bn:
                line += f"  sums {max(e[3], e[4]):.1e} (two-launch {max(e[5], e[6]):.1e})"
            line += f"  reproducible: {rep}"
            if timing:
                tf = t1(lambda: ops.gemm_dgrad_wgrad(gm, g_amax, y, y_amax, wt, addend, xn if bn else None, stat if bn else None, out=out))
                if bn:
                    td = t1(lambda: ops.gemm_nt_f16x3_bnred(gm, g_amax, wt, xn, stat, None, addend, out=o2))
                else:
                    td = t1(lambda: ops.gemm_nt_f16x3(gm, g_amax, wt, None, addend, out=o2))
                tw = t1(lambda: ops.gemm_tn(gm, y, g_amax, y_amax))
                line += f"   fused {tf:7.1f} us | input gradient {td:7.1f} + weight gradient {tw:7.1f} = {td + tw:7.1f} us"
            print(line, flush=True)
            worst = max(worst, *e[:2], *(e[3:5] if bn else []))
            ok = ok and same and rep
            del c64
    return worst, ok


def main(argv):
    sizes = [int(v) for v in argv] or [4096 + 17, 65536 + 1, 676200]
    worst, ok = 0.0, True
    for T in sizes:
        w, o = check(T)
        worst, ok = max(worst, w), ok and o
    good = ok and worst < 5e-6
    print("OK" if good else "FAILED", f"(worst {worst:.2e}, bound 5e-6 of the largest reference element; bit checks {ok})")
    return 0 if good else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
