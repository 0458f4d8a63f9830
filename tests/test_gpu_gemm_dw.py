"""csrc/gemm_dw.hip - the input gradient and the weight gradient of a [M, 256] x [256, 256] projection in ONE pass over the
incoming gradient (alignn/models/alignn.py:101, autograd's grad_input + grad_weight of ``edge_gate``) - at the kernel level:
against float64, against the two launches it replaces (same bits for the input gradient), run to run, on ragged row counts, with
padded leading dimensions, with and without the residual addend / the BatchNorm-backward sums; and at the model level against
the two-launch path (ALIGNN_AMD_DW_FUSED=0)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

from alignn_amd import ALIGNN, ALIGNNConfig, GraphBatch, _lib, ops  # noqa: E402
from alignn_amd.synthetic import make_batch  # noqa: E402

DEV = "cuda"
H = 256


def _rel(x, ref):
    return float((x.double() - ref).abs().max() / ref.abs().max())


def _operands(M, seed, ld=H):
    g = torch.Generator().manual_seed(seed)

    def mat(scale=1.0, shift=None):
        t = torch.randn(M, H, generator=g) * scale
        if shift is not None:
            t = t + shift
        buf = torch.zeros(M, ld)
        buf[:, :H] = t
        return buf.to(DEV)[:, :H]  # (a view with leading dimension ld)

    # element (r, c) patterns that a transposed / permuted operand would not reproduce
    gm = mat(1.0) * (1 + torch.arange(H, device=DEV) / 64.0)
    y = mat(1.0, 0.3 * torch.sin(torch.arange(H) * 0.37))
    res = mat(1.0)
    xn = mat(1.3, 0.2)
    w = (torch.randn(H, H, generator=g) / 16).to(DEV)
    gamma, beta = (1 + 0.1 * torch.randn(H, generator=g)).to(DEV), (0.1 * torch.randn(H, generator=g)).to(DEV)
    mean, var = xn.mean(0), xn.var(0, unbiased=False)
    rstd = torch.rsqrt(var + 1e-5)
    stat = torch.stack([mean, rstd, gamma * rstd, beta]).contiguous()
    return gm, y, res, xn, w, stat


@pytest.mark.parametrize("M,ld", [(4096, H), (4096 + 17, H), (65536 + 1, H), (70001, H + 8), (676200, H)])
def test_fused_pass_against_float64_and_the_two_launches(M, ld):
    assert _lib.load().alignn_gemm_dgrad_wgrad_supported(M, H, H) == 1
    gm, y, res, xn, w, stat = _operands(M, M % 97, ld)
    if ld != H:
        gm = gm.contiguous()  # (the gradient and the output may be padded independently: here y, addend, xn are, gm is not)
    g_amax, y_amax = ops.absmax(gm), ops.absmax(y)
    wt = ops.split_f16x2(w, True)
    dw64 = gm.double().t() @ y.double()
    mean, rstd, beta = stat[0].double(), stat[1].double(), stat[3].double()
    z = (xn.double() - mean) * stat[2].double() + beta
    sg = torch.sigmoid(z)
    ds = sg * (1 + z * (1 - sg))
    xhat = (xn.double() - mean) * rstd
    del z, sg
    for addend in (None, res):
        c64 = gm.double() @ w.double() + (addend.double() if addend is not None else 0)
        for bn in (False, True):
            what = (M, ld, addend is not None, bn)
            out, dW, red = ops.gemm_dgrad_wgrad(gm, g_amax, y, y_amax, wt, addend, xn if bn else None, stat if bn else None)
            if bn:
                o2, _ = ops.gemm_nt_f16x3_bnred(gm, g_amax, wt, xn, stat, None, addend)
            else:
                o2 = ops.gemm_nt_f16x3(gm, g_amax, wt, None, addend)
            assert torch.equal(out, o2), what  # the input gradient: the bits of the kernel it replaces
            assert _rel(out, c64) < 2e-6, (what, _rel(out, c64))
            assert _rel(dW, dw64) < 3e-6, (what, _rel(dW, dw64))  # (fp32 FMA chains over 676 200 rows sit at 1e-4)
            if bn:
                gz = c64 * ds
                assert _rel(red[0], gz.sum(0)) < 1e-5 and _rel(red[1], (gz * xhat).sum(0)) < 1e-5, what
                del gz
            out_b, dW_b, red_b = ops.gemm_dgrad_wgrad(gm, g_amax, y, y_amax, wt, addend, xn if bn else None, stat if bn else None)
            assert torch.equal(out, out_b) and torch.equal(dW, dW_b) and (red is None or torch.equal(red, red_b)), what
        del c64


def test_rows_below_the_threshold_and_other_shapes_are_refused():
    lib = _lib.load()
    assert lib.alignn_gemm_dgrad_wgrad_supported(4095, H, H) == 0
    assert lib.alignn_gemm_dgrad_wgrad_supported(100000, 128, H) == 0 and lib.alignn_gemm_dgrad_wgrad_supported(100000, H, 512) == 0
    assert 1 <= lib.alignn_gemm_dgrad_wgrad_slabs(4096) <= 64 and lib.alignn_gemm_dgrad_wgrad_slabs(10 ** 6) % 8 == 0
    gm, y, res, xn, w, stat = _operands(4096, 1)
    wt = ops.split_f16x2(w, True)
    g_amax, y_amax = ops.absmax(gm), ops.absmax(y)
    out, dW = torch.empty_like(gm), torch.empty(H, H, device=DEV)
    nbytes = lib.alignn_gemm_dgrad_wgrad_workspace(4096)
    ws = torch.empty(nbytes // 4, device=DEV)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
    st = _lib.stream()

    def call(**kw):
        a = dict(G=gm, ldg=H, Y=y, ldy=H, ldc=H, ws=ws, nbytes=nbytes, xn=None, nstat=None, partial=None)
        a.update(kw)
        return lib.alignn_gemm_dgrad_wgrad_f16x3(p(a["G"]), a["ldg"], p(g_amax), p(a["Y"]), a["ldy"], p(y_amax), p(wt.buf), p(wt.amax), None,
                                                 0, p(out), a["ldc"], p(a["xn"]), H, p(a["nstat"]), p(a["partial"]), p(dW), H, 4096,
                                                 p(a["ws"]), a["nbytes"], st)

    assert call() == 0
    assert call(ldg=H + 1) != 0 and call(ldc=H + 2) != 0  # leading dimensions must keep rows 16-byte aligned
    assert call(ws=None) != 0 and call(nbytes=nbytes - 4) != 0  # the workspace is the caller's, and must be large enough
    assert call(xn=xn) != 0  # the BatchNorm-backward sums need all three of xn / nstat / partial
    torch.cuda.synchronize()


def test_the_model_takes_the_fused_pass_and_matches_the_two_launch_path(monkeypatch):
    """One training step of the default model at 16 x 60 atoms (T = 169 k rows >= ops.DW_MIN_ROWS): every line-graph convolution's
    edge_gate takes the fused pass; predictions equal and every parameter gradient within 1e-4 (measured 2e-5) of the two-launch path's (the
    weight gradient is summed per workgroup instead of per 256-row slab; the input gradient has the same bits)."""
    raw = make_batch(16, 60, seed0=3)
    batch = GraphBatch.from_raw(raw, device=DEV)
    assert raw.num_triplets >= ops.DW_MIN_ROWS
    target = torch.randn(16, generator=torch.Generator().manual_seed(5)).to(DEV)
    outs = []
    for fused in (True, False):
        monkeypatch.setattr(ops, "DGRAD_WGRAD_FUSED", fused)
        ops.DW_STATS["fused"] = 0
        torch.manual_seed(0)
        model = ALIGNN(ALIGNNConfig(name="alignn")).to(DEV).train()
        pred = model(batch)
        torch.nn.functional.l1_loss(pred, target).backward()
        torch.cuda.synchronize()
        outs.append((pred.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
    (pa, ga), (pb, gb) = outs
    assert torch.equal(pa, pb)
    gmax = max(float(v.abs().max()) for v in gb.values())
    worst = max((float((ga[k] - gb[k]).abs().max()) / max(float(gb[k].abs().max()), 1e-3 * gmax), k) for k in gb)
    # (both are fp32 schedules of the same sums: the BatchNorm-backward sums and dW are accumulated per workgroup instead of per
    # row tile, and eight layers of BatchNorm backward carry that 1e-6 on; against float64 both sit at 7e-6 - tests/test_gpu_full_size.py)
    assert worst[0] < 1e-4, worst
    assert any(not torch.equal(ga[k], gb[k]) for k in gb), "both runs took the same kernels"
