"""Dual-number kernels (csrc/dual.hip) and the forward-over-reverse force training path (alignn_amd/ff2.py) against
torch float64 autograd (jvp + reverse) of the same formulas, and against the composed twice-differentiable path."""

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from alignn_amd import GraphBatch, ff2, ops  # noqa: E402
from alignn_amd.graph import build_csr  # noqa: E402
from alignn_amd.synthetic import make_batch  # noqa: E402
from tests.helpers import rel_err  # noqa: E402

DEV = "cuda"


def _f(t):
    return t.float().to(DEV).contiguous()


@pytest.mark.parametrize("rows,Fw,res", [(37, 64, True), (300, 256, True), (50, 256, False), (9, 512, True)])
def test_layernorm_silu_dual_forward_and_reverse(rows, Fw, res):
    g = torch.Generator().manual_seed(rows + Fw)
    x = torch.randn(rows, Fw, generator=g, dtype=torch.float64) * 1.5 + 0.3
    t = torch.randn(rows, Fw, generator=g, dtype=torch.float64)
    r = torch.randn(rows, Fw, generator=g, dtype=torch.float64)
    rt = torch.randn(rows, Fw, generator=g, dtype=torch.float64)
    gam = 1 + 0.2 * torch.randn(Fw, generator=g, dtype=torch.float64)
    bet = 0.2 * torch.randn(Fw, generator=g, dtype=torch.float64)
    gy = torch.randn(rows, Fw, generator=g, dtype=torch.float64)
    gyt = torch.randn(rows, Fw, generator=g, dtype=torch.float64)

    def fn(x_, gam_, bet_):
        return F.silu(F.layer_norm(x_, (Fw,), gam_, bet_, 1e-5))

    def dual(x_, t_, gam_, bet_):
        """value and tangent written out with differentiable torch operations (torch.func.jvp and
        torch.autograd.functional.jvp(create_graph=True) both return tangents whose .backward() w.r.t. the PRIMAL input
        is wrong - checked against finite differences -, so the reference tangent is explicit)"""
        mean = x_.mean(1, keepdim=True)
        rho = (((x_ - mean) ** 2).mean(1, keepdim=True) + 1e-5).rsqrt()
        xh = (x_ - mean) * rho
        th = rho * (t_ - t_.mean(1, keepdim=True) - xh * (xh * t_).mean(1, keepdim=True))
        z, zt = gam_ * xh + bet_, gam_ * th
        sg = torch.sigmoid(z)
        return z * sg, (sg + z * sg * (1 - sg)) * zt

    xr, tr, gr, br = (v.clone().requires_grad_(True) for v in (x, t, gam, bet))
    y, yt = dual(xr, tr, gr, br)
    assert rel_err(y, fn(x, gam, bet)) < 1e-12
    assert rel_err(yt, (fn(x + 1e-6 * t, gam, bet) - fn(x - 1e-6 * t, gam, bet)) / 2e-6) < 1e-7  # the tangent IS the derivative
    if res:
        y, yt = y + r, yt + rt
    ((y * gy).sum() + (yt * gyt).sum()).backward()

    X = ff2.Dual(_f(x), _f(t))
    Y, stats = ff2._ln_fwd(X, ff2.Dual(_f(r), _f(rt)) if res else None, _f(gam), _f(bet))
    assert rel_err(Y.p, y) < 2e-6 and rel_err(Y.t, yt) < 2e-6
    GX, red = ff2._ln_bwd(ff2.Dual(_f(gy), _f(gyt)), X, _f(gam), _f(bet), stats)
    assert rel_err(GX.p, xr.grad) < 2e-5 and rel_err(GX.t, tr.grad) < 2e-5
    assert rel_err(red[0], br.grad) < 2e-5 and rel_err(red[1], gr.grad) < 2e-5


@pytest.mark.parametrize("H,n,m,seed", [(16, 9, 40, 0), (256, 120, 1500, 1), (64, 1, 5, 2)])
def test_gate_pass_dual_forward_and_reverse(H, n, m, seed):
    lib = ops._lib.load()
    g = torch.Generator().manual_seed(seed)
    u = torch.randint(0, n, (m,), generator=g)
    v = torch.randint(0, max(n - 1, 1), (m,), generator=g)  # last node isolated when n > 1
    csr = build_csr(u.to(DEV), v.to(DEV), n)
    mk = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)  # noqa: E731
    P, Pt, C, Ct = mk(n, 4 * H), mk(n, 4 * H), mk(m, H), mk(m, H)
    w_xp, w_xpt, w_m, w_mt = mk(n, H), mk(n, H), mk(m, H), mk(m, H)
    perm, inv = csr.perm.cpu(), csr.inv.cpu()

    def fn(P_, C_):
        A, Bd, Bh, Ux = P_[:, :H], P_[:, H:2 * H], P_[:, 2 * H:3 * H], P_[:, 3 * H:]
        mm = A[u] + Bd[v] + C_
        sg = torch.sigmoid(mm)
        s1 = torch.zeros(n, H, dtype=P_.dtype).index_add(0, v, sg * Bh[u])
        s0 = torch.zeros(n, H, dtype=P_.dtype).index_add(0, v, sg)
        return Ux + s1 / (s0 + 1e-6), mm

    def dual(P_, Pt_, C_, Ct_):
        """value and tangent as explicit differentiable torch operations (see the LayerNorm test)"""
        A, Bd, Bh, Ux = P_[:, :H], P_[:, H:2 * H], P_[:, 2 * H:3 * H], P_[:, 3 * H:]
        At, Bdt, Bht, Uxt = Pt_[:, :H], Pt_[:, H:2 * H], Pt_[:, 2 * H:3 * H], Pt_[:, 3 * H:]
        mm, mmt = A[u] + Bd[v] + C_, At[u] + Bdt[v] + Ct_
        sg = torch.sigmoid(mm)
        sgt = sg * (1 - sg) * mmt
        z = lambda: torch.zeros(n, H, dtype=P_.dtype)  # noqa: E731
        s1, s0 = z().index_add(0, v, sg * Bh[u]), z().index_add(0, v, sg)
        s1t, s0t = z().index_add(0, v, sgt * Bh[u] + sg * Bht[u]), z().index_add(0, v, sgt)
        h = s1 / (s0 + 1e-6)
        return (Ux + h, mm), (Uxt + (s1t - h * s0t) / (s0 + 1e-6), mmt)

    Pr, Ptr, Cr, Ctr = (t.clone().requires_grad_(True) for t in (P, Pt, C, Ct))
    (xp, mm), (xpt, mmt) = dual(Pr, Ptr, Cr, Ctr)
    fd = [(a_ - b_) / 2e-6 for a_, b_ in zip(fn(P + 1e-6 * Pt, C + 1e-6 * Ct), fn(P - 1e-6 * Pt, C - 1e-6 * Ct))]
    assert rel_err(xpt, fd[0]) < 1e-6 and rel_err(mmt, fd[1]) < 1e-7
    ((xp * w_xp).sum() + (xpt * w_xpt).sum() + (mm * w_m).sum() + (mmt * w_mt).sum()).backward()

    Pd = ff2.Dual(_f(P), _f(Pt))
    Md = ff2.Dual(_f(C[perm]), _f(Ct[perm]))  # canonical slot order
    xpre = ff2.Dual(torch.empty(n, H, device=DEV), torch.empty(n, H, device=DEV))
    s0, hh, s0t, hht = (torch.empty(n, H, device=DEV) for _ in range(4))
    ff2.check(lib.alignn_egc_gate_dual_fwd(ff2.ptr(Pd.p), ff2.ptr(Pd.t), ff2.ptr(Md.p), ff2.ptr(Md.t), ff2.ptr(csr.seg_ptr),
                                           ff2.ptr(csr.seg_node), ff2.ptr(csr.src), n, m, H, ff2.ptr(xpre.p), ff2.ptr(xpre.t),
                                           ff2.ptr(s0), ff2.ptr(hh), ff2.ptr(s0t), ff2.ptr(hht), ff2.stream()), "gate")
    assert rel_err(xpre.p, xp) < 2e-5 and rel_err(xpre.t, xpt) < 2e-5
    assert rel_err(Md.p[inv], mm) < 2e-6 and rel_err(Md.t[inv], mmt) < 2e-6
    # reverse: adjoints of (xpre, xpre_t) = (w_xp, w_xpt), of (m, mt) = (w_m, w_mt) as the "LayerNorm branch" input
    q1, q0, q1t, q0t = (torch.empty(n, H, device=DEV) for _ in range(4))
    gxp, gxpt = _f(w_xp), _f(w_xpt)
    ff2.check(lib.alignn_egc_node_dual_bwd(ff2.ptr(gxp), ff2.ptr(gxpt), H, ff2.ptr(s0), ff2.ptr(hh), ff2.ptr(s0t), ff2.ptr(hht),
                                           ff2.ptr(q1), ff2.ptr(q0), ff2.ptr(q1t), ff2.ptr(q0t), n, H, ff2.stream()), "node")
    GL = ff2.Dual(_f(w_m[perm]), _f(w_mt[perm]))
    GM = ff2.Dual(torch.empty(m, H, device=DEV), torch.empty(m, H, device=DEV))
    GP = ff2.Dual(torch.zeros(n, 4 * H, device=DEV), torch.zeros(n, 4 * H, device=DEV))
    slabs = lib.alignn_dual_slabs(n)
    gb = torch.empty(slabs, H, device=DEV)
    ff2.check(lib.alignn_egc_dual_bwd_dst(ff2.ptr(GL.p), ff2.ptr(GL.t), ff2.ptr(Md.p), ff2.ptr(Md.t), ff2.ptr(Pd.p), ff2.ptr(Pd.t),
                                          ff2.ptr(q1), ff2.ptr(q0), ff2.ptr(q1t), ff2.ptr(q0t), ff2.ptr(csr.seg_ptr),
                                          ff2.ptr(csr.seg_node), ff2.ptr(csr.src), n, H, ff2.ptr(GM.p), ff2.ptr(GM.t),
                                          ff2.ptr(GP.p), ff2.ptr(GP.t), ff2.ptr(gb), None, None, ff2.stream()), "dst")
    ff2.check(lib.alignn_egc_dual_bwd_src(ff2.ptr(GM.p), ff2.ptr(GM.t), ff2.ptr(Md.p), ff2.ptr(Md.t), ff2.ptr(q1), ff2.ptr(q1t),
                                          ff2.ptr(csr.out_ptr), ff2.ptr(csr.out_slot), ff2.ptr(csr.dst), n, H, ff2.ptr(GP.p),
                                          ff2.ptr(GP.t), None, ff2.stream()), "src")
    GP.p[:, 3 * H:] = gxp  # the Ux block is the adjoint of xpre itself
    GP.t[:, 3 * H:] = gxpt
    fl = 1e-2 * float(Pr.grad.abs().max())
    assert rel_err(GM.p[inv], Cr.grad, fl) < 5e-5 and rel_err(GM.t[inv], Ctr.grad, fl) < 5e-5
    assert rel_err(GP.p, Pr.grad, fl) < 5e-5 and rel_err(GP.t, Ptr.grad, fl) < 5e-5
    assert rel_err(gb.sum(0), Cr.grad.sum(0), fl) < 5e-5


def test_forward_over_reverse_equals_reverse_over_reverse():
    """ForcesFn (dual pass) against the composed twice-differentiable path: same energies / forces / stresses, same
    parameter gradients of an energy + force + stress loss."""
    from alignn_amd import ALIGNNAtomWise, ALIGNNAtomWiseConfig
    from alignn_amd import alignn_atomwise as AW

    raw = make_batch(3, 14, seed0=31)
    raw.r[0] *= 0.3  # one short bond: the penalty branch contributes to the forces
    raw.r[1] *= 0.3
    batch = GraphBatch.from_raw(raw, device=DEV)
    gen = torch.Generator().manual_seed(4)
    te, tf, ts = (torch.randn(3, generator=gen).to(DEV), torch.randn(raw.num_nodes, 3, generator=gen).to(DEV),
                  torch.randn(3, 3, 3, generator=gen).to(DEV))
    outs = []
    for fused in (True, False):
        torch.manual_seed(7)
        cfg = ALIGNNAtomWiseConfig(name="alignn_atomwise", alignn_layers=2, gcn_layers=2, hidden_features=64,
                                   embedding_features=32, atom_input_features=92, calculate_gradient=True,
                                   stresswise_weight=0.05)
        model = ALIGNNAtomWise(cfg).to(DEV).train()
        with torch.no_grad():
            for k, p in model.named_parameters():
                if ".bn_" in k or ".layer.1." in k:
                    p.add_(0.1 * torch.randn_like(p))
        AW.FUSED_FORCE_TRAINING = fused
        try:
            res = model(batch)
            L = F.l1_loss
            loss = L(res["out"], te) + L(res["grad"], tf) + 0.05 * L(res["stresses"], ts)
            loss.backward()
        finally:
            AW.FUSED_FORCE_TRAINING = True
        outs.append((res, loss.item(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
    (ra, la, ga), (rb, lb, gb) = outs
    assert rel_err(ra["out"], rb["out"]) < 1e-5 and rel_err(ra["grad"], rb["grad"]) < 1e-4
    assert rel_err(ra["stresses"], rb["stresses"]) < 1e-4 and abs(la - lb) < 1e-5
    assert ga.keys() == gb.keys() and len(ga) > 60
    gmax = max(float(v.abs().max()) for v in gb.values())
    worst = 0.0
    for k in ga:
        e = float((ga[k] - gb[k]).abs().max()) / max(float(gb[k].abs().max()), 1e-3 * gmax)
        worst = max(worst, e)
        assert e < 2e-3, (k, e)
    print("forward-over-reverse vs reverse-over-reverse: worst per-parameter gradient difference", worst)


@pytest.mark.parametrize("shape", ["small", "deg_over_16"])
def test_dense_block_dual_reverse_equals_the_two_pass_kernels(shape):
    """alignn_egc_dual_bwd_lg_dense (one pass over a line graph's dense blocks, 6 row passes) against alignn_egc_dual_bwd_dst
    + _src (10): the same parameter gradients of an energy + force + stress loss up to summation order - on a small batch
    and on one whose atoms have more than 16 in-edges (the kernel then takes several passes of 16 sources)."""
    from alignn_amd import ALIGNNAtomWise, ALIGNNAtomWiseConfig

    raw = make_batch(3, 14, seed0=31) if shape == "small" else make_batch(4, 3, seed0=11)  # tiny cells: many self images, up to 19 in-edges
    batch = GraphBatch.from_raw(raw, device=DEV)
    if shape == "deg_over_16":
        sp = batch.g.seg_ptr.long()
        assert int((sp[1:] - sp[:-1]).max()) > 16, "this case is meant to exceed 16 in-edges per atom"
    assert batch.lg.dense_max_src > 0
    gen = torch.Generator().manual_seed(4)
    B = raw.batch_size
    te, tf, ts = (torch.randn(B, generator=gen).to(DEV), torch.randn(raw.num_nodes, 3, generator=gen).to(DEV),
                  torch.randn(B, 3, 3, generator=gen).to(DEV))
    outs = []
    for dense in (True, False):
        torch.manual_seed(7)
        cfg = ALIGNNAtomWiseConfig(name="alignn_atomwise", alignn_layers=2, gcn_layers=1, hidden_features=64,
                                   embedding_features=32, atom_input_features=92, calculate_gradient=True,
                                   stresswise_weight=0.05)
        model = ALIGNNAtomWise(cfg).to(DEV).train()
        ff2.DENSE_LG_REVERSE = dense
        try:
            res = model(batch)
            loss = F.l1_loss(res["out"], te) + F.l1_loss(res["grad"], tf) + 0.05 * F.l1_loss(res["stresses"], ts)
            loss.backward()
            torch.cuda.synchronize()
        finally:
            ff2.DENSE_LG_REVERSE = True
        outs.append({k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
    ga, gb = outs
    assert ga.keys() == gb.keys() and len(ga) > 40
    gmax = max(float(v.abs().max()) for v in gb.values())
    worst = 0.0
    for k in ga:
        e = float((ga[k] - gb[k]).abs().max()) / max(float(gb[k].abs().max()), 1e-3 * gmax)
        worst = max(worst, e)
        assert e < 2e-5, (k, e)
    print(f"dense-block dual reverse vs two passes ({shape}): worst per-parameter gradient difference {worst:.2e}")


@pytest.mark.parametrize("shape,hidden,path", [("small", 64, "c"), ("deg_over_16", 64, "c"), ("medium", 256, "c"), ("medium", 256, "ops"),
                                               ("small", 64, "ops")])
def test_edge_layernorm_inside_the_gate_passes_equals_the_separate_kernels(shape, hidden, path, monkeypatch):
    """csrc/convln.hip (the edge LayerNorm of a line-graph convolution formed inside the gate passes: forward, reverse, dual
    forward, dual reverse) against the separate LayerNorm kernels + gate passes it replaces (ALIGNN_AMD_LN_FUSED=0): energies,
    forces, stresses and every parameter gradient of an energy + force + stress loss, up to summation order.  On the whole-model
    C calls and on the per-operator path; with atoms of more than 16 in-edges (several passes of the dense reverse kernels); and
    at 2 x 60 atoms / 256 features, where the edge projection adds the gathered rows in its epilogue and the forward pass with
    the LayerNorm inside applies."""
    from alignn_amd import ALIGNNAtomWise, ALIGNNAtomWiseConfig, cmodel

    raw = {"small": lambda: make_batch(3, 14, seed0=31), "deg_over_16": lambda: make_batch(4, 3, seed0=11),
           "medium": lambda: make_batch(2, 60, seed0=5)}[shape]()
    batch = GraphBatch.from_raw(raw, device=DEV)
    gen = torch.Generator().manual_seed(4)
    B = raw.batch_size
    te, tf, ts = (torch.randn(B, generator=gen).to(DEV), torch.randn(raw.num_nodes, 3, generator=gen).to(DEV),
                  torch.randn(B, 3, 3, generator=gen).to(DEV))
    monkeypatch.setattr(cmodel, "ENABLED", path == "c")
    outs = []
    for fused in ("0", "2"):  # (2: also on line graphs small enough for the separate kernels to be the default)
        monkeypatch.setenv("ALIGNN_AMD_LN_FUSED", fused)
        torch.manual_seed(7)
        cfg = ALIGNNAtomWiseConfig(name="alignn_atomwise", alignn_layers=2, gcn_layers=1, hidden_features=hidden,
                                   embedding_features=32, atom_input_features=92, calculate_gradient=True,
                                   stresswise_weight=0.05)
        model = ALIGNNAtomWise(cfg).to(DEV).train()
        before = dict(cmodel.STATS)
        res = model(batch)
        loss = F.l1_loss(res["out"], te) + F.l1_loss(res["grad"], tf) + 0.05 * F.l1_loss(res["stresses"], ts)
        loss.backward()
        torch.cuda.synchronize()
        if path == "c":
            assert cmodel.STATS["ff_eval"] == before.get("ff_eval", 0) + 1, "the whole-model C calls did not take this model"
        o = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
        o.update({"out": res["out"].detach().clone(), "grad": res["grad"].detach().clone(), "stresses": res["stresses"].detach().clone()})
        outs.append(o)
    ga, gb = outs
    assert ga.keys() == gb.keys() and len(ga) > 40
    gmax = max(float(v.abs().max()) for k, v in gb.items() if k not in ("out", "grad", "stresses"))
    worst = 0.0
    for k in ga:
        e = float((ga[k] - gb[k]).abs().max()) / max(float(gb[k].abs().max()), 1e-3 * gmax)
        worst = max(worst, e)
        assert e < 2e-5, (k, e)
    assert any(not torch.equal(ga[k], gb[k]) for k in ga), "both runs took the same kernels"
    print(f"edge LayerNorm inside the gate passes vs separate kernels ({shape}, H={hidden}, {path}): worst difference {worst:.2e}")


def test_tangent_only_dual_forward_equals_the_full_dual_forward():
    """ff2.REUSE_FORWARD: the dual forward takes its VALUES from the force evaluation that preceded it (ops.FORWARD_TAPE) and
    computes tangents only - same parameter gradients as the dual forward that recomputes both, up to the rounding of two
    different kernels writing the same value (1e-5 of every gradient's scale)."""
    from alignn_amd import ALIGNNAtomWise, ALIGNNAtomWiseConfig

    raw = make_batch(6, 40, seed0=17)  # E = 3 k, T = 40 k rows
    batch = GraphBatch.from_raw(raw, device=DEV)
    gen = torch.Generator().manual_seed(4)
    B = raw.batch_size
    te, tf, ts = (torch.randn(B, generator=gen).to(DEV), torch.randn(raw.num_nodes, 3, generator=gen).to(DEV),
                  torch.randn(B, 3, 3, generator=gen).to(DEV))
    outs = []
    for reuse in (True, False):
        torch.manual_seed(7)
        cfg = ALIGNNAtomWiseConfig(name="alignn_atomwise", alignn_layers=2, gcn_layers=2, hidden_features=128,
                                   embedding_features=32, atom_input_features=92, calculate_gradient=True,
                                   stresswise_weight=0.05)
        model = ALIGNNAtomWise(cfg).to(DEV).train()
        ff2.REUSE_FORWARD = reuse
        try:
            res = model(batch)
            loss = F.l1_loss(res["out"], te) + F.l1_loss(res["grad"], tf) + 0.05 * F.l1_loss(res["stresses"], ts)
            loss.backward()
            torch.cuda.synchronize()
        finally:
            ff2.REUSE_FORWARD = True
        assert ops.FORWARD_TAPE is None
        outs.append({k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
    ga, gb = outs
    assert ga.keys() == gb.keys() and len(ga) > 60
    gmax = max(float(v.abs().max()) for v in gb.values())
    worst = 0.0
    for k in ga:
        e = float((ga[k] - gb[k]).abs().max()) / max(float(gb[k].abs().max()), 1e-3 * gmax)
        worst = max(worst, e)
        assert e < 2e-5, (k, e)
    print(f"tangent-only dual forward vs full dual forward: worst per-parameter gradient difference {worst:.2e}")


@pytest.mark.parametrize("shape,H", [("small", 256), ("deg_over_16", 64)])
def test_reverse_kernels_with_the_layernorm_inside_against_the_separate_kernels(shape, H):
    """csrc/convln.hip at the kernel level: alignn_egc_bwd_lg_dense_ln and alignn_egc_dual_bwd_lg_dense_ln against
    alignn_ln_silu_bwd + alignn_egc_bwd_lg_dense(MODE 2) and alignn_ln_silu_dual_bwd + alignn_egc_dual_bwd_lg_dense on the
    same random operands - every output: the edge adjoints GM (GMt), all blocks of GP (GPt) the kernels write, the bias-gradient
    slabs and the LayerNorm parameter gradients - and every launch variant of the two kernels (ALIGNN_AMD_LN_REV)."""
    import os

    from alignn_amd import _lib
    from alignn_amd.ops import ptr, stream

    lib = _lib.load()
    raw = make_batch(3, 14, seed0=31) if shape == "small" else make_batch(4, 3, seed0=11)
    lg = GraphBatch.from_raw(raw, device=DEV).lg
    n, m = lg.n_nodes, lg.n_edges
    groups = lg.grp_seg_ptr.numel() - 1
    g = torch.Generator(device=DEV).manual_seed(2)
    R = lambda *s: torch.randn(*s, device=DEV, generator=g)  # noqa: E731
    M, Mt, GY, GYt = R(m, H), R(m, H), R(m, H), R(m, H)
    P, Pt = R(n, 4 * H), R(n, 4 * H)
    q1, q0, q1t, q0t = R(n, H), R(n, H), R(n, H), R(n, H)
    gamma, beta = 1 + 0.1 * R(H), 0.1 * R(H)
    e_stat = torch.stack([M.mean(1), 1.0 / torch.sqrt(M.var(1, unbiased=False) + 1e-5)], 1).contiguous()
    st = stream()

    def close(a, b, what, tol=2e-5):
        e = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)
        assert e < tol, (what, e)

    # ---- value reverse
    GL, GMs, GPs = torch.empty(m, H, device=DEV), torch.empty(m, H, device=DEV), torch.zeros(n, 4 * H, device=DEV)
    gbs = torch.empty(groups, H, device=DEV)
    vslabs = lib.alignn_ln_slabs(m)
    vpart, red_s = torch.empty(vslabs, 2, H, device=DEV), torch.empty(2, H, device=DEV)
    assert lib.alignn_ln_silu_bwd(ptr(GY), H, ptr(M), H, ptr(gamma), ptr(beta), ptr(e_stat), ptr(GL), H, ptr(vpart), m, H, None, st) == 0
    assert lib.alignn_bn_bwd_finalize(ptr(vpart), vslabs, H, ptr(red_s), st) == 0
    assert lib.alignn_egc_bwd_lg_dense(ptr(GL), ptr(M), ptr(P), ptr(q1), ptr(q0), None, None, 0, m, ptr(lg.grp_seg_ptr),
                                       ptr(lg.grp_src_ptr), groups, lg.dense_max_src, ptr(lg.seg_ptr), ptr(lg.seg_node), H, ptr(GMs),
                                       ptr(GPs), ptr(gbs), None, None, st) == 0
    prev = os.environ.get("ALIGNN_AMD_LN_REV")
    try:
        for v in "012":
            os.environ["ALIGNN_AMD_LN_REV"] = v + "0"
            GMf, GPf = torch.empty(m, H, device=DEV), torch.zeros(n, 4 * H, device=DEV)
            gbf, lnp, red_f = torch.empty(groups, H, device=DEV), torch.empty(groups, 2, H, device=DEV), torch.empty(2, H, device=DEV)
            assert lib.alignn_egc_bwd_lg_dense_ln(ptr(GY), ptr(M), ptr(P), ptr(q1), ptr(q0), ptr(gamma), ptr(beta), ptr(e_stat), m,
                                                  ptr(lg.grp_seg_ptr), ptr(lg.grp_src_ptr), groups, lg.dense_max_src, ptr(lg.seg_ptr),
                                                  ptr(lg.seg_node), H, ptr(GMf), ptr(GPf), ptr(gbf), ptr(lnp), None, None, st) == 0
            assert lib.alignn_bn_bwd_finalize(ptr(lnp), groups, H, ptr(red_f), st) == 0
            torch.cuda.synchronize()
            close(GMf, GMs, "GM " + v)
            close(GPf[:, :3 * H], GPs[:, :3 * H], "GP " + v)
            close(gbf.sum(0), gbs.sum(0), "bias gradient " + v)
            close(red_f, red_s, "LayerNorm parameter gradients " + v, 1e-4)
        # ---- dual reverse
        GLp, GLt = torch.empty(m, H, device=DEV), torch.empty(m, H, device=DEV)
        dslabs = lib.alignn_dual_slabs(m)
        dpart = torch.empty(dslabs, 2, H, device=DEV)
        assert lib.alignn_ln_silu_dual_bwd(ptr(GY), ptr(GYt), H, ptr(M), ptr(Mt), H, ptr(gamma), ptr(beta), ptr(e_stat), ptr(GLp),
                                           ptr(GLt), H, ptr(dpart), m, H, None, st) == 0
        assert lib.alignn_bn_bwd_finalize(ptr(dpart), dslabs, H, ptr(red_s), st) == 0
        outs_s = [torch.empty(m, H, device=DEV), torch.empty(m, H, device=DEV), torch.zeros(n, 4 * H, device=DEV),
                  torch.zeros(n, 4 * H, device=DEV)]
        assert lib.alignn_egc_dual_bwd_lg_dense(ptr(GLp), ptr(GLt), ptr(M), ptr(Mt), ptr(P), ptr(Pt), ptr(q1), ptr(q0), ptr(q1t), ptr(q0t),
                                                m, ptr(lg.grp_seg_ptr), ptr(lg.grp_src_ptr), groups, ptr(lg.seg_ptr), ptr(lg.seg_node), H,
                                                ptr(outs_s[0]), ptr(outs_s[1]), ptr(outs_s[2]), ptr(outs_s[3]), ptr(gbs), None, None,
                                                st) == 0
        for v in "012":
            os.environ["ALIGNN_AMD_LN_REV"] = "0" + v
            outs_f = [torch.empty(m, H, device=DEV), torch.empty(m, H, device=DEV), torch.zeros(n, 4 * H, device=DEV),
                      torch.zeros(n, 4 * H, device=DEV)]
            gbf, lnp, red_f = torch.empty(groups, H, device=DEV), torch.empty(groups, 2, H, device=DEV), torch.empty(2, H, device=DEV)
            assert lib.alignn_egc_dual_bwd_lg_dense_ln(ptr(GY), ptr(GYt), ptr(M), ptr(Mt), ptr(P), ptr(Pt), ptr(q1), ptr(q0), ptr(q1t),
                                                       ptr(q0t), ptr(gamma), ptr(beta), ptr(e_stat), m, ptr(lg.grp_seg_ptr),
                                                       ptr(lg.grp_src_ptr), groups, ptr(lg.seg_ptr), ptr(lg.seg_node), H, ptr(outs_f[0]),
                                                       ptr(outs_f[1]), ptr(outs_f[2]), ptr(outs_f[3]), ptr(gbf), ptr(lnp), None, None,
                                                       st) == 0
            assert lib.alignn_bn_bwd_finalize(ptr(lnp), groups, H, ptr(red_f), st) == 0
            torch.cuda.synchronize()
            for a, b, name in zip(outs_f, outs_s, ("GM", "GMt", "GP", "GPt")):
                close(a[:, :3 * H] if a.shape[1] == 4 * H else a, b[:, :3 * H] if b.shape[1] == 4 * H else b, name + " dual " + v)
            close(gbf.sum(0), gbs.sum(0), "bias gradient, dual " + v)
            close(red_f, red_s, "LayerNorm parameter gradients, dual " + v, 1e-4)
    finally:
        if prev is None:
            os.environ.pop("ALIGNN_AMD_LN_REV", None)
        else:
            os.environ["ALIGNN_AMD_LN_REV"] = prev


@pytest.mark.parametrize("shape,H", [("small", 256), ("deg_over_16", 64)])
def test_forward_kernels_with_the_layernorm_inside_against_the_separate_kernels(shape, H):
    """csrc/convln.hip, forward side: alignn_egc_gate_fwd_pre_ln against alignn_egc_gate_fwd_pre + alignn_ln_silu_fwd, and
    alignn_egc_gate_dual_tan_ln against alignn_egc_gate_dual_fwd_tangent + alignn_ln_silu_dual_fwd (tangent only): edge output,
    row statistics, the node sums and pre-activations, the tangents."""
    from alignn_amd import _lib
    from alignn_amd.ops import ptr, stream

    lib = _lib.load()
    raw = make_batch(3, 14, seed0=31) if shape == "small" else make_batch(4, 3, seed0=11)
    lg = GraphBatch.from_raw(raw, device=DEV).lg
    n, m = lg.n_nodes, lg.n_edges
    g = torch.Generator(device=DEV).manual_seed(5)
    R = lambda *s: torch.randn(*s, device=DEV, generator=g)  # noqa: E731
    M, Y, Yt, Ct = R(m, H), R(m, H), R(m, H), R(m, H)
    P, Pt = R(n, 4 * H), R(n, 4 * H)
    gamma, beta = 1 + 0.1 * R(H), 0.1 * R(H)
    st = stream()
    E = lambda *s: torch.empty(*s, device=DEV)  # noqa: E731

    def close(a, b, what, tol=1e-5):
        e = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)
        assert e < tol, (what, e)

    # ---- values
    xs, s0s, hhs, ys, sts = E(n, H), E(n, H), E(n, H), E(m, H), E(m, 2)
    assert lib.alignn_egc_gate_fwd_pre(ptr(P), ptr(M), ptr(lg.seg_ptr), ptr(lg.seg_node), ptr(lg.src), n, m, H, ptr(xs), ptr(s0s),
                                       ptr(hhs), None, None, st) == 0
    assert lib.alignn_ln_silu_fwd(ptr(M), H, ptr(Y), H, ptr(gamma), ptr(beta), 1e-5, ptr(ys), H, ptr(sts), m, H, None, st) == 0
    xf, s0f, hhf, yf, stf, am = E(n, H), E(n, H), E(n, H), E(m, H), E(m, 2), torch.zeros(2, device=DEV)
    assert lib.alignn_egc_gate_fwd_pre_ln(ptr(P), ptr(M), ptr(lg.seg_ptr), ptr(lg.seg_node), ptr(lg.src), n, m, H, ptr(xf), ptr(s0f),
                                          ptr(hhf), ptr(gamma), ptr(beta), 1e-5, ptr(Y), ptr(yf), ptr(stf), ptr(am), st) == 0
    torch.cuda.synchronize()
    assert torch.equal(xf, xs) and torch.equal(s0f, s0s) and torch.equal(hhf, hhs)  # (same sums in the same order)
    close(yf, ys, "edge output")
    close(stf, sts, "row statistics")
    assert abs(float(am[0]) - float(yf.abs().max())) == 0.0
    # ---- tangents (values known: M, the node sums and the row statistics of the pass above)
    Mt_s, Mt_f = Ct.clone(), Ct.clone()
    xts, s0ts, hhts, yts, st2 = E(n, H), E(n, H), E(n, H), E(m, H), E(m, 2)
    assert lib.alignn_egc_gate_dual_fwd_tangent(ptr(P), ptr(Pt), ptr(M), ptr(Mt_s), ptr(lg.seg_ptr), ptr(lg.seg_node), ptr(lg.src), n, m,
                                                H, ptr(xts), ptr(s0s), ptr(hhs), ptr(s0ts), ptr(hhts), st) == 0
    assert lib.alignn_ln_silu_dual_fwd(ptr(M), ptr(Mt_s), H, ptr(Y), ptr(Yt), H, ptr(gamma), ptr(beta), 1e-5, None, ptr(yts), H, ptr(st2),
                                       m, H, None, st) == 0
    xtf, s0tf, hhtf, ytf, am2 = E(n, H), E(n, H), E(n, H), E(m, H), torch.zeros(2, device=DEV)
    assert lib.alignn_egc_gate_dual_tan_ln(ptr(P), ptr(Pt), ptr(M), ptr(Mt_f), ptr(lg.seg_ptr), ptr(lg.seg_node), ptr(lg.src), n, m, H,
                                           ptr(xtf), ptr(s0s), ptr(hhs), ptr(s0tf), ptr(hhtf), ptr(gamma), ptr(beta), ptr(stf), ptr(Yt),
                                           ptr(ytf), ptr(am2), st) == 0
    torch.cuda.synchronize()
    assert torch.equal(Mt_f, Mt_s) and torch.equal(xtf, xts) and torch.equal(s0tf, s0ts) and torch.equal(hhtf, hhts)
    close(ytf, yts, "tangent of the edge output")
    assert float(am2[0]) == 0.0 and abs(float(am2[1]) - float(ytf.abs().max())) == 0.0
