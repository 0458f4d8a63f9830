"""Training THROUGH the forces on fused kernels: the dual-number pass.

``ALIGNNAtomWise`` with ``calculate_gradient=True`` returns forces ``F = reduce(-dE_tot/dr)`` and virial stresses,
and the training loss differentiates through them (alignn/models/alignn_atomwise.py:512-638,
``autograd.grad(create_graph=True)``; alignn/train.py:291-387).  Reverse-over-reverse needs the second derivative of
every layer.  The same gradient has a cheaper form: with ``w_e = dL/d(pair force of bond e)`` (known once the loss has
been evaluated, i.e. in backward) the force / stress part of the loss is linear in the pair forces,

    L_FS(theta) = sum_e w_e . f_e(theta),   f_e = c dE_tot/dr_e   ==>   L_FS = c D_w E_tot(theta),

``c`` times the DIRECTIONAL derivative of the total energy along the bond-vector displacement ``w``.  So

    dL/dtheta = sum_g gE_g dE_g/dtheta  +  c d/dtheta [ D_w E_tot ]

is ONE reverse pass over a forward pass that carries, next to every activation ``p``, its tangent ``pt = D_w p``
(forward-over-reverse instead of reverse-over-reverse).  ``ForcesFn`` is that scheme as one autograd node:

* forward:  energies, forces and stresses as VALUES, by the fused first-order path (``_forward_fused``: fused forward +
  hand-written backward w.r.t. the bond vectors - what inference / MD uses);
* backward: build ``w`` and the seeds from the incoming gradients, run the dual forward (``csrc/dual.hip`` for the
  non-linear pieces, the ordinary MFMA projections for every Linear layer: the tangent of ``x W^T + b`` is
  ``xt W^T``) and its reverse, hand the parameter gradients to autograd.

Everything is computed by the HIP library; torch does index bookkeeping and a few [E,3] / [T,40] element-wise
operations for the tangents of the geometry features.  Same gradients as the composed twice-differentiable path
(``alignn_amd.ff``) and as the reference's class (goldens), ~4x faster at BASELINE configs[3].
"""

from __future__ import annotations

import torch

from . import _lib, ops
from ._lib import check, ptr, stream
from .graph import CSRGraph, GraphBatch

LN_EPS = 1e-5


def _empty(*shape, like):
    return torch.empty(*shape, dtype=torch.float32, device=like.device)


def _amax2(like):
    """two consecutive zeroed device scalars (value, tangent)"""
    a = ops._AMAX_ARENA
    if a["buf"] is None or a["next"] + 2 > a["buf"].numel() or a["buf"].device != like.device:
        ops.new_amax_arena(like.device)
    i = a["next"]
    a["next"] = i + 2
    return a["buf"][i:i + 2]


def _track(rows):
    return ops._track(rows)


DENSE_LG_REVERSE = True  # line graphs: the dual reverse of the gate pass as one dense-block kernel (tests flip it to compare)


# The dual forward repeats, value for value, what the force evaluation in ForcesFn.forward has just computed (12 T-row
# projections and the value half of every dual kernel).  With this on, that evaluation files its activations
# (ops.FORWARD_TAPE) and the dual forward computes TANGENTS only: tangent projections, alignn_egc_gate_dual_fwd_tangent,
# alignn_ln_silu_dual_fwd with Y = NULL.  Tests flip it to compare.
REUSE_FORWARD = True


class Dual:
    """value ``p`` and tangent ``t`` of one activation (+ the max|.| scalars their producer tracked, or None; ``amax_p``: the
    value's scalar when it comes from another producer than the tangent's - a value taken over from the force evaluation)"""

    __slots__ = ("p", "t", "amax", "amax_p")

    def __init__(self, p, t, amax=None, amax_p=None):
        self.p, self.t, self.amax, self.amax_p = p, t, amax, amax_p

    def am(self, i):
        if i == 0 and self.amax_p is not None:
            return self.amax_p
        return None if self.amax is None else self.amax[i:i + 1]


def _taken_over(t):
    """(tensor without autograd history, its tracked max|.| or None) of an activation of the force evaluation"""
    return t.detach(), ops.get_amax(t)


def _project(x: Dual, w, b):
    """(x W^T + b, xt W^T)"""
    return (ops.project(x.p, w, b, a_amax=x.am(0)), ops.project(x.t, w, None, a_amax=x.am(1)))


def _ln_fwd(x: Dual, res, gamma, beta, value_out=None):
    """``value_out``: the value output is known already (taken over from the force evaluation): only the tangent is written."""
    lib = _lib.load()
    rows, F = x.p.shape
    known = value_out is not None
    yp = value_out[0] if known else _empty(rows, F, like=x.p)
    yt = _empty(rows, F, like=x.p)
    stats = _empty(rows, 2, like=x.p)
    amax = _amax2(x.p) if _track(rows) else None
    check(lib.alignn_ln_silu_dual_fwd(ptr(x.p), ptr(x.t), x.p.stride(0), ptr(res.p) if res else None,
                                      ptr(res.t) if res else None, res.p.stride(0) if res else 0, ptr(gamma), ptr(beta),
                                      LN_EPS, None if known else ptr(yp), ptr(yt), F, ptr(stats), rows, F, ptr(amax), stream()),
          "ln_silu_dual_fwd")
    return Dual(yp, yt, amax, value_out[1] if known else None), stats


def _ln_bwd(g: Dual, x: Dual, gamma, beta, stats, out_p=None, out_t=None, amax=None, node=None):
    """-> (gx Dual, red [2,F] = dbeta | dgamma).  ``out_*``: write into these (possibly strided) blocks.  ``node`` = (s0, hh, s0t,
    hht, q1, q0, q1t, q0t): the reverse of the node-level quotient in the same pass (alignn_ln_silu_dual_bwd_node)."""
    lib = _lib.load()
    rows, F = x.p.shape
    if out_p is None:
        out_p, out_t = _empty(rows, F, like=x.p), _empty(rows, F, like=x.p)
    if amax is None and _track(rows):
        amax = _amax2(x.p)
    slabs = lib.alignn_dual_slabs(rows)
    partial = _empty(slabs, 2, F, like=x.p)
    if node is not None:
        check(lib.alignn_ln_silu_dual_bwd_node(ptr(g.p), ptr(g.t), g.p.stride(0), ptr(x.p), ptr(x.t), x.p.stride(0), ptr(gamma),
                                               ptr(beta), ptr(stats), ptr(out_p), ptr(out_t), out_p.stride(0), ptr(partial), rows,
                                               F, ptr(amax), *[ptr(t_) for t_ in node], stream()), "ln_silu_dual_bwd_node")
    else:
        check(lib.alignn_ln_silu_dual_bwd(ptr(g.p), ptr(g.t), g.p.stride(0), ptr(x.p), ptr(x.t), x.p.stride(0), ptr(gamma),
                                          ptr(beta), ptr(stats), ptr(out_p), ptr(out_t), out_p.stride(0), ptr(partial), rows,
                                          F, ptr(amax), stream()), "ln_silu_dual_bwd")
    red = _empty(2, F, like=x.p)
    check(lib.alignn_bn_bwd_finalize(ptr(partial), slabs, F, ptr(red), stream()), "ln_dual_finalize")
    return Dual(out_p, out_t, amax), red


class _Grads:
    """parameter -> accumulated gradient"""

    def __init__(self):
        self.g = {}

    def add(self, p, g):
        if p is None or g is None:
            return
        k = id(p)
        if k in self.g:
            self.g[k] = (p, self.g[k][1] + g)
        else:
            self.g[k] = (p, g)

    def get(self, p):
        e = self.g.get(id(p))
        return None if e is None else e[1]


def _wgrad(g: Dual, x: Dual):
    """W-bar = g^T x + gt^T xt"""
    return ops.gemm_tn(g.p, x.p, g.am(0), x.am(0)) + ops.gemm_tn(g.t, x.t, g.am(1), x.am(1))


def _dgrad(g: Dual, w, addend: Dual = None):
    return Dual(ops._dgrad(g.p, w, addend.p if addend else None, g_amax=g.am(0)),
                ops._dgrad(g.t, w, addend.t if addend else None, g_amax=g.am(1)))


# ---------------------------------------------------------------------------------------------
# layers
# ---------------------------------------------------------------------------------------------
def mlp_fwd(layer, x: Dual, tape, fwd=None):
    lin, ln = layer.layer[0], layer.layer[1]
    ent = fwd.get(lin.weight.data_ptr()) if fwd else None
    if ent is not None and ent[0] == "mlp" and ent[2].shape[0] == x.t.shape[0]:
        pre = Dual(ent[2].detach(), ops.project(x.t, lin.weight, None, a_amax=x.am(1)))  # the tangent projection only
        y, stats = _ln_fwd(pre, None, ln.weight, ln.bias, value_out=_taken_over(ent[3]))
    else:
        pre = Dual(*_project(x, lin.weight, lin.bias))
        y, stats = _ln_fwd(pre, None, ln.weight, ln.bias)
    tape.append(("mlp", layer, x, pre, stats))
    return y


def mlp_bwd(entry, g: Dual, grads: _Grads, need_input_grad=True):
    _, layer, x, pre, stats = entry
    lin, ln = layer.layer[0], layer.layer[1]
    gpre, red = _ln_bwd(g, pre, ln.weight, ln.bias, stats)
    grads.add(ln.bias, red[0])
    grads.add(ln.weight, red[1])
    grads.add(lin.weight, _wgrad(gpre, x))
    grads.add(lin.bias, ops.col_sum(gpre.p))
    return _dgrad(gpre, lin.weight) if need_input_grad else None


def conv_fwd(conv, graph: CSRGraph, x: Dual, y: Dual, need_y, tape, fwd=None):
    lib = _lib.load()
    n, H = x.p.shape
    m = y.p.shape[0]
    wcat, bcat = conv._fused_node_projection()
    res = conv.residual
    ent = fwd.get(conv.edge_gate.weight.data_ptr()) if fwd else None
    if (ent is not None and ent[0] == "conv" and tuple(ent[3].shape) == (n, 4 * H) and tuple(ent[4].shape) == (m, H)
            and (ent[9] is not None or not need_y)):
        # values from the force evaluation; here the tangents only: two tangent projections, the tangent half of the gate pass
        _, _x, _y, P_p, M_p, xpre_p, s0, hh, xo, yo, e_rows = ent
        P = Dual(P_p.detach(), ops.project(x.t, wcat, None, a_amax=x.am(1)))
        M = Dual(M_p.detach(), ops.project(y.t, conv.edge_gate.weight, None, a_amax=y.am(1)))
        xpre = Dual(xpre_p.detach(), _empty(n, H, like=x.p))
        s0t, hht = _empty(n, H, like=x.p), _empty(n, H, like=x.p)
        if need_y and e_rows is not None and tuple(e_rows.shape) == (m, 2) and lib.alignn_egc_ln_fused_supported(H, m):
            # the tangent of the edge LayerNorm inside the gate pass (csrc/convln.hip); row statistics from the evaluation
            yp, yp_amax = _taken_over(yo)
            yt = _empty(m, H, like=x.p)
            amax = _amax2(x.p) if _track(m) else None
            check(lib.alignn_egc_gate_dual_tan_ln(ptr(P.p), ptr(P.t), ptr(M.p), ptr(M.t), ptr(graph.seg_ptr), ptr(graph.seg_node),
                                                  ptr(graph.src), n, m, H, ptr(xpre.t), ptr(s0), ptr(hh), ptr(s0t), ptr(hht),
                                                  ptr(conv.bn_edges.weight), ptr(conv.bn_edges.bias), ptr(e_rows),
                                                  ptr(y.t) if res else None, ptr(yt), ptr(amax), stream()),
                  "egc_gate_dual_tan_ln")
            y_out, e_stats = Dual(yp, yt, amax, yp_amax), e_rows.detach()
            x_out, n_stats = _ln_fwd(xpre, x if res else None, conv.bn_nodes.weight, conv.bn_nodes.bias, value_out=_taken_over(xo))
            tape.append(("conv", conv, graph, x, y, P, M, xpre, (s0, hh, s0t, hht), n_stats, e_stats))
            return x_out, y_out
        check(lib.alignn_egc_gate_dual_fwd_tangent(ptr(P.p), ptr(P.t), ptr(M.p), ptr(M.t), ptr(graph.seg_ptr), ptr(graph.seg_node),
                                                   ptr(graph.src), n, m, H, ptr(xpre.t), ptr(s0), ptr(hh), ptr(s0t), ptr(hht),
                                                   stream()), "egc_gate_dual_fwd_tangent")
        x_out, n_stats = _ln_fwd(xpre, x if res else None, conv.bn_nodes.weight, conv.bn_nodes.bias, value_out=_taken_over(xo))
        y_out, e_stats = (None, None)
        if need_y:
            y_out, e_stats = _ln_fwd(M, y if res else None, conv.bn_edges.weight, conv.bn_edges.bias, value_out=_taken_over(yo))
        tape.append(("conv", conv, graph, x, y, P, M, xpre, (s0, hh, s0t, hht), n_stats, e_stats))
        return x_out, y_out
    P = Dual(*_project(x, wcat, bcat))
    M = Dual(*_project(y, conv.edge_gate.weight, conv.edge_gate.bias))
    xpre = Dual(_empty(n, H, like=x.p), _empty(n, H, like=x.p))
    s0, hh, s0t, hht = (_empty(n, H, like=x.p) for _ in range(4))
    check(lib.alignn_egc_gate_dual_fwd(ptr(P.p), ptr(P.t), ptr(M.p), ptr(M.t), ptr(graph.seg_ptr), ptr(graph.seg_node),
                                       ptr(graph.src), n, m, H, ptr(xpre.p), ptr(xpre.t), ptr(s0), ptr(hh), ptr(s0t),
                                       ptr(hht), stream()), "egc_gate_dual_fwd")
    x_out, n_stats = _ln_fwd(xpre, x if res else None, conv.bn_nodes.weight, conv.bn_nodes.bias)
    y_out, e_stats = (None, None)
    if need_y:
        y_out, e_stats = _ln_fwd(M, y if res else None, conv.bn_edges.weight, conv.bn_edges.bias)
    tape.append(("conv", conv, graph, x, y, P, M, xpre, (s0, hh, s0t, hht), n_stats, e_stats))
    return x_out, y_out


def conv_bwd(entry, gx: Dual, gy, grads: _Grads):
    """(gx, gy) = adjoints of the outputs (gy None: dead edge output) -> adjoints of the inputs"""
    lib = _lib.load()
    _, conv, graph, x, y, P, M, xpre, (s0, hh, s0t, hht), n_stats, e_stats = entry
    n, H = x.p.shape
    m = y.p.shape[0]
    if gx is None:  # (node output unused downstream: cannot happen inside ALIGNNAtomWise, kept for completeness)
        gx = Dual(torch.zeros_like(x.p), torch.zeros_like(x.p))
    GP = Dual(_empty(n, 4 * H, like=x.p), _empty(n, 4 * H, like=x.p), _amax2(x.p) if _track(n) else None)
    # node branch: LayerNorm/SiLU reverse straight into the Ux blocks
    q1, q0, q1t, q0t = (_empty(n, H, like=x.p) for _ in range(4))
    _gxpre, n_red = _ln_bwd(gx, xpre, conv.bn_nodes.weight, conv.bn_nodes.bias, n_stats, GP.p[:, 3 * H:], GP.t[:, 3 * H:],
                            amax=GP.amax, node=(s0, hh, s0t, hht, q1, q0, q1t, q0t))
    grads.add(conv.bn_nodes.bias, n_red[0])
    grads.add(conv.bn_nodes.weight, n_red[1])
    dense = (DENSE_LG_REVERSE and graph.grp_seg_ptr is not None and graph.dense_max_src > 0 and ops.FUSED_LG_BACKWARD
             and ops.DENSE_LG_BACKWARD)
    ln_inside = bool(gy is not None and dense and lib.alignn_egc_ln_fused_supported(H, m))  # (csrc/convln.hip)
    ln_dst = bool(gy is not None and not dense and lib.alignn_egc_ln_dst_supported(H))  # (the bond graph: same file)
    GL = None
    if gy is not None and not ln_inside and not ln_dst:
        GL, e_red = _ln_bwd(gy, M, conv.bn_edges.weight, conv.bn_edges.bias, e_stats)
        grads.add(conv.bn_edges.bias, e_red[0])
        grads.add(conv.bn_edges.weight, e_red[1])
    GM = Dual(_empty(m, H, like=x.p), _empty(m, H, like=x.p), _amax2(x.p) if _track(m) else None)
    if ln_inside:  # the LayerNorm reverse inside the dense gate reverse
        slabs = graph.grp_seg_ptr.numel() - 1
        gb_part = _empty(slabs, H, like=x.p)
        ln_part = _empty(slabs, 2, H, like=x.p)
        check(lib.alignn_egc_dual_bwd_lg_dense_ln(ptr(gy.p), ptr(gy.t), ptr(M.p), ptr(M.t), ptr(P.p), ptr(P.t), ptr(q1), ptr(q0),
                                                  ptr(q1t), ptr(q0t), ptr(conv.bn_edges.weight), ptr(conv.bn_edges.bias),
                                                  ptr(e_stats), m, ptr(graph.grp_seg_ptr), ptr(graph.grp_src_ptr), slabs,
                                                  ptr(graph.seg_ptr), ptr(graph.seg_node), H, ptr(GM.p), ptr(GM.t), ptr(GP.p),
                                                  ptr(GP.t), ptr(gb_part), ptr(ln_part), ptr(GM.amax), ptr(GP.amax), stream()),
              "egc_dual_bwd_lg_dense_ln")
        e_red = _empty(2, H, like=x.p)
        check(lib.alignn_bn_bwd_finalize(ptr(ln_part), slabs, H, ptr(e_red), stream()), "ln_dual_finalize")
        grads.add(conv.bn_edges.bias, e_red[0])
        grads.add(conv.bn_edges.weight, e_red[1])
    elif dense:  # line graph: destination- and source-ordered halves in one pass over the dense blocks (6 row passes, not 10)
        slabs = graph.grp_seg_ptr.numel() - 1
        gb_part = _empty(slabs, H, like=x.p)
        check(lib.alignn_egc_dual_bwd_lg_dense(ptr(GL.p) if GL else None, ptr(GL.t) if GL else None, ptr(M.p), ptr(M.t),
                                               ptr(P.p), ptr(P.t), ptr(q1), ptr(q0), ptr(q1t), ptr(q0t), m,
                                               ptr(graph.grp_seg_ptr), ptr(graph.grp_src_ptr), slabs, ptr(graph.seg_ptr),
                                               ptr(graph.seg_node), H, ptr(GM.p), ptr(GM.t), ptr(GP.p), ptr(GP.t),
                                               ptr(gb_part), ptr(GM.amax), ptr(GP.amax), stream()), "egc_dual_bwd_lg_dense")
    elif ln_dst:  # bond graph: the LayerNorm reverse inside the destination-ordered half
        slabs = lib.alignn_egc_ln_dst_slabs(n)
        gb_part = _empty(slabs, H, like=x.p)
        ln_part = _empty(slabs, 2, H, like=x.p)
        check(lib.alignn_egc_dual_bwd_dst_ln(ptr(gy.p), ptr(gy.t), ptr(M.p), ptr(M.t), ptr(P.p), ptr(P.t), ptr(q1), ptr(q0), ptr(q1t),
                                             ptr(q0t), ptr(conv.bn_edges.weight), ptr(conv.bn_edges.bias), ptr(e_stats),
                                             ptr(graph.seg_ptr), ptr(graph.seg_node), ptr(graph.src), n, H, ptr(GM.p), ptr(GM.t),
                                             ptr(GP.p), ptr(GP.t), ptr(gb_part), ptr(ln_part), ptr(GM.amax), ptr(GP.amax), stream()),
              "egc_dual_bwd_dst_ln")
        e_red = _empty(2, H, like=x.p)
        check(lib.alignn_bn_bwd_finalize(ptr(ln_part), slabs, H, ptr(e_red), stream()), "ln_dual_finalize")
        grads.add(conv.bn_edges.bias, e_red[0])
        grads.add(conv.bn_edges.weight, e_red[1])
        check(lib.alignn_egc_dual_bwd_src(ptr(GM.p), ptr(GM.t), ptr(M.p), ptr(M.t), ptr(q1), ptr(q1t), ptr(graph.out_ptr),
                                          ptr(graph.out_slot), ptr(graph.dst), n, H, ptr(GP.p), ptr(GP.t), ptr(GP.amax),
                                          stream()), "egc_dual_bwd_src")
    else:
        slabs = lib.alignn_dual_slabs(n)
        gb_part = _empty(slabs, H, like=x.p)
    if not dense and not ln_dst:
        check(lib.alignn_egc_dual_bwd_dst(ptr(GL.p) if GL else None, ptr(GL.t) if GL else None, ptr(M.p), ptr(M.t), ptr(P.p),
                                          ptr(P.t), ptr(q1), ptr(q0), ptr(q1t), ptr(q0t), ptr(graph.seg_ptr),
                                          ptr(graph.seg_node), ptr(graph.src), n, H, ptr(GM.p), ptr(GM.t), ptr(GP.p),
                                          ptr(GP.t), ptr(gb_part), ptr(GM.amax), ptr(GP.amax), stream()), "egc_dual_bwd_dst")
        check(lib.alignn_egc_dual_bwd_src(ptr(GM.p), ptr(GM.t), ptr(M.p), ptr(M.t), ptr(q1), ptr(q1t), ptr(graph.out_ptr),
                                          ptr(graph.out_slot), ptr(graph.dst), n, H, ptr(GP.p), ptr(GP.t), ptr(GP.amax),
                                          stream()), "egc_dual_bwd_src")
    wcat, _ = conv._fused_node_projection()
    res = conv.residual
    g_x = _dgrad(GP, wcat, gx if res else None)
    w_eg = conv.edge_gate.weight
    add = gy if (res and gy is not None) else None
    Kin = y.p.shape[1]
    g_weg = None
    if (Kin == H and tuple(w_eg.shape) == (H, Kin) and GM.am(1) is not None and y.am(1) is not None
            and ops.dgrad_wgrad_applies(m, H, Kin, GM.am(0), y.am(0)) and ops._x6_shape_ok(ops._Shape(m, H), Kin, H)):
        # value and tangent halves: input gradient + weight gradient in one pass over each half's g_m (csrc/gemm_dw.hip)
        wt = ops.split_f16x2(w_eg, True)
        gyp, dwp, _ = ops.gemm_dgrad_wgrad(GM.p, GM.am(0), y.p, y.am(0), wt, add.p if add is not None else None)
        gyt, dwt, _ = ops.gemm_dgrad_wgrad(GM.t, GM.am(1), y.t, y.am(1), wt, add.t if add is not None else None)
        g_y, g_weg = Dual(gyp, gyt), dwp + dwt
    else:
        g_y = _dgrad(GM, w_eg, add)
    g_wcat = _wgrad(GP, x)
    g_bcat = ops.col_sum(GP.p)
    for i, lin in enumerate((conv.src_gate, conv.dst_gate, conv.dst_update, conv.src_update)):
        grads.add(lin.weight, g_wcat[i * H:(i + 1) * H])
        grads.add(lin.bias, g_bcat[i * H:(i + 1) * H])
    grads.add(conv.edge_gate.weight, g_weg if g_weg is not None else _wgrad(GM, y))
    g_beg = _empty(H, like=x.p)
    check(lib.alignn_slab_sum(ptr(gb_part), slabs, H, ptr(g_beg), stream()), "slab_sum")
    grads.add(conv.edge_gate.bias, g_beg)
    return g_x, g_y


# ---------------------------------------------------------------------------------------------
# geometry features and their tangents (a few element-wise operations on [E,3] / [T,3] / [T,bins] tensors)
# ---------------------------------------------------------------------------------------------
def _rbf_dual(d, dt, mod):
    """(RBF expansion of d - the kernel the forward used -, its tangent along dt)"""
    lib = _lib.load()
    d = d.contiguous()
    val = ops._rbf_raw(d, mod.centers, mod.gamma)
    tan = _empty(d.numel(), mod.centers.numel(), like=d)
    check(lib.alignn_rbf_tangent(ptr(d), ptr(dt), ptr(mod.centers), float(mod.gamma), ptr(tan), d.numel(), mod.centers.numel(),
                                 stream()), "rbf_tangent")
    return Dual(val, tan)


def _cos_dual(r, rt, lg: CSRGraph):
    """compute_bond_cosines (alignn/graphs.py:847-864) and its derivative along rt; r1 = -r[e1], r2 = r[e2]"""
    lib = _lib.load()
    h = ops._bond_cosines_raw(r, lg.src, lg.dst)
    ht = _empty(lg.n_edges, like=r)
    check(lib.alignn_bond_cosine_tangent(ptr(r), ptr(rt), ptr(lg.src), ptr(lg.dst), ptr(ht), lg.n_edges, stream()), "bond_cosine_tangent")
    return h, ht


def supported(cfg) -> bool:
    """configurations the dual pass covers (everything else trains on the composed path of alignn_amd.ff)"""
    return (cfg.calculate_gradient and not cfg.include_pos_deriv and not cfg.use_cutoff_function
            and cfg.extra_features == 0 and cfg.output_features == 1 and not cfg.classification and cfg.link == "identity"
            and cfg.additional_output_features == 0 and not (cfg.atomwise_output_features > 0 and cfg.atomwise_weight != 0)
            and (cfg.stresswise_weight == 0 or cfg.batch_stress))


def dual_pass(model, b: GraphBatch, w, wmax, g_energy, c, fwd=None):
    """Parameter gradients of  sum_g g_energy[g] E_g + c_g (D_w E)_g, c_g = c (* atoms of g: energy_mult_natoms)
    -> {id(param): (param, grad)}.  ``w`` [E, 3] = dL/d(pair forces), ``wmax`` = max|w| (device scalar): the tangent direction is
    w / 2^floor(log2 wmax) - tangent activations then have the scale of the values - and the seeds carry the factor back."""
    lib = _lib.load()
    cfg = model.config
    tape, grads = [], _Grads()
    n_a, n_g = len(model.alignn_layers), len(model.gcn_layers)
    r = b.r.contiguous()
    E = r.shape[0]
    d = ops._bond_length_raw(r)
    rt, dt = _empty(E, 3, like=r), _empty(E, like=r)
    check(lib.alignn_ff_tangent_geometry(ptr(r), ptr(w), ptr(wmax), ptr(d), ptr(rt), ptr(dt), E, stream()), "ff_tangent_geometry")
    af = b.atom_features
    x = mlp_fwd(model.atom_embedding, Dual(af, torch.zeros_like(af)), tape, fwd)
    y = mlp_fwd(model.edge_embedding[2], mlp_fwd(model.edge_embedding[1], _rbf_dual(d, dt, model.edge_embedding[0]), tape, fwd), tape, fwd)
    if n_a > 0:
        if cfg.lg_on_fly:
            h, ht = _cos_dual(r, rt, b.lg)
        else:
            h, ht = b.h, torch.zeros_like(b.h)
        z = mlp_fwd(model.angle_embedding[2], mlp_fwd(model.angle_embedding[1], _rbf_dual(h, ht, model.angle_embedding[0]), tape, fwd), tape, fwd)
    for i, layer in enumerate(model.alignn_layers):
        x, m = conv_fwd(layer.node_update, b.g, x, y, True, tape, fwd)
        y, z = conv_fwd(layer.edge_update, b.lg, m, z, i + 1 < n_a, tape, fwd)
    for i, layer in enumerate(model.gcn_layers):
        x, y = conv_fwd(layer, b.g, x, y, i + 1 < n_g, tape, fwd)
    # readout: E_g = fc(mean_i x_i)  (alignn_atomwise.py:464-466); reverse with the two seeds
    hp = ops.AvgPoolFn.apply(x.p, b.graph_ptr)
    hpt = ops.AvgPoolFn.apply(x.t, b.graph_ptr)
    fc = model.fc
    B, H = b.batch_size, hp.shape[1]
    emn = int(cfg.energy_mult_natoms)
    gW, gb = _empty(1, H, like=r), _empty(1, like=r)
    check(lib.alignn_ff_fc_grad(ptr(g_energy), float(c), emn, ptr(wmax), ptr(b.graph_ptr), ptr(hp), ptr(hpt), ptr(gW), ptr(gb), B, H,
                                stream()), "ff_fc_grad")
    grads.add(fc.weight, gW)
    grads.add(fc.bias, gb)
    N = b.g.n_nodes
    gx = Dual(_empty(N, H, like=r), _empty(N, H, like=r))
    check(lib.alignn_ff_readout_seed(ptr(g_energy), float(c), emn, ptr(wmax), ptr(b.graph_ptr), ptr(fc.weight), ptr(gx.p), ptr(gx.t), B,
                                     N, H, stream()), "ff_readout_seed")
    gy = None
    gz = None
    # reverse over the tape
    k = len(tape) - 1
    for i in reversed(range(n_g)):
        gx, gy_in = conv_bwd(tape[k], gx, gy, grads)
        gy = gy_in
        k -= 1
    for i in reversed(range(n_a)):
        gy_e, gz = conv_bwd(tape[k], gy, gz, grads)  # edge_update: nodes = bonds (m), edges = triplets
        k -= 1
        gx, gy = conv_bwd(tape[k], gx, gy_e, grads)  # node_update: outputs (x, m)
        k -= 1
    if n_a > 0:
        gz1 = mlp_bwd(tape[k], gz, grads)
        mlp_bwd(tape[k - 1], gz1, grads, need_input_grad=False)
        k -= 2
    gy1 = mlp_bwd(tape[k], gy, grads)
    mlp_bwd(tape[k - 1], gy1, grads, need_input_grad=False)
    mlp_bwd(tape[k - 2], gx, grads, need_input_grad=False)
    return grads


class ForcesFn(torch.autograd.Function):
    """(E, forces, stresses) of ALIGNNAtomWise as one autograd node; see the module docstring."""

    @staticmethod
    def forward(ctx, model, batch, *params):
        prev, ops.FORWARD_TAPE = ops.FORWARD_TAPE, ({} if REUSE_FORWARD else None)
        try:
            with torch.enable_grad(), ops.no_param_grad():
                res = model._forward_fused(batch, True)
        finally:
            ctx.fwd_tape, ops.FORWARD_TAPE = ops.FORWARD_TAPE, prev
        ctx.model, ctx.batch = model, batch
        ctx.params = params
        ctx.has_stress = torch.is_tensor(res["stresses"]) and res["stresses"].dim() == 3
        out, forces, stress = res["out"].detach(), res["grad"].detach(), res["stresses"].detach()
        return out, forces, stress

    @staticmethod
    def backward(ctx, g_out, g_forces, g_stress):
        model, b, cfg = ctx.model, ctx.batch, ctx.model.config
        gg = b.g
        lib = _lib.load()
        with torch.no_grad():
            dev = b.r.device
            E = gg.n_edges
            # w_e = dL/d(pair force of bond e): gF[dst] - gF[src] (the in-minus-out force reduction reversed) + k_g gS_g^T r_e
            # (stress_g = k_g sum_e r_e (x) pf_e), and max|w| - one kernel (csrc/ff.hip)
            gF = None if g_forces is None else g_forces.reshape(gg.n_nodes, 3).to(torch.float32).contiguous()
            gS = g_stress.reshape(-1, 3, 3).to(torch.float32).contiguous() if (ctx.has_stress and g_stress is not None) else None
            r = b.r.contiguous()
            w = _empty(E, 3, like=r)
            wmax = torch.zeros(1, dtype=torch.float32, device=dev)
            check(lib.alignn_ff_pair_weights(ptr(gF), ptr(gS), ptr(r), ptr(gg.src), ptr(gg.dst), ptr(b.graph_ptr), ptr(gg.seg_ptr),
                                             ptr(b.cell_volumes()) if gS is not None else None, float(cfg.stress_multiplier) * (-160.21766208),
                                             int(cfg.add_reverse_forces), b.batch_size, E, ptr(w), ptr(wmax), stream()), "ff_pair_weights")
            c = float(cfg.grad_multiplier) * (gg.n_nodes if cfg.force_mult_natoms else 1)
            ge = g_out.reshape(-1).to(torch.float32).contiguous() if g_out is not None else None
            grads = dual_pass(model, b, w, wmax, ge, c, ctx.fwd_tape)
            ctx.fwd_tape = None  # (its tensors are the force evaluation's activations: let them go)
        return (None, None) + tuple(grads.get(p) for p in ctx.params)
