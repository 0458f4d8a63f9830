"""Torch-only stand-in for the handful of DGL calls the ALIGNN model files make.

TEST INFRASTRUCTURE ONLY.  This is *not* DGL and is never imported by the
product package (``alignn_amd``).  Its sole purpose is to let the reference's
own model files (``/root/reference/alignn/models/*.py``) be imported unmodified
inside the authoring container, where the real ``dgl`` wheel is absent, so that
golden vectors can be generated from the reference's first-party arithmetic
(see ``oracle/make_golden.py``).

Pinned upstream: ``dgl<=1.1.1`` (reference ``setup.py:24``) / ``dgl==2.1.0``
(reference ``environment.yml:237``).  Semantics restated from the public DGL
API documentation (edge ``e`` runs ``u_e -> v_e``; ``u`` = source, ``v`` =
destination):

* ``apply_edges(fn.u_add_v(a, b, o))``     : ``edata[o][e] = ndata[a][u_e] + ndata[b][v_e]``
* ``update_all(fn.u_mul_e(a, b, m), fn.sum(m, o))``
                                           : ``ndata[o][i] = sum_{e: v_e = i} ndata[a][u_e] * edata[b][e]``
                                             (zero for nodes without in-edges)
* ``update_all(fn.copy_e(a, m), fn.sum(m, o))``
                                           : ``ndata[o][i] = sum_{e: v_e = i} edata[a][e]``
* ``apply_edges(udf)``                     : ``udf`` sees ``.src[k] = ndata[k][u]``, ``.dst[k] = ndata[k][v]``, ``.data = edata``
* ``g.line_graph(shared=True)``            : node ``i`` of L(g) is edge ``i`` of g; ``e1 -> e2`` iff ``v_{e1} == u_{e2}`` and ``e1 != e2``
                                             (backtracking pairs included - DGL default)
* ``dgl.batch``                            : disjoint union, cumulative id offsets, keeps ``batch_num_nodes/edges``
* ``dgl.reverse``                          : swap ``u`` and ``v``
* ``dgl.nn.AvgPooling / SumPooling``       : per-graph mean / sum over nodes, by ``batch_num_nodes``

Call sites in the reference that rely on these: ``alignn/models/alignn.py:88,100,105-108,242,325``;
``alignn/models/alignn_atomwise.py:179,184-187,384-386,429-430,492,548-557``;
``alignn/graphs.py:588-589``; ``alignn/tests/test_force_reduction.py:155-205``.
"""

from __future__ import annotations

import torch

from . import function  # noqa: F401
from . import nn  # noqa: F401
from . import data  # noqa: F401

__version__ = "0.0-shim"


class _Frame(dict):
    """Feature dictionary (``g.ndata`` / ``g.edata``)."""


class _EdgeView:
    """What a user-defined edge function receives."""

    def __init__(self, g):
        u, v = g._u, g._v
        self.src = {k: t[u] for k, t in g.ndata.items()}
        self.dst = {k: t[v] for k, t in g.ndata.items()}
        self.data = g.edata


class DGLGraph:
    def __init__(self, u, v, num_nodes=None, batch_num_nodes=None, batch_num_edges=None):
        u = torch.as_tensor(u, dtype=torch.int64)
        v = torch.as_tensor(v, dtype=torch.int64)
        if num_nodes is None:
            num_nodes = int(max(u.max().item(), v.max().item())) + 1 if u.numel() else 0
        self._u, self._v = u, v
        self._n = int(num_nodes)
        self.ndata = _Frame()
        self.edata = _Frame()
        self._bnn = batch_num_nodes
        self._bne = batch_num_edges

    # ---- structure -----------------------------------------------------
    def edges(self):
        return self._u, self._v

    def num_nodes(self):
        return self._n

    number_of_nodes = num_nodes

    def num_edges(self):
        return int(self._u.numel())

    number_of_edges = num_edges

    def in_degrees(self):
        return torch.bincount(self._v, minlength=self._n)

    def batch_num_nodes(self):
        if self._bnn is None:
            return torch.tensor([self._n], dtype=torch.int64, device=self._u.device)
        return self._bnn

    def batch_num_edges(self):
        if self._bne is None:
            return torch.tensor([self.num_edges()], dtype=torch.int64, device=self._u.device)
        return self._bne

    @property
    def batch_size(self):
        return int(self.batch_num_nodes().numel())

    @property
    def device(self):
        return self._u.device

    def to(self, device, **_):
        g = DGLGraph(
            self._u.to(device),
            self._v.to(device),
            self._n,
            None if self._bnn is None else self._bnn.to(device),
            None if self._bne is None else self._bne.to(device),
        )
        g.ndata.update({k: t.to(device) for k, t in self.ndata.items()})
        g.edata.update({k: t.to(device) for k, t in self.edata.items()})
        return g

    def local_var(self):
        g = DGLGraph(self._u, self._v, self._n, self._bnn, self._bne)
        g.ndata.update(self.ndata)
        g.edata.update(self.edata)
        return g

    local_scope = None  # not used by the hot path

    # ---- message passing -------------------------------------------------
    def apply_edges(self, func):
        if isinstance(func, function._Binary):
            lhs = self._operand(func.lhs_target, func.lhs)
            rhs = self._operand(func.rhs_target, func.rhs)
            self.edata[func.out] = func.op(lhs, rhs)
        else:
            self.edata.update(func(_EdgeView(self)))

    def _operand(self, target, name):
        if target == "u":
            return self.ndata[name][self._u]
        if target == "v":
            return self.ndata[name][self._v]
        return self.edata[name]

    def update_all(self, message_func, reduce_func):
        if isinstance(message_func, function._Binary):
            msg = message_func.op(
                self._operand(message_func.lhs_target, message_func.lhs),
                self._operand(message_func.rhs_target, message_func.rhs),
            )
        elif isinstance(message_func, function._CopyE):
            msg = self.edata[message_func.name]
        elif isinstance(message_func, function._CopyU):
            msg = self.ndata[message_func.name][self._u]
        else:  # pragma: no cover
            raise NotImplementedError(type(message_func))
        if not isinstance(reduce_func, function._Sum):  # pragma: no cover
            raise NotImplementedError(type(reduce_func))
        out = torch.zeros((self._n,) + tuple(msg.shape[1:]), dtype=msg.dtype, device=msg.device)
        self.ndata[reduce_func.out] = out.index_add(0, self._v, msg)

    # ---- derived graphs --------------------------------------------------
    def line_graph(self, backtracking=True, shared=False):
        u, v = self._u, self._v
        m = self.num_edges()
        # e1 -> e2 iff v[e1] == u[e2]; group e1 by its destination node.
        order = torch.argsort(v, stable=True)
        counts = torch.bincount(v, minlength=self._n)
        ptr = torch.zeros(self._n + 1, dtype=torch.int64)
        ptr[1:] = torch.cumsum(counts, 0)
        deg_in_of_src = counts[u]  # number of e1 candidates for every e2
        e2 = torch.repeat_interleave(torch.arange(m), deg_in_of_src)
        start = ptr[u][e2]
        first = torch.cumsum(deg_in_of_src, 0) - deg_in_of_src
        within = torch.arange(e2.numel()) - first[e2]
        e1 = order[start + within]
        keep = e1 != e2
        if not backtracking:
            keep &= u[e1] != v[e2]
        lg = DGLGraph(e1[keep], e2[keep], m)
        if shared:
            lg.ndata.update(self.edata)
        return lg


def graph(data, num_nodes=None, **_):
    u, v = data
    return DGLGraph(u, v, num_nodes)


def batch(graphs):
    us, vs, bnn, bne = [], [], [], []
    off = 0
    for g in graphs:
        us.append(g._u + off)
        vs.append(g._v + off)
        bnn.append(g.num_nodes())
        bne.append(g.num_edges())
        off += g.num_nodes()
    out = DGLGraph(
        torch.cat(us), torch.cat(vs), off, torch.tensor(bnn, dtype=torch.int64), torch.tensor(bne, dtype=torch.int64)
    )
    for k in graphs[0].ndata:
        out.ndata[k] = torch.cat([g.ndata[k] for g in graphs], 0)
    for k in graphs[0].edata:
        out.edata[k] = torch.cat([g.edata[k] for g in graphs], 0)
    return out


def unbatch(g):
    outs = []
    n0 = e0 = 0
    for n, e in zip(g.batch_num_nodes().tolist(), g.batch_num_edges().tolist()):
        s = DGLGraph(g._u[e0 : e0 + e] - n0, g._v[e0 : e0 + e] - n0, n)
        for k, t in g.ndata.items():
            s.ndata[k] = t[n0 : n0 + n]
        for k, t in g.edata.items():
            s.edata[k] = t[e0 : e0 + e]
        outs.append(s)
        n0 += n
        e0 += e
    return outs


def reverse(g, copy_ndata=True, copy_edata=False):
    r = DGLGraph(g._v, g._u, g._n, g._bnn, g._bne)
    if copy_ndata:
        r.ndata.update(g.ndata)
    if copy_edata:
        r.edata.update(g.edata)
    return r


def radius_graph(x, r, self_loop=False, **_):
    """All ordered pairs (j -> i) with ||x_i - x_j|| <= r (non-periodic)."""
    d = torch.cdist(x.detach(), x.detach())
    mask = d <= r
    if not self_loop:
        mask.fill_diagonal_(False)
    dst, src = torch.nonzero(mask, as_tuple=True)
    return DGLGraph(src, dst, x.shape[0])
