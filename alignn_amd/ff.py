"""Twice-differentiable composition of the conv stack for the ALIGNN-FF head.

``ALIGNNAtomWise`` with ``calculate_gradient=True`` takes ``pair_forces = -dE/dr`` with
``autograd.grad(..., create_graph=True)`` INSIDE the forward and the training loss then differentiates through
that gradient (alignn/models/alignn_atomwise.py:512-565, alignn/train.py:387).  The fused conv kernels have a
hand-written first derivative only, so this path is composed from primitives that are closed under
differentiation:

* ``gather`` / ``segment_sum`` over a :class:`Relation` - one is the other's adjoint (HIP: ``alignn_gather_rows``,
  ``alignn_segment_sum``; they replace DGL's ``u_add_v`` / ``u_mul_e+sum`` / ``copy_e+sum`` and their autograd);
* ``matmul_nt / matmul_nn / matmul_tn`` - the three MFMA GEMMs, each other's derivatives;
* element-wise math and LayerNorm through torch's own twice-differentiable ops (sigmoid, silu, layer_norm, exp,
  norm, clamp) - the only place on any path where torch kernels do arithmetic; they are HBM-bound one-liners and
  this path is not the throughput benchmark.

Forward values equal the fused kernels' (same formulas); ``tests/test_gpu_model.py`` checks energies, forces,
stresses and all second-order parameter gradients against the reference's own class.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn.functional as F

from . import _lib, ops
from ._lib import check, ptr, require_f32, stream
from .graph import CSRGraph


@dataclass
class Relation:
    """Rows k = 0..m-1 each belong to one group: ``idx[k]``.  ``ptr/slot/node`` list the rows of every group
    (segment s = rows ``slot[ptr[s]:ptr[s+1]]`` (or that range itself) and belongs to group ``node[s]`` (or s))."""

    idx: torch.Tensor  # int32 [m]
    ptr: torch.Tensor  # int32 [n+1]
    slot: Optional[torch.Tensor]  # int32 [m] or None (contiguous)
    node: Optional[torch.Tensor]  # int32 [n] or None (identity)
    n: int


def by_dst(g: CSRGraph) -> Relation:
    return Relation(g.dst, g.seg_ptr, None, g.seg_node, g.n_nodes)


def by_src(g: CSRGraph) -> Relation:
    return Relation(g.src, g.out_ptr, g.out_slot, None, g.n_nodes)


def by_graph(graph_ptr: torch.Tensor) -> Relation:
    counts = (graph_ptr[1:] - graph_ptr[:-1]).to(torch.int64)
    idx = torch.repeat_interleave(torch.arange(counts.numel(), device=graph_ptr.device, dtype=torch.int32), counts)
    return Relation(idx, graph_ptr, None, None, int(counts.numel()))


def _gather_raw(x, rel: Relation):
    lib = _lib.load()
    require_f32(x)
    x = x.contiguous()
    Fw = x.shape[1]
    out = torch.empty(rel.idx.numel(), Fw, dtype=torch.float32, device=x.device)
    check(lib.alignn_gather_rows(ptr(x), ptr(rel.idx), ptr(out), rel.idx.numel(), Fw, stream()), "gather_rows")
    return out


def _segsum_raw(v, rel: Relation):
    lib = _lib.load()
    require_f32(v)
    v = v.contiguous()
    Fw = v.shape[1]
    out = torch.empty(rel.n, Fw, dtype=torch.float32, device=v.device)
    check(
        lib.alignn_segment_sum(ptr(v), v.stride(0), ptr(rel.ptr), ptr(rel.slot), ptr(rel.node), ptr(out), out.stride(0),
                               rel.n, Fw, stream()),
        "segment_sum",
    )
    return out


class GatherFn(torch.autograd.Function):
    """y[k] = x[idx[k]]; adjoint: segment_sum."""

    @staticmethod
    def forward(ctx, x, rel):
        ctx.rel = rel
        return _gather_raw(x, rel)

    @staticmethod
    def backward(ctx, g):
        return SegSumFn.apply(g, ctx.rel), None


class SegSumFn(torch.autograd.Function):
    """out[j] = sum_{k: idx[k] = j} v[k]; adjoint: gather."""

    @staticmethod
    def forward(ctx, v, rel):
        ctx.rel = rel
        return _segsum_raw(v, rel)

    @staticmethod
    def backward(ctx, g):
        return GatherFn.apply(g, ctx.rel), None


# float64 / 16-bit tensors (alignn/train.py:89-95 sets torch's default dtype from the config) take plain torch operations
# with the same meaning - index_select / index_add / matmul, twice differentiable as they are - instead of the float32
# kernels; like alignn_amd/torch_path.py this is NOT a fallback for float32.
def _f32(t):
    return t.dtype == torch.float32


def gather(x, rel):
    if not _f32(x):
        return x.index_select(0, rel.idx.long())
    return GatherFn.apply(x, rel)


def segment_sum(v, rel):
    if not _f32(v):
        return v.new_zeros((rel.n,) + tuple(v.shape[1:])).index_add(0, rel.idx.long(), v)
    return SegSumFn.apply(v, rel)


class MatmulNT(torch.autograd.Function):
    """a[M,K] @ w[N,K]^T."""

    @staticmethod
    def forward(ctx, a, w):
        ctx.save_for_backward(a, w)
        return ops.project(a.contiguous(), w.contiguous())

    @staticmethod
    def backward(ctx, g):
        a, w = ctx.saved_tensors
        return MatmulNN.apply(g, w), MatmulTN.apply(g, a)


class MatmulNN(torch.autograd.Function):
    """g[M,N] @ w[N,K]."""

    @staticmethod
    def forward(ctx, g, w):
        ctx.save_for_backward(g, w)
        return ops.project(g.contiguous(), w.contiguous(), transpose_w=True)

    @staticmethod
    def backward(ctx, go):
        g, w = ctx.saved_tensors
        return MatmulNT.apply(go, w), MatmulTN.apply(g, go)


class MatmulTN(torch.autograd.Function):
    """g[M,N]^T @ a[M,K]."""

    @staticmethod
    def forward(ctx, g, a):
        ctx.save_for_backward(g, a)
        return ops.gemm_tn(g.contiguous(), a.contiguous())

    @staticmethod
    def backward(ctx, go):
        g, a = ctx.saved_tensors
        return MatmulNT.apply(a, go), MatmulNN.apply(g, go)


def linear(x, lin):
    if not _f32(x):
        return x @ lin.weight.t() + lin.bias
    return MatmulNT.apply(x, lin.weight) + lin.bias


def mlp_layer(x, layer):
    """MLPLayer, LayerNorm flavour (alignn/models/utils.py:277-292)."""
    lin, ln = layer.layer[0], layer.layer[1]
    return F.silu(F.layer_norm(linear(x, lin), (lin.weight.shape[0],), ln.weight, ln.bias, ln.eps))


def rbf(d, mod):
    """RBFExpansion.forward (alignn/models/utils.py:40-44) with gradient w.r.t. the distances."""
    return torch.exp(-mod.gamma * (d.unsqueeze(1) - mod.centers.to(d.dtype)) ** 2)


def bond_cosines(r, lg: CSRGraph):
    """compute_bond_cosines (alignn/graphs.py:847-864) with gradient w.r.t. r."""
    r1 = -gather(r, by_src(lg))
    r2 = gather(r, by_dst(lg))
    c = torch.sum(r1 * r2, dim=1) / (torch.norm(r1, dim=1) * torch.norm(r2, dim=1))
    return torch.clamp(c, -1, 1)


def edge_gated_conv(g: CSRGraph, x, y, mod):
    """EdgeGatedGraphConv.forward, LayerNorm flavour (alignn_atomwise.py:157-208), features in canonical order."""
    rs, rd = by_src(g), by_dst(g)
    H = x.shape[1]
    m = gather(linear(x, mod.src_gate), rs) + gather(linear(x, mod.dst_gate), rd) + linear(y, mod.edge_gate)
    sigma = torch.sigmoid(m)
    bh = linear(x, mod.dst_update)
    s1 = segment_sum(sigma * gather(bh, rs), rd)
    s0 = segment_sum(sigma, rd)
    xn = linear(x, mod.src_update) + s1 / (s0 + 1e-6)
    xo = F.silu(F.layer_norm(xn, (H,), mod.bn_nodes.weight, mod.bn_nodes.bias, mod.bn_nodes.eps))
    yo = F.silu(F.layer_norm(m, (H,), mod.bn_edges.weight, mod.bn_edges.bias, mod.bn_edges.eps))
    if mod.residual:
        xo = x + xo
        yo = y + yo
    return xo, yo


def pair_force_reduce(pf, g: CSRGraph, add_reverse: bool = True):
    """forces_i = sum_{e: dst e = i} pf_e  -  sum_{e: src e = i} pf_e  (alignn_atomwise.py:547-565)."""
    f = segment_sum(pf, by_dst(g))
    if add_reverse:
        f = f - segment_sum(pf, by_src(g))
    return f
