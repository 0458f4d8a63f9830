"""CPU oracle: a functional torch restatement of the reference's ALIGNN hot path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this file; the product package
``alignn_amd`` never does (it fails loudly when its HIP library is missing).

What it restates (every function cites the reference file:line it follows):
the BatchNorm-flavoured ``alignn.models.alignn.ALIGNN`` forward, written with
explicit COO gathers / ``index_add`` in place of the DGL primitives, taking the
*reference's own* ``state_dict`` names so a checkpoint or a freshly initialised
reference module can be evaluated by both implementations.

Parity pinning: the reference has no golden vectors for this path (SURVEY.md
section 8(c)).  The oracle is pinned instead against the reference's *own model
files* executed in the authoring container on the DGL shim in ``oracle/shims``
(``oracle/make_golden.py`` -> ``tests/golden/*.npz``; checked by
``tests/test_oracle_golden.py``).  The DGL primitive semantics themselves are
restated from DGL's public documentation (dgl pinned ``<=1.1.1`` by the
reference's ``setup.py:24``) and cannot be cross-checked against real DGL here.

Runs in fp32 or fp64 (``dtype=``); autograd works through it, so it also yields
oracle gradients for every parameter.
"""

from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

EPS_GATE = 1e-6  # alignn/models/alignn.py:109
BN_EPS = 1e-5  # torch.nn.BatchNorm1d default, alignn.py:72,76,178
BN_MOMENTUM = 0.1


def rbf_expand(d: torch.Tensor, vmin: float, vmax: float, bins: int, centers: Optional[torch.Tensor] = None):
    """alignn/models/utils.py:11-44 (lengthscale=None branch): exp(-gamma (d - c_k)^2), gamma = 1/mean(diff(c))."""
    c = torch.linspace(vmin, vmax, bins, dtype=torch.float32) if centers is None else centers
    lengthscale = float((c[1:] - c[:-1]).to(torch.float32).mean()) if bins > 1 else 1.0
    # the reference computes np.diff(centers).mean() on the float32 buffer
    gamma = 1.0 / lengthscale
    return torch.exp(-gamma * (d.unsqueeze(1) - c.to(d.dtype)) ** 2), gamma


def _bn(x, p, prefix, training, stats):
    """BatchNorm1d: batch statistics over all rows in training (alignn.py:72,76,178)."""
    w, b = p[prefix + ".weight"], p[prefix + ".bias"]
    rm, rv = p[prefix + ".running_mean"], p[prefix + ".running_var"]
    if training:
        mean = x.mean(0)
        var = x.var(0, unbiased=False)
        if stats is not None:
            n = x.shape[0]
            stats[prefix] = (mean.detach().clone(), (var * (n / max(n - 1, 1))).detach().clone())
        return (x - mean) * torch.rsqrt(var + BN_EPS) * w + b
    return (x - rm) * torch.rsqrt(rv + BN_EPS) * w + b


LN_EPS = 1e-5


def _ln(x, p, prefix):
    """nn.LayerNorm over the feature dimension (alignn_atomwise.py:151,155; models/utils.py:285)."""
    return F.layer_norm(x, (x.shape[1],), p[prefix + ".weight"], p[prefix + ".bias"], LN_EPS)


def _norm(x, p, prefix, training, stats, norm):
    return _ln(x, p, prefix) if norm == "layer" else _bn(x, p, prefix, training, stats)


def _linear(x, p, prefix):
    return x @ p[prefix + ".weight"].t() + p[prefix + ".bias"]


def mlp_layer(x, p, prefix, training, stats, norm="batch"):
    """alignn.py:170-184: SiLU(BatchNorm1d(Linear(x))); LayerNorm twin: models/utils.py:277-292."""
    return F.silu(_norm(_linear(x, p, prefix + ".layer.0"), p, prefix + ".layer.1", training, stats, norm))


def edge_gated_conv(p, prefix, u, v, x, y, training=True, stats=None, record=None, residual=True, norm="batch"):
    """alignn.py:78-129 with the DGL calls spelled out.

    ``u_add_v``  (alignn.py:100)      : A[u] + Bd[v]
    ``u_mul_e`` + ``sum`` (:105-107)   : S1[i] = sum_{e: v_e = i} Bh[u_e] * sigma_e
    ``copy_e`` + ``sum``  (:108)       : S0[i] = sum_{e: v_e = i} sigma_e
    """
    n = x.shape[0]
    a = _linear(x, p, prefix + ".src_gate")
    bd = _linear(x, p, prefix + ".dst_gate")
    m = a[u] + bd[v] + _linear(y, p, prefix + ".edge_gate")
    sigma = torch.sigmoid(m)
    bh = _linear(x, p, prefix + ".dst_update")
    s1 = torch.zeros(n, x.shape[1], dtype=x.dtype).index_add(0, v, bh[u] * sigma)
    s0 = torch.zeros(n, x.shape[1], dtype=x.dtype).index_add(0, v, sigma)
    h = s1 / (s0 + EPS_GATE)
    xn = _linear(x, p, prefix + ".src_update") + h
    if record is not None:
        record[prefix + ".m_pre"] = m.detach()
        record[prefix + ".x_pre"] = xn.detach()
    xo = F.silu(_norm(xn, p, prefix + ".bn_nodes", training, stats, norm))
    yo = F.silu(_norm(m, p, prefix + ".bn_edges", training, stats, norm))
    if residual:
        xo = x + xo
        yo = y + yo
    return xo, yo


def alignn_forward(
    p: Dict[str, torch.Tensor],
    graph,
    alignn_layers: int = 4,
    gcn_layers: int = 4,
    training: bool = True,
    stats: Optional[dict] = None,
    record: Optional[dict] = None,
    link: str = "identity",
    classification: bool = False,
):
    """alignn.py:282-349 (extra_features == 0 path).

    ``graph`` is a ``RawGraph``-like object of torch tensors: ``u, v, r,
    atom_features, lg_u, lg_v, h, batch_num_nodes``.
    """
    dt = p["fc.weight"].dtype
    u, v = graph.u, graph.v
    e1, e2 = graph.lg_u, graph.lg_v
    if alignn_layers > 0:
        zc = p["angle_embedding.0.centers"]
        z, _ = rbf_expand(graph.h.to(dt), -1.0, 1.0, zc.numel(), zc)
        z = mlp_layer(z, p, "angle_embedding.1", training, stats)
        z = mlp_layer(z, p, "angle_embedding.2", training, stats)
    x = mlp_layer(graph.atom_features.to(dt), p, "atom_embedding", training, stats)
    bondlength = torch.norm(graph.r.to(dt), dim=1)
    yc = p["edge_embedding.0.centers"]
    y, _ = rbf_expand(bondlength, 0.0, 8.0, yc.numel(), yc)
    y = mlp_layer(y, p, "edge_embedding.1", training, stats)
    y = mlp_layer(y, p, "edge_embedding.2", training, stats)
    if record is not None:
        record["x0"], record["y0"] = x.detach(), y.detach()
        if alignn_layers > 0:
            record["z0"] = z.detach()
    for i in range(alignn_layers):
        # alignn.py:145-167: x,m = node_update(g,x,y); y,z = edge_update(lg,m,z)
        x, m = edge_gated_conv(p, f"alignn_layers.{i}.node_update", u, v, x, y, training, stats, record)
        y, z = edge_gated_conv(p, f"alignn_layers.{i}.edge_update", e1, e2, m, z, training, stats, record)
        if record is not None:
            record[f"alignn.{i}.x"], record[f"alignn.{i}.y"], record[f"alignn.{i}.z"] = x.detach(), y.detach(), z.detach()
    for i in range(gcn_layers):
        x, y = edge_gated_conv(p, f"gcn_layers.{i}", u, v, x, y, training, stats, record)
        if record is not None:
            record[f"gcn.{i}.x"], record[f"gcn.{i}.y"] = x.detach(), y.detach()
    # AvgPooling (alignn.py:325): per-graph mean over nodes
    bnn = graph.batch_num_nodes
    seg = torch.repeat_interleave(torch.arange(bnn.numel()), bnn)
    hsum = torch.zeros(bnn.numel(), x.shape[1], dtype=dt).index_add(0, seg, x)
    hg = hsum / bnn.to(dt).unsqueeze(1)
    out = _linear(hg, p, "fc")
    if link == "log":
        out = torch.exp(out)
    elif link == "logit":
        out = torch.sigmoid(out)
    if classification:
        out = F.log_softmax(out, dim=1)
    return torch.squeeze(out)


def bond_cosines(r, e1, e2):
    """alignn/graphs.py:847-864: cos = clamp(-r[e1].r[e2] / (|r[e1]||r[e2]|), -1, 1)."""
    r1, r2 = -r[e1], r[e2]
    c = torch.sum(r1 * r2, dim=1) / (torch.norm(r1, dim=1) * torch.norm(r2, dim=1))
    return torch.clamp(c, -1, 1)


def alignn_atomwise_forward(p, graph, alignn_layers=2, gcn_layers=2, lg_on_fly=True, record=None, link="identity",
                            calculate_gradient=False, volume=None, batch_num_edges=None, stress=False,
                            use_penalty=True, penalty_factor=0.1, penalty_threshold=1.0, grad_multiplier=-1,
                            stress_multiplier=1.0):
    """alignn_atomwise.py:364-660: LayerNorm everywhere, bond-angle cosines recomputed from r when lg_on_fly
    (:424-431).  Returns result["out"]; with ``calculate_gradient`` returns (out, forces, stresses) where
    pair forces = grad_multiplier * dE/dr taken with create_graph=True (:530-539), forces reduced over in- and
    out-edges (:547-565) and the per-crystal virial stress -160.21766208 r^T f / V (:615-638)."""
    dt = p["fc.weight"].dtype
    u, v, e1, e2 = graph.u, graph.v, graph.lg_u, graph.lg_v
    x = mlp_layer(graph.atom_features.to(dt), p, "atom_embedding", True, None, "layer")
    r = graph.r.to(dt)
    if calculate_gradient:
        r = r.detach().clone().requires_grad_(True)
    bondlength = torch.norm(r, dim=1)
    if alignn_layers > 0:
        h = bond_cosines(r, e1, e2) if lg_on_fly else graph.h.to(dt)
        zc = p["angle_embedding.0.centers"]
        z, _ = rbf_expand(h, -1.0, 1.0, zc.numel(), zc)
        z = mlp_layer(z, p, "angle_embedding.1", True, None, "layer")
        z = mlp_layer(z, p, "angle_embedding.2", True, None, "layer")
    yc = p["edge_embedding.0.centers"]
    y, _ = rbf_expand(bondlength, 0.0, 8.0, yc.numel(), yc)
    y = mlp_layer(y, p, "edge_embedding.1", True, None, "layer")
    y = mlp_layer(y, p, "edge_embedding.2", True, None, "layer")
    for i in range(alignn_layers):
        x, m = edge_gated_conv(p, f"alignn_layers.{i}.node_update", u, v, x, y, True, None, record, True, "layer")
        y, z = edge_gated_conv(p, f"alignn_layers.{i}.edge_update", e1, e2, m, z, True, None, record, True, "layer")
        if record is not None:
            record[f"alignn.{i}.x"], record[f"alignn.{i}.y"], record[f"alignn.{i}.z"] = x.detach(), y.detach(), z.detach()
    for i in range(gcn_layers):
        x, y = edge_gated_conv(p, f"gcn_layers.{i}", u, v, x, y, True, None, record, True, "layer")
        if record is not None:
            record[f"gcn.{i}.x"], record[f"gcn.{i}.y"] = x.detach(), y.detach()
    bnn = graph.batch_num_nodes
    seg = torch.repeat_interleave(torch.arange(bnn.numel()), bnn)
    hsum = torch.zeros(bnn.numel(), x.shape[1], dtype=dt).index_add(0, seg, x)
    out = torch.squeeze(_linear(hsum / bnn.to(dt).unsqueeze(1), p, "fc"))
    forces = stresses = None
    if calculate_gradient:
        en_out = out * bnn.to(dt)  # energy_mult_natoms (:494-497)
        if use_penalty:  # :498-510
            pen = torch.where(bondlength < penalty_threshold, penalty_factor * (penalty_threshold - bondlength),
                              torch.zeros_like(bondlength))
            en_out = en_out + torch.sum(pen)
        pf = grad_multiplier * torch.autograd.grad(en_out, r, grad_outputs=torch.ones_like(en_out), create_graph=True,
                                                   retain_graph=True)[0]
        n = x.shape[0]
        f_ji = torch.zeros(n, 3, dtype=dt).index_add(0, v, pf)
        f_ij = torch.zeros(n, 3, dtype=dt).index_add(0, u, pf)
        forces = f_ji - f_ij
        if stress:
            sts, e0 = [], 0
            for b_, ne in enumerate(batch_num_edges.tolist()):
                sts.append(-160.21766208 * (r[e0:e0 + ne].t() @ pf[e0:e0 + ne]) / volume[b_])
                e0 += ne
            stresses = stress_multiplier * torch.stack(sts)
    if link == "log":
        out = torch.exp(out)
    elif link == "logit":
        out = torch.sigmoid(out)
    if calculate_gradient:
        return out, forces, stresses
    return out


def running_stats_after_step(p, stats):
    """What BatchNorm1d leaves in its buffers after one training forward (momentum 0.1, unbiased var)."""
    out = {}
    for prefix, (mean, var_unbiased) in stats.items():
        out[prefix + ".running_mean"] = (1 - BN_MOMENTUM) * p[prefix + ".running_mean"] + BN_MOMENTUM * mean
        out[prefix + ".running_var"] = (1 - BN_MOMENTUM) * p[prefix + ".running_var"] + BN_MOMENTUM * var_unbiased
    return out


class TorchGraph:
    """RawGraph (numpy) -> torch CPU tensors, same field names."""

    def __init__(self, raw):
        self.u = torch.from_numpy(raw.u)
        self.v = torch.from_numpy(raw.v)
        self.r = torch.from_numpy(raw.r)
        self.atom_features = torch.from_numpy(raw.atom_features)
        self.lg_u = torch.from_numpy(raw.lg_u)
        self.lg_v = torch.from_numpy(raw.lg_v)
        self.h = torch.from_numpy(raw.h)
        self.batch_num_nodes = torch.from_numpy(raw.batch_num_nodes)


def init_state_dict(
    alignn_layers=4,
    gcn_layers=4,
    atom_input_features=92,
    edge_input_features=80,
    triplet_input_features=40,
    embedding_features=64,
    hidden_features=256,
    output_features=1,
    seed=0,
    dtype=torch.float32,
):
    """A state_dict with the reference's key names/shapes and torch-default-style init.

    Used when neither the reference nor the product module is at hand (CPU baseline timing).
    """
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def lin(prefix, fin, fout):
        bound = 1.0 / math.sqrt(fin)
        sd[prefix + ".weight"] = (torch.rand(fout, fin, generator=g) * 2 - 1) * bound
        sd[prefix + ".bias"] = (torch.rand(fout, generator=g) * 2 - 1) * bound

    def bn(prefix, f):
        sd[prefix + ".weight"] = torch.ones(f)
        sd[prefix + ".bias"] = torch.zeros(f)
        sd[prefix + ".running_mean"] = torch.zeros(f)
        sd[prefix + ".running_var"] = torch.ones(f)
        sd[prefix + ".num_batches_tracked"] = torch.tensor(0)

    def mlp(prefix, fin, fout):
        lin(prefix + ".layer.0", fin, fout)
        bn(prefix + ".layer.1", fout)

    def conv(prefix, f):
        for nm in ("src_gate", "dst_gate", "edge_gate", "src_update", "dst_update"):
            lin(f"{prefix}.{nm}", f, f)
        bn(prefix + ".bn_edges", f)
        bn(prefix + ".bn_nodes", f)

    mlp("atom_embedding", atom_input_features, hidden_features)
    sd["edge_embedding.0.centers"] = torch.linspace(0, 8.0, edge_input_features)
    mlp("edge_embedding.1", edge_input_features, embedding_features)
    mlp("edge_embedding.2", embedding_features, hidden_features)
    sd["angle_embedding.0.centers"] = torch.linspace(-1, 1.0, triplet_input_features)
    mlp("angle_embedding.1", triplet_input_features, embedding_features)
    mlp("angle_embedding.2", embedding_features, hidden_features)
    for i in range(alignn_layers):
        conv(f"alignn_layers.{i}.node_update", hidden_features)
        conv(f"alignn_layers.{i}.edge_update", hidden_features)
    for i in range(gcn_layers):
        conv(f"gcn_layers.{i}", hidden_features)
    lin("fc", hidden_features, output_features)
    return {k: (t.to(dtype) if t.is_floating_point() else t) for k, t in sd.items()}


def as_params(state_dict, dtype=torch.float32, requires_grad=True):
    """Detach-copy a state_dict into leaf tensors (parameters get requires_grad)."""
    out = {}
    for k, t in state_dict.items():
        t = t.detach().clone()
        if t.is_floating_point():
            t = t.to(dtype)
            is_buffer = k.endswith(("running_mean", "running_var", "centers"))
            t.requires_grad_(requires_grad and not is_buffer)
        out[k] = t
    return out


# ---------------------------------------------------------------------------------------------
# helpers shared by oracle/make_golden_full.py (writer) and tests/test_gpu_full_size.py (reader)
# ---------------------------------------------------------------------------------------------
def perturbed_norm_state_dict(sd, seed=1, scale=0.1):
    """Non-trivial norm affines (weight 1 + 0.1 N(0,1), bias 0.1 N(0,1)) on top of ``init_state_dict``: with the
    default gamma = 1 / beta = 0 a wrong gamma/beta gradient path could go unnoticed."""
    g = torch.Generator().manual_seed(seed)
    out = dict(sd)
    for k in sorted(sd):
        if (".bn_" in k or ".layer.1." in k) and k.endswith((".weight", ".bias")):
            out[k] = sd[k] + scale * torch.randn(sd[k].shape, generator=g)
    return out


def full_size_sample(t, k=512, row_map=None):
    """[k strided elements of the row-major tensor, mean, mean|.|, l2 norm, max|.|] in float64.  ``row_map`` (int64
    [rows]): the tensor is stored with row i of the REFERENCE order at row ``row_map[i]`` (canonical slot order of
    alignn_amd) - the strided sample is then taken at the reference's positions without permuting the tensor."""
    t = t.detach()
    if t.dim() == 0:
        t = t.reshape(1)
    n = t.numel()
    idx = torch.linspace(0, n - 1, min(k, n), dtype=torch.float64).long().to(t.device)
    if row_map is None or t.dim() < 2:
        vals = t.reshape(-1)[idx]
    else:
        w = t.shape[1]
        vals = t[row_map.to(t.device)[idx // w], idx % w]
    f = t.reshape(-1).double()
    mom = torch.stack([f.mean(), f.abs().mean(), f.norm(), f.abs().max()])
    return torch.cat([vals.double().reshape(-1), mom]).cpu().numpy()


def input_signature(raw):
    """A few exact integers + float sums of a RawGraph: pins that make_batch regenerated the golden's inputs."""
    import numpy as np

    def h(a):
        a = np.asarray(a).astype(np.uint64)
        w = (np.arange(a.shape[0], dtype=np.uint64) % np.uint64(1000003)) + np.uint64(1)
        return float(int(np.sum((a + np.uint64(1)) * w, dtype=np.uint64)) % (1 << 52))

    return np.array([raw.num_nodes, raw.num_edges, raw.num_triplets, h(raw.u), h(raw.v), h(raw.lg_u), h(raw.lg_v),
                     float(np.abs(raw.r.astype(np.float64)).sum()), float(raw.h.astype(np.float64).sum())], dtype=np.float64)
