"""Plain-torch execution of the ALIGNN layers for dtypes the HIP kernels do not cover.

``alignn/train.py:89-95`` sets torch's global default dtype from the config, and the reference's own numerical test
runs ``EdgeGatedGraphConv`` in float64 (``alignn/tests/test_force_reduction.py:11,200-268``).  The HIP library computes in
float32 only ("only fp32 needs kernels - fall back to the torch path otherwise"), so a module whose
parameters / inputs are float64, bfloat16 or float16 is evaluated here: the module's own ``nn.Linear`` / norm children
(which otherwise only hold parameters) are called as torch modules and the eight DGL primitives become
``index_select`` / ``index_add`` on the canonical slot order.  Differentiable to any order by torch's autograd, any
device.

This is NOT a fallback for float32: float32 tensors always take the HIP kernels and raise if they cannot
(``_lib.require_f32``) - the measured path and the parity claims are the kernels'.  ``ops.REGISTRY_STATS`` is not
touched here; the first use warns once.
"""
from __future__ import annotations

import warnings

import torch
import torch.nn.functional as F

_WARNED = [False]


def wanted(*tensors) -> bool:
    """True if any of the given tensors is a floating tensor that is not float32 (-> torch path)."""
    hit = any(t is not None and t.is_floating_point() and t.dtype != torch.float32 for t in tensors)
    if hit and not _WARNED[0]:
        _WARNED[0] = True
        dt = next(t.dtype for t in tensors if t is not None and t.is_floating_point() and t.dtype != torch.float32)
        warnings.warn(f"alignn_amd: {dt} tensors run on plain torch operations (alignn_amd/torch_path.py), not on the HIP "
                      "kernels, which compute in float32 only", RuntimeWarning, stacklevel=3)
    return hit


def rbf(distance, centers, gamma):
    """exp(-gamma (d - c_k)^2)  (alignn/models/utils.py:40-44)."""
    return torch.exp(-gamma * (distance.unsqueeze(1) - centers.to(distance.dtype)) ** 2)


def mlp_layer(mod, x):
    """Linear + norm + SiLU: the module's own Sequential (alignn.py:170-184 / utils.py:277-292)."""
    return mod.layer(x)


def edge_gated_conv(mod, src, dst, n_nodes, x, y, need_y=True):
    """``EdgeGatedGraphConv.forward`` (alignn/models/alignn.py:78-129) on COO indices ``src -> dst`` (int tensors, one entry
    per row of ``y``): u_add_v, sigmoid gate, the two destination sums, h = S1 / (S0 + 1e-6), norm + SiLU + residual."""
    u, v = src.long(), dst.long()
    m = mod.src_gate(x).index_select(0, u) + mod.dst_gate(x).index_select(0, v) + mod.edge_gate(y)
    sigma = torch.sigmoid(m)
    bh = mod.dst_update(x).index_select(0, u)
    zeros = x.new_zeros(n_nodes, m.shape[1])
    s1 = zeros.index_add(0, v, sigma * bh)
    s0 = zeros.index_add(0, v, sigma)
    h = s1 / (s0 + 1e-6)
    x_new = F.silu(mod.bn_nodes(mod.src_update(x) + h))
    y_new = F.silu(mod.bn_edges(m)) if (need_y or mod.training) else None  # (train mode: running statistics side effect)
    if mod.residual:
        x_new = x + x_new
        if y_new is not None:
            y_new = y + y_new
    return x_new, (y_new if need_y else None)


def segment_mean(x, graph_ptr):
    """dgl.nn.AvgPooling (alignn.py:325): per-crystal mean over atoms; ``graph_ptr`` int [B+1]."""
    counts = (graph_ptr[1:] - graph_ptr[:-1]).long()
    B = counts.numel()
    owner = torch.repeat_interleave(torch.arange(B, device=x.device), counts)
    out = x.new_zeros(B, x.shape[1]).index_add(0, owner, x)
    return out / counts.clamp_min(1).to(x.dtype).unsqueeze(1)


def alignn_forward(model, b):
    """``ALIGNN.forward`` (alignn/models/alignn.py:282-349) on a canonical ``GraphBatch``."""
    dt = model.fc.weight.dtype
    cfg = model.config
    if len(model.alignn_layers) > 0:
        z = model.angle_embedding(b.h.to(dt))
    x = model.atom_embedding(b.atom_features.to(dt))
    y = model.edge_embedding(torch.norm(b.r.to(dt), dim=1))
    for layer in model.alignn_layers:
        x, m = layer.node_update(b.g, x, y)
        y, z = layer.edge_update(b.lg, m, z)
    for layer in model.gcn_layers:
        x, y = layer(b.g, x, y)
    h = segment_mean(x, b.graph_ptr)
    if cfg.extra_features != 0:
        feats = model.extra_feature_embedding(b.extra_features.to(dt))
        out = model.fc3(model.fc2(model.fc1(torch.cat((h, segment_mean(feats, b.graph_ptr)), 1))))
    else:
        out = model.fc(h)
    if model.link:
        out = model.link(out)
    if model.classification:
        out = model.softmax(out)
    return torch.squeeze(out)
