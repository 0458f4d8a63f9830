"""MI355X-native drop-in for ``alignn.models.alignn`` (BatchNorm flavour).

Same classes, constructor arguments, attribute names, ``forward`` signatures and ``state_dict``
keys as the reference (``/root/reference/alignn/models/alignn.py``: ``ALIGNNConfig`` :19-45,
``EdgeGatedGraphConv`` :48-129, ``ALIGNNConv`` :132-167, ``MLPLayer`` :170-184, ``ALIGNN``
:187-349; ``RBFExpansion`` ``alignn/models/utils.py:11-44``), so a reference checkpoint loads with
``load_state_dict`` and an instance can be handed to ``alignn.train.train_dgl(model=...)``.

The ``nn.Linear`` / ``nn.BatchNorm1d`` children are kept purely as parameter/buffer containers
(that is what fixes the key names); their ``forward`` is never called.  All arithmetic goes
through ``alignn_amd.ops`` -> ``libalignn_hip.so``.
"""

from __future__ import annotations

from typing import Literal, Optional, Sequence, Union

import numpy as np
import torch
from torch import nn

try:  # the reference derives its config from pydantic-settings; fall back to plain pydantic
    from pydantic_settings import BaseSettings as _Base

    _CONFIG = {"extra": "forbid", "env_prefix": "jv_model", "protected_namespaces": ()}
except Exception:  # pragma: no cover - depends on the image
    from pydantic import BaseModel as _Base

    _CONFIG = {"extra": "forbid", "protected_namespaces": ()}

from . import _lib, cmodel, ops, torch_path
from .graph import CSRGraph, GraphBatch, build_csr, cached_dgl_batch


class ALIGNNConfig(_Base):
    """Hyperparameter schema; field-for-field the reference's (alignn/models/alignn.py:22-40)."""

    name: Literal["alignn"]
    alignn_layers: int = 4
    gcn_layers: int = 4
    atom_input_features: int = 92
    edge_input_features: int = 80
    triplet_input_features: int = 40
    embedding_features: int = 64
    hidden_features: int = 256
    output_features: int = 1
    link: Literal["identity", "log", "logit"] = "identity"
    zero_inflated: bool = False
    classification: bool = False
    num_classes: int = 2
    extra_features: int = 0

    model_config = _CONFIG


class RBFExpansion(nn.Module):
    """exp(-gamma (d - c_k)^2); ``centers`` is a registered buffer (state_dict key ``*.centers``)."""

    def __init__(self, vmin: float = 0, vmax: float = 8, bins: int = 40, lengthscale: Optional[float] = None):
        super().__init__()
        self.vmin, self.vmax, self.bins = vmin, vmax, bins
        self.register_buffer("centers", torch.linspace(self.vmin, self.vmax, self.bins))
        if lengthscale is None:
            # gamma = 1 / mean spacing of the centres (NOT squared) - alignn/models/utils.py:30-34
            # (computed in float32 exactly as numpy does on the float32 buffer: 9.875 / 19.500002)
            ls32 = np.diff(self.centers.numpy()).mean(dtype=np.float32)
            self.lengthscale = float(ls32)
            self.gamma = float(np.float32(1.0) / ls32)
        else:
            self.lengthscale = lengthscale
            self.gamma = 1.0 / (lengthscale**2)

    def forward(self, distance: torch.Tensor) -> torch.Tensor:
        if torch_path.wanted(distance):  # float64 / 16-bit: plain torch (the HIP kernels are float32)
            return torch_path.rbf(distance, self.centers, self.gamma)
        return ops.rbf_expand(distance, self.centers, self.gamma)


class _TorchMLPLayer(nn.Module):
    """The reference's MLPLayer verbatim in structure (alignn.py:170-184) for the tiny per-crystal descriptor head of
    ``extra_features != 0`` - torch modules, same parameter names (``layer.0.*`` Linear, ``layer.1.*`` BatchNorm1d)."""

    def __init__(self, in_features: int, out_features: int):
        super().__init__()
        self.layer = nn.Sequential(nn.Linear(in_features, out_features), nn.BatchNorm1d(out_features), nn.SiLU())

    def forward(self, x):
        return self.layer(x)


_DEFERRED_BUMPS = None  # inside a whole-model forward: the counters to bump, flushed as ONE _foreach kernel


def _bump(bn: nn.Module, training: bool):
    """BatchNorm1d's ``num_batches_tracked += 1`` (29 one-element kernels per training step when done one by one)."""
    if training and getattr(bn, "num_batches_tracked", None) is not None:
        if _DEFERRED_BUMPS is not None:
            _DEFERRED_BUMPS.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked += 1


class _deferred_bumps:
    def __enter__(self):
        global _DEFERRED_BUMPS
        self.prev, _DEFERRED_BUMPS = _DEFERRED_BUMPS, []

    def __exit__(self, *exc):
        global _DEFERRED_BUMPS
        todo, _DEFERRED_BUMPS = _DEFERRED_BUMPS, self.prev
        if todo:
            torch._foreach_add_(todo, 1)


class MLPLayer(nn.Module):
    """Linear + BatchNorm1d + SiLU (alignn/models/alignn.py:170-184)."""

    _norm = "batch"  # the LayerNorm twin lives in alignn_amd.alignn_atomwise

    @staticmethod
    def _norm_layer(features: int) -> nn.Module:
        return nn.BatchNorm1d(features)

    def __init__(self, in_features: int, out_features: int):
        super().__init__()
        self.layer = nn.Sequential(nn.Linear(in_features, out_features), self._norm_layer(out_features), nn.SiLU())

    def forward(self, x):
        lin, bn = self.layer[0], self.layer[1]
        if torch_path.wanted(x, lin.weight):
            return torch_path.mlp_layer(self, x)
        # batch statistics or running statistics: the NORM module's own flag decides, as in the reference's nn.Sequential
        # (``for m in model.modules(): if isinstance(m, BatchNorm1d): m.eval()`` freezes the statistics of a training model)
        training = bn.training
        _bump(bn, training)
        with _lib.device_guard(x):
            return ops.MLPLayerFn.apply(
                x, lin.weight, lin.bias, bn.weight, bn.bias, getattr(bn, "running_mean", None),
                getattr(bn, "running_var", None), training, self._norm,
            )


def _as_csr(g, device) -> tuple[CSRGraph, bool]:
    """Accept a canonical CSRGraph, or any DGL-like graph (converted; edge rows then need permuting)."""
    if isinstance(g, CSRGraph):
        return g, True
    cached = getattr(g, "_alignn_amd_csr", None)
    if cached is not None and cached.src.device == device:
        return cached, False
    u, v = g.edges()
    csr = build_csr(u.to(device), v.to(device), g.num_nodes())
    try:
        g._alignn_amd_csr = csr
    except Exception:
        pass
    return csr, False


class EdgeGatedGraphConv(nn.Module):
    """Edge-gated graph convolution (arxiv:1711.07553), alignn/models/alignn.py:48-129.

    ``forward(g, node_feats, edge_feats) -> (x, y)``; ``g`` may be a DGL-like graph (features in
    the caller's edge order, as in the reference) or a canonical ``CSRGraph`` (features already in
    slot order - the fast path used inside ``ALIGNN``).
    """

    _norm = "batch"

    @staticmethod
    def _norm_layer(features: int) -> nn.Module:
        return nn.BatchNorm1d(features)

    def __init__(self, input_features: int, output_features: int, residual: bool = True):
        super().__init__()
        self.residual = residual
        self.src_gate = nn.Linear(input_features, output_features)
        self.dst_gate = nn.Linear(input_features, output_features)
        self.edge_gate = nn.Linear(input_features, output_features)
        self.bn_edges = self._norm_layer(output_features)
        self.src_update = nn.Linear(input_features, output_features)
        self.dst_update = nn.Linear(input_features, output_features)
        self.bn_nodes = self._norm_layer(output_features)

    def _fused_node_projection(self):
        """(wcat [4H,K], bcat [4H]) = src_gate | dst_gate | dst_update | src_update as ONE buffer whose row blocks
        ARE the four Linear parameters (their ``.data`` are views into it), so the fused node projection
        ``P = x wcat^T = A | Bd | Bh | Ux`` needs no per-forward ``torch.cat`` and its weight gradient lands in the four
        ``.grad``s as row blocks of one GEMM.  ``state_dict`` keys / shapes are the reference's; the aliasing is
        re-established lazily whenever something re-homed the parameters (``.to()``, ``load_state_dict(assign=True)``)."""
        lins = (self.src_gate, self.dst_gate, self.dst_update, self.src_update)
        ws, bs = [m.weight for m in lins], [m.bias for m in lins]
        if self.__dict__.get("_fused_pinned"):
            # somebody else owns the parameters' storage and could not keep the eight of them adjacent (FlatAdamW with a
            # frozen / unused member): concatenate per forward, do NOT re-home
            with torch.no_grad():
                return torch.cat([w.detach() for w in ws], 0), torch.cat([b.detach() for b in bs], 0)
        fused = self.__dict__.get("_fused_wb")
        if fused is not None:
            wcat, bcat = fused
            rows = ws[0].shape[0]
            wstep, bstep = ws[0].numel() * ws[0].element_size(), bs[0].numel() * bs[0].element_size()
            ok = wcat.device == ws[0].device and wcat.dtype == ws[0].dtype and wcat.shape[0] == 4 * rows
            for i in range(4):
                ok = ok and ws[i].data_ptr() == wcat.data_ptr() + i * wstep and bs[i].data_ptr() == bcat.data_ptr() + i * bstep
            if ok:
                return fused
        with torch.no_grad():
            wcat = torch.cat([w.detach() for w in ws], 0).contiguous()
            bcat = torch.cat([b.detach() for b in bs], 0).contiguous()
            rows = ws[0].shape[0]
            for i in range(4):
                ws[i].data = wcat[i * rows:(i + 1) * rows]
                bs[i].data = bcat[i * rows:(i + 1) * rows]
        self.__dict__["_fused_wb"] = (wcat, bcat)
        return wcat, bcat

    def _fused_parameter_groups(self):
        """(weights, biases) in the row-block order of the fused node projection - ``alignn_amd.optim.FlatAdamW`` keeps
        each list contiguous when it re-homes the parameters and then calls ``_adopt_fused_buffers``."""
        lins = (self.src_gate, self.dst_gate, self.dst_update, self.src_update)
        return [m.weight for m in lins], [m.bias for m in lins]

    def _gradient_runs(self):
        """Parameter lists whose GRADIENTS the whole-model backward (csrc/model.hip) writes as one contiguous block each, in
        this order: a norm's (dbeta | dgamma) pair.  ``alignn_amd.optim.FlatAdamW`` keeps such a run adjacent in its flat
        buffers, so that the backward can write straight into the optimizer's packed gradient buffer."""
        return [[self.bn_nodes.bias, self.bn_nodes.weight], [self.bn_edges.bias, self.bn_edges.weight]]

    def _adopt_fused_buffers(self):
        """The four weights (biases) are already adjacent row blocks of one buffer somebody else owns: use THAT as the
        fused buffer instead of re-fusing into a new one (which would take the parameters away from their owner)."""
        ws, bs = self._fused_parameter_groups()
        rows = ws[0].shape[0]
        for group in (ws, bs):
            step = group[0].numel() * group[0].element_size()
            if any(group[i].data_ptr() != group[0].data_ptr() + i * step for i in range(4)):
                raise ValueError("parameters are not adjacent")
        wcat = torch.as_strided(ws[0].data, (4 * rows, ws[0].shape[1]), (ws[0].shape[1], 1))
        bcat = torch.as_strided(bs[0].data, (4 * rows,), (1,))
        self.__dict__["_fused_wb"] = (wcat, bcat)
        self.__dict__.pop("_fused_pinned", None)

    def _pin_unfused(self):
        """The owner of the parameters' storage cannot offer them as adjacent row blocks: stop re-fusing (which would move
        them into a private buffer) and concatenate per forward instead."""
        self.__dict__["_fused_pinned"] = True
        self.__dict__.pop("_fused_wb", None)

    def _own_params_need_grad(self) -> bool:
        return any(p.requires_grad for p in self.parameters(recurse=True))

    def forward(self, g, node_feats: torch.Tensor, edge_feats: torch.Tensor, need_edge_out: bool = True):
        """``need_edge_out=False`` (internal): the caller will discard ``y``; it is then returned as None and
        its normalise/activate pass is skipped (BatchNorm running statistics are still updated)."""
        with _lib.device_guard(node_feats):
            return self._forward(g, node_feats, edge_feats, need_edge_out)

    def _forward(self, g, node_feats, edge_feats, need_edge_out):
        if torch_path.wanted(node_feats, edge_feats, self.edge_gate.weight):
            if isinstance(g, CSRGraph):
                return torch_path.edge_gated_conv(self, g.src, g.dst, g.n_nodes, node_feats, edge_feats, need_edge_out)
            u, v = g.edges()  # the caller's edge order, as in the reference: no permutation needed
            dev = node_feats.device
            return torch_path.edge_gated_conv(self, u.to(dev), v.to(dev), g.num_nodes(), node_feats, edge_feats, need_edge_out)
        csr, canonical = _as_csr(g, node_feats.device)
        # (a permuting gather is a plain torch consumer: it must not read a lane-T tensor without the event)
        y_in = edge_feats if canonical else ops.main_reads(edge_feats)[csr.perm]
        # fused node projection: P = x [W_sg; W_dg; W_du; W_su]^T -> A | Bd | Bh | Ux
        wcat, bcat = self._fused_node_projection()
        # the norm modules' own flags decide between batch and running statistics (see MLPLayer.forward); the kernels carry
        # ONE mode per convolution
        training = self.bn_nodes.training
        if self._norm == "batch" and self.bn_edges.training != training:
            raise NotImplementedError("EdgeGatedGraphConv: bn_nodes and bn_edges in different modes (one .eval(), one .train()) "
                                      "- the fused convolution runs one mode; freeze both or neither")
        if (self._norm == "batch" and not training and ops.INFER_FUSED
                and not (torch.is_grad_enabled() and (node_feats.requires_grad or edge_feats.requires_grad
                                                      or self._own_params_need_grad()))):
            # (with grad enabled the shortcut is taken only if NOTHING here can receive a gradient: frozen upstream
            # layers must not silently cost this layer's parameters their gradients)
            # pure inference (pretrained.py, model.eval() under no_grad): BatchNorm folded into the gate pass
            with torch.no_grad():
                x, y = ops.edge_gated_conv_infer(
                    csr, node_feats, y_in, wcat, bcat, self.edge_gate.weight, self.edge_gate.bias,
                    self.bn_nodes.weight, self.bn_nodes.bias, self.bn_nodes.running_mean, self.bn_nodes.running_var,
                    self.bn_edges.weight, self.bn_edges.bias, self.bn_edges.running_mean, self.bn_edges.running_var,
                    self.residual, need_edge_out)
            if not canonical and y is not None:
                y = ops.main_reads(y)[csr.inv]
            return x, y
        _bump(self.bn_nodes, training)
        _bump(self.bn_edges, training)
        x, y = ops.EdgeGatedConvFn.apply(
            csr, node_feats, y_in, wcat, bcat,
            self.src_gate.weight, self.dst_gate.weight, self.dst_update.weight, self.src_update.weight,
            self.src_gate.bias, self.dst_gate.bias, self.dst_update.bias, self.src_update.bias,
            self.edge_gate.weight, self.edge_gate.bias,
            self.bn_nodes.weight, self.bn_nodes.bias, getattr(self.bn_nodes, "running_mean", None),
            getattr(self.bn_nodes, "running_var", None),
            self.bn_edges.weight, self.bn_edges.bias, getattr(self.bn_edges, "running_mean", None),
            getattr(self.bn_edges, "running_var", None),
            training, self.residual, need_edge_out, self._norm,
        )
        if not canonical and y is not None:
            y = ops.main_reads(y)[csr.inv]
        return x, y


def _prepare_split_weights(model: nn.Module):
    """max|W| + slice images of every weight a split-product projection of this forward / backward may use (the fused node
    projection and the edge gate of every convolution, the wide MLP layers) in one batched call - ``ops.WeightPrep``."""
    if not model.training and not torch.is_grad_enabled():
        return  # (inference: the per-weight path slices what it needs, nothing is reused by a backward)
    mc = cmodel.model_cache(model)  # (not model.__dict__: the preparation holds weak references and device buffers)
    prep = mc.get("weight_prep")
    if prep is None:
        prep = mc["weight_prep"] = ops.WeightPrep()
    if prep.modules is None:  # (the module tree is walked once; the weights are looked up afresh every step)
        prep.modules = ([m for m in model.modules() if isinstance(m, EdgeGatedGraphConv)],
                        [m for m in model.modules() if isinstance(m, MLPLayer) and m.layer[0].weight.shape[0] >= 128
                         and m.layer[0].weight.shape[0] % 16 == 0 and m.layer[0].weight.shape[1] % 16 == 0])
    convs, mlps = prep.modules
    ws = []
    for m in convs:
        ws.append(m._fused_node_projection()[0])
        ws.append(m.edge_gate.weight)
    for m in mlps:
        ws.append(m.layer[0].weight)
    if ws and ws[0].dtype == torch.float32:
        prep.run(ws)


class ALIGNNConv(nn.Module):
    """Line graph update (alignn/models/alignn.py:132-167): bond-graph conv, then line-graph conv
    whose node inputs are the bond messages ``m``."""

    _conv = EdgeGatedGraphConv

    def __init__(self, in_features: int, out_features: int):
        super().__init__()
        self.node_update = self._conv(in_features, out_features)
        self.edge_update = self._conv(out_features, out_features)

    def forward(self, g, lg, x: torch.Tensor, y: torch.Tensor, z: torch.Tensor, need_z: bool = True):
        x, m = self.node_update(g, x, y)
        y, z = self.edge_update(lg, m, z, need_z)
        return x, y, z


class ALIGNN(nn.Module):
    """Atomistic line graph network (alignn/models/alignn.py:187-349)."""

    def __init__(self, config: ALIGNNConfig = ALIGNNConfig(name="alignn")):
        super().__init__()
        self.config = config
        self.classification = config.classification
        self.atom_embedding = MLPLayer(config.atom_input_features, config.hidden_features)
        self.edge_embedding = nn.Sequential(
            RBFExpansion(vmin=0, vmax=8.0, bins=config.edge_input_features),
            MLPLayer(config.edge_input_features, config.embedding_features),
            MLPLayer(config.embedding_features, config.hidden_features),
        )
        self.angle_embedding = nn.Sequential(
            RBFExpansion(vmin=-1, vmax=1.0, bins=config.triplet_input_features),
            MLPLayer(config.triplet_input_features, config.embedding_features),
            MLPLayer(config.embedding_features, config.hidden_features),
        )
        self.alignn_layers = nn.ModuleList(
            [ALIGNNConv(config.hidden_features, config.hidden_features) for _ in range(config.alignn_layers)]
        )
        self.gcn_layers = nn.ModuleList(
            [EdgeGatedGraphConv(config.hidden_features, config.hidden_features) for _ in range(config.gcn_layers)]
        )
        if self.classification:
            self.fc = nn.Linear(config.hidden_features, config.num_classes)
            self.softmax = nn.LogSoftmax(dim=1)
        else:
            self.fc = nn.Linear(config.hidden_features, config.output_features)
        if config.extra_features != 0:
            # per-crystal descriptor head (alignn.py:250-267): [N,k] and [B,H+k] matrices - plain torch modules with the
            # reference's own layout (Sequential(Linear, BatchNorm1d, SiLU) => identical state_dict keys), off the hot path
            k = config.extra_features
            self.extra_feature_embedding = _TorchMLPLayer(k, k)
            self.fc3 = nn.Linear(config.hidden_features + k, config.output_features)
            self.fc1 = _TorchMLPLayer(k + config.hidden_features, k + config.hidden_features)
            self.fc2 = _TorchMLPLayer(k + config.hidden_features, k + config.hidden_features)
        self.link = None
        self.link_name = config.link
        if config.link == "identity":
            self.link = lambda x: x
        elif config.link == "log":
            self.link = torch.exp
            avg_gap = 0.7  # reference initialises the bias to log(average band gap), alignn.py:273-278
            self.fc.bias.data = torch.tensor(np.log(avg_gap), dtype=torch.float)
        elif config.link == "logit":
            self.link = torch.sigmoid

    # ------------------------------------------------------------------
    def _batch(self, g) -> GraphBatch:
        dev = self.fc.weight.device
        if isinstance(g, GraphBatch):
            return g
        if isinstance(g, (tuple, list)):
            if isinstance(g[0], GraphBatch):
                return g[0]
            gg, lg = g[0], (g[1] if len(self.alignn_layers) > 0 else None)
        else:
            gg, lg = g, None
        # index structures cached on the graph object, features re-read every call (like the reference's forward)
        return cached_dgl_batch(gg, lg, dev)

    def forward(self, g: Union[Sequence, GraphBatch]):
        """``g`` = ``(g, lg, lat)`` of DGL-like graphs as in the reference (alignn.py:291-295), a bare
        graph when ``alignn_layers == 0``, or a prebuilt ``GraphBatch``.  Returns ``squeeze(out)``."""
        if torch_path.wanted(self.fc.weight):  # model.double() / .bfloat16(): see alignn_amd/torch_path.py
            return torch_path.alignn_forward(self, self._batch(g))
        b = self._batch(g)
        if cmodel.applicable(self, b):  # training step: the whole forward (and its backward) as ONE C call each
            out = cmodel.forward(self, b)
            if out is not None:
                return self._head(out)
        elif cmodel.infer_applicable(self, b):  # eval mode, no autograd: one C call (BatchNorm folded into the gate passes)
            out = cmodel.infer(self, b)
            if out is not None:
                return self._head(out)
        ops.new_weight_generation()  # weight images cached by an earlier forward are not this forward's (ops._WGEN)
        with _lib.device_guard(self.fc.weight), _deferred_bumps():
            _prepare_split_weights(self)  # (on the caller's stream, BEFORE the lanes fork: lane T waits for it)
            with ops.lanes(self.fc.weight.device):
                return self._forward(b)

    def _head(self, out):
        """link / classification head / squeeze (alignn.py:343-349) on the [B, out_features] readout."""
        if self.link:
            out = self.link(out)
        if self.classification:
            out = self.softmax(out)
        return torch.squeeze(out)

    def _forward(self, b: GraphBatch):
        if b.atom_features is None or b.r is None:
            raise ValueError("graph lacks ndata['atom_features'] / edata['r']")
        if len(self.alignn_layers) > 0:
            if b.lg is None or b.h is None:
                raise ValueError("alignn_layers > 0 needs the line graph with edata['h']")
            rbf, l1, l2 = self.angle_embedding[0], self.angle_embedding[1], self.angle_embedding[2]
            if (len(self.angle_embedding) == 3 and type(l1) is MLPLayer and type(l2) is MLPLayer
                    and not (self.angle_embedding._forward_hooks or self.angle_embedding._forward_pre_hooks)
                    # (the fused passes run ONE mode for both layers: a block frozen with .eval() takes the chain of layers)
                    and all(m.training == self.training for m in (l1, l2, l1.layer[1], l2.layer[1]))
                    and ops.angle_fused_applies(b.h, rbf, l1, l2, self.training)):
                if self.training:
                    _bump(l1.layer[1], True)
                    _bump(l2.layer[1], True)
                    z = ops.angle_embed(b.h, rbf, l1, l2)  # csrc/angle.hip: no [T, bins] / [T, 64] / [T, 256] intermediates
                else:
                    z = ops.angle_embed_infer(b.h, rbf, l1, l2)
            else:
                z = self.angle_embedding(b.h)
        x = self.atom_embedding(b.atom_features)
        y = self.edge_embedding(ops.bond_length(b.r))
        # the triplet features of the last ALIGNN layer and the bond features of the last GCN layer are
        # never read again (alignn.py:317-325): do not materialise them
        n_a, n_g = len(self.alignn_layers), len(self.gcn_layers)
        for i, layer in enumerate(self.alignn_layers):
            x, y, z = layer(b.g, b.lg, x, y, z, need_z=i + 1 < n_a)
        for i, layer in enumerate(self.gcn_layers):
            x, y = layer(b.g, x, y, need_edge_out=i + 1 < n_g)
        h = ops.AvgPoolFn.apply(x, b.graph_ptr)
        if self.config.extra_features != 0:  # alignn.py:328-339
            if b.extra_features is None:
                raise ValueError("extra_features != 0 needs g.ndata['extra_features'] (GraphBatch.extra_features)")
            feats = self.extra_feature_embedding(b.extra_features)
            if feats.shape[1] % 4 == 0:
                h_feat = ops.AvgPoolFn.apply(feats, b.graph_ptr)
            else:  # the pooling kernel works in float4 columns: pad the k descriptor columns
                pad = torch.nn.functional.pad(feats, (0, -feats.shape[1] % 4))
                h_feat = ops.AvgPoolFn.apply(pad, b.graph_ptr)[:, :feats.shape[1]]
            out = self.fc3(self.fc2(self.fc1(torch.cat((h, h_feat), 1))))
        else:
            out = ops.linear(h, self.fc.weight, self.fc.bias.reshape(-1))
        if self.link:
            out = self.link(out)
        if self.classification:
            out = self.softmax(out)
        return torch.squeeze(out)
