"""MI355X-native drop-in for ``alignn.models.alignn_atomwise`` - the LayerNorm flavour that
``alignn/train.py`` actually trains (``"alignn_" in config.model.name``, train.py:238).

Reference: ``/root/reference/alignn/models/alignn_atomwise.py`` (``ALIGNNAtomWiseConfig`` :28-77,
``EdgeGatedGraphConv`` with ``nn.LayerNorm`` :127-208, ``ALIGNNConv`` :211-246, ``ALIGNNAtomWise``
:249-660) and ``alignn/models/utils.py:277-292`` (``MLPLayer`` = Linear + LayerNorm + SiLU).  Same class names,
constructor arguments, ``state_dict`` keys and result dictionary.

Two execution paths:

* energy / property (``calculate_gradient=False`` or ``gradwise_weight == 0`` - what
  ``examples/sample_data/config_example.json`` runs): the fused kernels (LayerNorm flavour), including the in-forward
  recomputation of the bond-angle cosines (``lg_on_fly``, :424-431);
* force / stress head (``calculate_gradient=True``: ``autograd.grad(..., create_graph=True)`` inside the forward,
  :512-638, differentiated again by the training loss): composed from the twice-differentiable primitives of
  ``alignn_amd.ff`` (HIP gather / segment-sum / MFMA GEMMs + torch element-wise), same formulas.
"""

from __future__ import annotations

from typing import Literal, Sequence, Union

import numpy as np
import torch
from torch import nn

from . import alignn as _bn
from . import _lib, ops, torch_path
from .alignn import _Base, _CONFIG, RBFExpansion  # noqa: F401  (RBFExpansion re-exported like the reference)
from .graph import GraphBatch, cached_dgl_batch


FUSED_FORCE_TRAINING = True  # tests flip this to compare against the composed twice-differentiable path (alignn_amd.ff)


class ALIGNNAtomWiseConfig(_Base):
    """Field-for-field the reference's schema (alignn/models/alignn_atomwise.py:31-72)."""

    name: Literal["alignn_atomwise"]
    alignn_layers: int = 2
    gcn_layers: int = 2
    atom_input_features: int = 1
    edge_input_features: int = 80
    triplet_input_features: int = 40
    embedding_features: int = 64
    hidden_features: int = 64
    output_features: int = 1
    grad_multiplier: int = -1
    calculate_gradient: bool = True
    atomwise_output_features: int = 0
    graphwise_weight: float = 1.0
    gradwise_weight: float = 1.0
    stresswise_weight: float = 0.0
    atomwise_weight: float = 0.0
    link: Literal["identity", "log", "logit"] = "identity"
    zero_inflated: bool = False
    classification: bool = False
    force_mult_natoms: bool = False
    energy_mult_natoms: bool = True
    include_pos_deriv: bool = False
    use_cutoff_function: bool = False
    inner_cutoff: float = 3
    stress_multiplier: float = 1
    add_reverse_forces: bool = True
    lg_on_fly: bool = True
    batch_stress: bool = True
    multiply_cutoff: bool = False
    use_penalty: bool = True
    extra_features: int = 0
    exponent: int = 5
    penalty_factor: float = 0.1
    penalty_threshold: float = 1
    additional_output_features: int = 0
    additional_output_weight: float = 0

    model_config = _CONFIG


class MLPLayer(_bn.MLPLayer):
    """Linear + LayerNorm + SiLU (alignn/models/utils.py:277-292)."""

    _norm = "layer"

    @staticmethod
    def _norm_layer(features: int) -> nn.Module:
        return nn.LayerNorm(features)


class EdgeGatedGraphConv(_bn.EdgeGatedGraphConv):
    """Edge-gated convolution with LayerNorm on both branches (alignn_atomwise.py:127-208)."""

    _norm = "layer"

    @staticmethod
    def _norm_layer(features: int) -> nn.Module:
        return nn.LayerNorm(features)


class ALIGNNConv(_bn.ALIGNNConv):
    _conv = EdgeGatedGraphConv


def cutoff_function_based_edges(r, inner_cutoff=4, exponent=3):
    """Smooth polynomial envelope (alignn_atomwise.py:97-124); a few ops on the [E] bond-length vector."""
    ratio = r / inner_cutoff
    c1 = -(exponent + 1) * (exponent + 2) / 2
    c2 = exponent * (exponent + 2)
    c3 = -exponent * (exponent + 1) / 2
    env = 1 + c1 * ratio**exponent + c2 * ratio ** (exponent + 1) + c3 * ratio ** (exponent + 2)
    return torch.where(r <= inner_cutoff, env, torch.zeros_like(r))


def positions_to_bond_vectors(gg, lat, dev):
    """``compute_cartesian_coordinates`` + ``compute_pair_vector_and_distance`` (alignn/models/utils.py:47-56, 95-126):
    Cartesian positions from ``g.ndata["frac_coords"]`` and the lattices, bond vectors
    ``cart[dst] + g.edata["images"] - cart[src]`` in the caller's edge order."""
    if "frac_coords" not in gg.ndata or "images" not in gg.edata:
        raise ValueError("this option recomputes the bond vectors from positions: it needs g.ndata['frac_coords'] and "
                         "g.edata['images'] (Cartesian image shifts), as the reference does")
    u, v = gg.edges()
    lat = torch.as_tensor(lat).to(dev, torch.float32)
    if lat.dim() == 2:
        lat = lat.unsqueeze(0)
    bnn = torch.as_tensor(gg.batch_num_nodes()).to(dev, torch.int64)
    which = torch.repeat_interleave(torch.arange(bnn.numel(), device=dev), bnn)
    frac = gg.ndata["frac_coords"].to(dev, torch.float32)
    cart = torch.bmm(frac.unsqueeze(1), lat[which]).squeeze(1)
    r = cart[torch.as_tensor(v).to(dev).long()] + gg.edata["images"].to(dev, torch.float32) - cart[torch.as_tensor(u).to(dev).long()]
    return cart, r


def _check_pos_deriv(cfg, b):
    if cfg.stresswise_weight != 0:
        raise ValueError("include_pos_deriv gives no pair forces, so no virial: upstream fails here too (stresswise_weight must be 0)")


def _single_virial(b: GraphBatch, pair_forces):
    """batch_stress=False (alignn_atomwise.py:572-593): -160.21766208 * r^T f / (2 V[0]) with r from the positions."""
    r_pos = b.r_from_positions
    if r_pos is None:
        raise ValueError("batch_stress=False takes the bond vectors from the positions: pass (g, lg, lat) with "
                         "g.ndata['frac_coords'] and g.edata['images']")
    return -160.21766208 * (r_pos.to(pair_forces.dtype).t() @ pair_forces) / (2 * b.volume[0].to(pair_forces.dtype))


class _TorchLNMLPLayer(nn.Module):
    """models/utils.py:277-292 verbatim in structure, for the tiny descriptor head of ``extra_features != 0``."""

    def __init__(self, in_features: int, out_features: int):
        super().__init__()
        self.layer = nn.Sequential(nn.Linear(in_features, out_features), nn.LayerNorm(out_features), nn.SiLU())

    def forward(self, x):
        return self.layer(x)


class ALIGNNAtomWise(nn.Module):
    """Atomistic line graph network, LayerNorm flavour (alignn_atomwise.py:249-660)."""

    def __init__(self, config: ALIGNNAtomWiseConfig = ALIGNNAtomWiseConfig(name="alignn_atomwise")):
        super().__init__()
        self.classification = config.classification
        self.config = config
        if self.config.gradwise_weight == 0:
            self.config.calculate_gradient = False
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
        if config.extra_features != 0:
            # per-crystal descriptor head (alignn_atomwise.py:314-333): tiny [N,k] / [B,H+k] matrices - torch modules
            # with the reference's layout (Sequential(Linear, LayerNorm, SiLU) => identical state_dict keys)
            k = config.extra_features
            self.extra_feature_embedding = _TorchLNMLPLayer(k, k)
            self.fc3 = nn.Linear(config.hidden_features + k, config.output_features)
            self.fc1 = _TorchLNMLPLayer(k + config.hidden_features, k + config.hidden_features)
            self.fc2 = _TorchLNMLPLayer(k + config.hidden_features, k + config.hidden_features)
        if config.atomwise_output_features > 0:
            self.fc_atomwise = nn.Linear(config.hidden_features, config.atomwise_output_features)
        if config.additional_output_features:
            self.fc_additional_output = nn.Linear(config.hidden_features, config.additional_output_features)
        if self.classification:
            self.fc = nn.Linear(config.hidden_features, 1)
            self.softmax = nn.Sigmoid()
        else:
            self.fc = nn.Linear(config.hidden_features, config.output_features)
        self.link = None
        self.link_name = config.link
        if config.link == "identity":
            self.link = lambda x: x
        elif config.link == "log":
            self.link = torch.exp
            self.fc.bias.data = torch.tensor(np.log(0.7), dtype=torch.float)
        elif config.link == "logit":
            self.link = torch.sigmoid

    def _batch(self, g) -> GraphBatch:
        dev = self.fc.weight.device
        if isinstance(g, GraphBatch):
            return g
        if isinstance(g, (tuple, list)) and isinstance(g[0], GraphBatch):
            return g[0]
        gg = g[0]
        # (g, lg, lat) as the training loop passes it, or (g, lat): then L(g) is built inside the forward
        # (alignn_atomwise.py:376-386) - here on the device, straight into the canonical block layout
        lg = g[1] if (len(self.alignn_layers) > 0 and len(g) == 3) else None
        need_lg = len(self.alignn_layers) > 0 and lg is None
        if need_lg and not self.config.lg_on_fly:
            raise ValueError("forward((g, lat)) has no precomputed bond cosines: it needs lg_on_fly=True")
        # index structures cached on the graph object; features (r, atom_features, V, positions) re-read every call
        batch = cached_dgl_batch(gg, lg, dev, build_line_graph=need_lg)
        cfg = self.config
        if cfg.include_pos_deriv or (cfg.calculate_gradient and cfg.stresswise_weight != 0 and not cfg.batch_stress):
            # these branches take the bond vectors from the positions (alignn_atomwise.py:405-412, 572-577)
            _, r_pos = positions_to_bond_vectors(gg, g[-1], dev)
            r_pos = r_pos[batch.g.perm].contiguous()
            if cfg.include_pos_deriv:
                batch.r = r_pos
            else:
                batch.r_from_positions = r_pos
        return batch

    def _forward_ff(self, b: GraphBatch, with_forces: bool = True):
        """Energy + forces (+ stress): alignn_atomwise.py:364-660 with calculate_gradient=True.  Composed from the
        twice-differentiable primitives of ``alignn_amd.ff`` so that ``autograd.grad(create_graph=True)`` and the
        training loss's backward through it both work."""
        from . import ff

        cfg = self.config
        n_a = len(self.alignn_layers)
        dt = self.fc.weight.dtype
        x = ff.mlp_layer(b.atom_features.to(dt), self.atom_embedding)
        r = b.r.detach().to(dt).clone().requires_grad_(with_forces)  # canonical bond order; :420
        bondlength = torch.norm(r, dim=1)
        if n_a > 0:
            h = ff.bond_cosines(r, b.lg) if cfg.lg_on_fly else b.h.to(dt)
            z = ff.mlp_layer(ff.mlp_layer(ff.rbf(h, self.angle_embedding[0]), self.angle_embedding[1]), self.angle_embedding[2])
        d_in = bondlength
        c_off = None
        if cfg.use_cutoff_function:
            env = cutoff_function_based_edges(bondlength, cfg.inner_cutoff, cfg.exponent)
            if cfg.multiply_cutoff:
                c_off = env.unsqueeze(1)
            else:
                # upstream OVERWRITES ``bondlength`` with the envelope here (:446-451), so the short-bond penalty
                # below is taken on the envelope too
                d_in = bondlength = env
        y = ff.mlp_layer(ff.mlp_layer(ff.rbf(d_in, self.edge_embedding[0]), self.edge_embedding[1]), self.edge_embedding[2])
        if c_off is not None:
            y = y * c_off
        for layer in self.alignn_layers:
            x, m = ff.edge_gated_conv(b.g, x, y, layer.node_update)
            y, z = ff.edge_gated_conv(b.lg, m, z, layer.edge_update)
        for layer in self.gcn_layers:
            x, y = ff.edge_gated_conv(b.g, x, y, layer)
        counts = (b.graph_ptr[1:] - b.graph_ptr[:-1]).to(dt)
        if "atoms_by_graph" not in b.cache:  # (repeat_interleave sizes its output on the host: once per batch)
            b.cache["atoms_by_graph"] = ff.by_graph(b.graph_ptr)
        hpool = ff.segment_sum(x, b.cache["atoms_by_graph"]) / counts.unsqueeze(1)
        if cfg.extra_features != 0:
            # :391-393, 468-475: per-crystal descriptor head (torch modules: twice differentiable as they are).  fc3's
            # [B,1] output is NOT squeezed upstream, so ``out * natoms`` below broadcasts to [B,B] exactly as it does there
            if b.extra_features is None:
                raise ValueError("extra_features != 0 needs g.ndata['extra_features'] (GraphBatch.extra_features)")
            feats = self.extra_feature_embedding(b.extra_features.to(dt))
            h_feat = ff.segment_sum(feats, b.cache["atoms_by_graph"]) / counts.unsqueeze(1)
            hpool = self.fc2(self.fc1(torch.cat((hpool, h_feat), 1)))
            out = self.fc3(hpool)
        else:
            out = torch.squeeze(ff.linear(hpool, self.fc))
        additional_out = torch.empty(1)
        if cfg.additional_output_features > 0:
            additional_out = ff.linear(hpool, self.fc_additional_output)
        atomwise_pred = torch.empty(1)
        if cfg.atomwise_output_features > 0 and cfg.atomwise_weight != 0:
            atomwise_pred = ff.linear(x, self.fc_atomwise)
        en_out, out = self._total_energy(out, counts, bondlength)  # :494-510
        if not with_forces:
            return self._finish(out, additional_out, torch.empty(1), torch.empty(1), atomwise_pred)
        pair_forces = cfg.grad_multiplier * torch.autograd.grad(
            en_out, r, grad_outputs=torch.ones_like(en_out), create_graph=True, retain_graph=True)[0]  # :530-539
        stress = torch.empty(1)
        if cfg.include_pos_deriv:
            # :513-524: grad_multiplier * d(en_out * N)/d(cart_coords); with r_e = cart[dst] + image - cart[src] the
            # chain rule is the in-minus-out reduction of dE/dr
            _check_pos_deriv(cfg, b)
            forces = torch.squeeze(ff.pair_force_reduce(pair_forces, b.g, True)) * b.g.n_nodes
            return self._finish(out, additional_out, forces, stress, atomwise_pred)
        if cfg.force_mult_natoms:
            pair_forces = pair_forces * b.g.n_nodes
        forces = torch.squeeze(ff.pair_force_reduce(pair_forces, b.g, cfg.add_reverse_forces))  # :547-565
        if cfg.stresswise_weight != 0:
            if b.volume is None:
                raise ValueError("stress needs the cell volumes: g.ndata['V'] (or GraphBatch.volume)")
            if not cfg.batch_stress:  # :572-593: ONE virial over all bonds, the first crystal's volume, factor 1/2
                return self._finish(out, additional_out, forces, _single_virial(b, pair_forces), atomwise_pred)
            if b.volume is None:
                raise ValueError("stress needs the cell volumes: g.ndata['V'] (or GraphBatch.volume)")
            # per crystal: -160.21766208 * r_g^T f_g / V_g (:615-638); bonds of a crystal are contiguous slots
            outer = (r.unsqueeze(2) * pair_forces.unsqueeze(1)).reshape(-1, 9)
            if "bonds_by_graph" not in b.cache:
                b.cache["bonds_by_graph"] = ff.by_graph(b.edge_graph_ptr)
            st = ff.segment_sum(outer, b.cache["bonds_by_graph"]).reshape(-1, 3, 3)
            stress = cfg.stress_multiplier * (-160.21766208) * st / b.volume.to(dt).reshape(-1, 1, 1)
        return self._finish(out, additional_out, forces, stress, atomwise_pred)

    def _total_energy(self, out, counts, bondlength):
        """alignn_atomwise.py:494-510 -> (en_out, out): energy per crystal (x atoms if ``energy_mult_natoms``) plus the
        short-bond penalty summed over ALL bonds of the batch.  Without ``energy_mult_natoms`` upstream's ``en_out`` is
        the very tensor ``out`` and ``en_out += total_penalty`` is in place: the returned ``out`` carries the penalty."""
        cfg = self.config
        en_out = out * counts if cfg.energy_mult_natoms else out
        if cfg.use_penalty:
            pen = torch.where(bondlength < cfg.penalty_threshold,
                              cfg.penalty_factor * (cfg.penalty_threshold - bondlength), torch.zeros_like(bondlength))
            en_out = en_out + torch.sum(pen)
            if not cfg.energy_mult_natoms:
                out = en_out
        return en_out, out

    def _finish(self, out, additional_out, forces, stress, atomwise_pred):
        if self.link:
            out = self.link(out)
        if self.classification:
            out = self.softmax(out)
        return {"out": out, "additional": additional_out, "grad": forces, "stresses": stress,
                "atomwise_pred": atomwise_pred}

    def forward(self, g: Union[Sequence, GraphBatch]):
        cfg = self.config
        if torch_path.wanted(self.fc.weight):
            # float64 / 16-bit module (alignn/train.py:89-95): the composed path on plain torch operations (alignn_amd/ff.py
            # switches per dtype), energies with or without forces - twice differentiable by autograd like the reference
            return self._forward_ff(self._batch(g), with_forces=bool(cfg.calculate_gradient))
        b = self._batch(g)
        res = self._forward_c(b)  # the whole model as one or two C calls (alignn_amd/cmodel.py) where that applies
        if res is not None:
            return res
        ops.new_weight_generation()  # (see ops._WGEN)
        with _lib.device_guard(self.fc.weight):
            _bn._prepare_split_weights(self)  # all weight images of the step in one call (ops.WeightPrep)
        # Forces: in training the loss differentiates THROUGH them -> composed, twice-differentiable path.  In eval
        # mode (MD / calculators: alignn/ff/calculators.py) only the first derivative is needed -> the fused kernels
        # with their hand-written backward, r as a leaf.
        fused_forces = cfg.calculate_gradient and not self.training and torch.is_grad_enabled()
        if cfg.calculate_gradient and not fused_forces:
            from . import ff2

            if FUSED_FORCE_TRAINING and torch.is_grad_enabled() and ff2.supported(cfg):
                # training through the forces: values by the fused first-order path, the loss gradient by ONE reverse
                # pass over a forward pass that carries tangents (alignn_amd/ff2.py) - same gradients as
                # autograd.grad(create_graph=True) + a second backward, on fused kernels
                with _lib.device_guard(self.fc.weight):
                    out, forces, stress = ff2.ForcesFn.apply(self, b, *self.parameters())
                has_stress = cfg.stresswise_weight != 0
                return self._finish(out, torch.empty(1), forces, stress if has_stress else torch.empty(1), torch.empty(1))
            return self._forward_ff(b)
        if fused_forces:
            with ops.no_param_grad():
                return self._forward_fused(b, True)
        with ops.lanes(self.fc.weight.device):
            return self._forward_fused(b, False)

    def _forward_c(self, b: GraphBatch):
        """The whole-model C entry points (csrc/model.hip) where they apply - the plain energy / force / stress head, float32 on
        a HIP device, no hooks, every kernel-choice switch at its default: training through the forces (``alignn_ff_eval`` +
        ``alignn_ff_grad``: what alignn/train.py:291-387 runs), force evaluation in eval mode (``alignn_ff_eval``: MD,
        alignn/ff/calculators.py:280-291) and energy-only training (``alignn_model_fwd / _bwd``, LayerNorm flavour).  None:
        the per-operator path below."""
        from . import cmodel, ff2

        cfg = self.config
        grad = torch.is_grad_enabled()
        if cfg.calculate_gradient:
            if not (grad and ff2.supported(cfg) and ff2.REUSE_FORWARD):
                return None
            if self.training:
                if not (FUSED_FORCE_TRAINING and cmodel.atomwise_applicable(self, b, True)):
                    return None
                res = cmodel.ff_train(self, b)
            else:
                if not cmodel.atomwise_applicable(self, b, False):
                    return None
                res = cmodel.ff_eval(self, b)
            if res is None:
                return None
            out, forces, stress = res
            has_stress = cfg.stresswise_weight != 0
            return self._finish(torch.squeeze(out), torch.empty(1), torch.squeeze(forces), stress if has_stress else torch.empty(1),
                                torch.empty(1))
        if not (self.training and cfg.output_features is not None and (not cfg.use_penalty or cfg.energy_mult_natoms)
                and cmodel.atomwise_applicable(self, b, grad)):
            return None
        with _lib.device_guard(self.fc.weight):
            h = ops.bond_cosines(b.r, b.lg) if cfg.lg_on_fly else b.h
        out = cmodel.forward(self, b, h=h)
        if out is None:
            return None
        return self._finish(torch.squeeze(out), torch.empty(1), torch.empty(1), torch.empty(1), torch.empty(1))

    def _forward_fused(self, b: GraphBatch, fused_forces: bool):
        cfg = self.config
        n_a, n_g = len(self.alignn_layers), len(self.gcn_layers)
        x = self.atom_embedding(b.atom_features)
        r = b.r.detach().clone().requires_grad_(True) if fused_forces else b.r
        bondlength = ops.bond_length(r)
        if n_a > 0:
            # lg_on_fly (default): recompute the cosines from r inside the forward (:424-431); otherwise use
            # the loader's lg.edata["h"] (:370-371)
            h = ops.bond_cosines(r, b.lg) if cfg.lg_on_fly else b.h
            z = self.angle_embedding(h)
        if cfg.use_cutoff_function:
            if cfg.multiply_cutoff:
                c_off = cutoff_function_based_edges(bondlength, cfg.inner_cutoff, cfg.exponent).unsqueeze(1)
                # the multiply is a plain torch consumer: join lane T first (inside lanes() the last embedding layer
                # may have run there)
                y = ops.main_reads(self.edge_embedding(bondlength)) * c_off
            else:  # ``bondlength`` becomes the envelope from here on (:446-451), also for the penalty
                bondlength = cutoff_function_based_edges(bondlength, cfg.inner_cutoff, cfg.exponent)
                y = self.edge_embedding(bondlength)
        else:
            y = self.edge_embedding(bondlength)
        for i, layer in enumerate(self.alignn_layers):
            x, y, z = layer(b.g, b.lg, x, y, z, need_z=i + 1 < n_a)
        for i, layer in enumerate(self.gcn_layers):
            x, y = layer(b.g, x, y, need_edge_out=i + 1 < n_g)
        out = torch.empty(1)
        additional_out = torch.empty(1)
        if cfg.output_features is not None:
            hpool = ops.AvgPoolFn.apply(x, b.graph_ptr)
            if cfg.extra_features != 0:  # alignn_atomwise.py:468-473 (note: fc3's output is NOT squeezed upstream)
                if b.extra_features is None:
                    raise ValueError("extra_features != 0 needs g.ndata['extra_features'] (GraphBatch.extra_features)")
                feats = self.extra_feature_embedding(b.extra_features)
                pad = torch.nn.functional.pad(feats, (0, -feats.shape[1] % 4))  # the pooling kernel works in float4 columns
                h_feat = ops.AvgPoolFn.apply(pad.contiguous(), b.graph_ptr)[:, :feats.shape[1]]
                hpool = self.fc2(self.fc1(torch.cat((hpool, h_feat), 1)))
                out = self.fc3(hpool)
            else:
                out = torch.squeeze(ops.linear(hpool, self.fc.weight, self.fc.bias.reshape(-1)))
            if cfg.additional_output_features > 0:
                additional_out = ops.linear(hpool, self.fc_additional_output.weight, self.fc_additional_output.bias)
        atomwise_pred = torch.empty(1)
        if cfg.atomwise_output_features > 0 and cfg.atomwise_weight != 0:
            atomwise_pred = ops.linear(x, self.fc_atomwise.weight, self.fc_atomwise.bias)
        forces, stress = torch.empty(1), torch.empty(1)
        if cfg.output_features is not None and (fused_forces or (cfg.use_penalty and not cfg.energy_mult_natoms)):
            counts = (b.graph_ptr[1:] - b.graph_ptr[:-1]).to(torch.float32)
            en_out, out = self._total_energy(out, counts, bondlength)  # (the penalty reaches ``out`` without forces too)
            if fused_forces:
                forces, stress = self._forces_from_energy(b, en_out, r)
                out = out.detach()
        if self.link:
            out = self.link(out)
        if self.classification:
            out = self.softmax(out)
        return {
            "out": out,
            "additional": additional_out,
            "grad": forces,
            "stresses": stress,
            "atomwise_pred": atomwise_pred,
        }

    def _forces_from_energy(self, b: GraphBatch, en_out, r):
        """alignn_atomwise.py:512-638 for inference: E_tot -> pair forces -dE/dr (one fused backward) -> per-atom
        forces and per-crystal virial stresses.  Nothing here is differentiated again."""
        cfg = self.config
        (g_r,) = torch.autograd.grad(en_out, r, grad_outputs=torch.ones_like(en_out))
        pair_forces = cfg.grad_multiplier * g_r
        gg = b.g
        if cfg.include_pos_deriv:
            _check_pos_deriv(cfg, b)
            f = ops._segment_sum_raw(pair_forces, gg.seg_ptr, None, gg.seg_node, gg.n_nodes)
            f = f - ops._segment_sum_raw(pair_forces, gg.out_ptr, gg.out_slot, None, gg.n_nodes)
            return torch.squeeze(f) * gg.n_nodes, torch.empty(1)
        if cfg.stresswise_weight != 0 and not cfg.batch_stress:
            if b.volume is None:
                raise ValueError("stress needs the cell volumes: g.ndata['V'] (or GraphBatch.volume)")
            if cfg.force_mult_natoms:
                pair_forces = pair_forces * b.g.n_nodes
            forces = ops._segment_sum_raw(pair_forces, gg.seg_ptr, None, gg.seg_node, gg.n_nodes)
            if cfg.add_reverse_forces:
                forces = forces - ops._segment_sum_raw(pair_forces, gg.out_ptr, gg.out_slot, None, gg.n_nodes)
            return torch.squeeze(forces), _single_virial(b, pair_forces)
        # forces per atom = in-minus-out reduction of the pair forces (copy_e / sum on g and on dgl.reverse(g)), virial stress per
        # crystal: one kernel each (csrc/ff.hip) - the same two the whole-model call alignn_ff_eval launches
        lib = _lib.load()
        g_r = g_r.contiguous()
        scale = float(cfg.grad_multiplier) * (gg.n_nodes if cfg.force_mult_natoms else 1)
        forces = torch.empty(gg.n_nodes, 3, dtype=torch.float32, device=g_r.device)
        _lib.check(lib.alignn_pair_force_reduce(g_r.data_ptr(), scale, gg.seg_ptr.data_ptr(), gg.out_ptr.data_ptr(),
                                                gg.out_slot.data_ptr(), int(cfg.add_reverse_forces), forces.data_ptr(), gg.n_nodes,
                                                _lib.stream()), "pair_force_reduce")
        forces = torch.squeeze(forces)
        stress = torch.empty(1)
        if cfg.stresswise_weight != 0:
            vol = b.cell_volumes()  # float32, contiguous, on the device, one per crystal (or ValueError)
            stress = torch.empty(b.batch_size, 3, 3, dtype=torch.float32, device=g_r.device)
            rr = r.detach().contiguous()
            _lib.check(lib.alignn_virial_stress(rr.data_ptr(), g_r.data_ptr(), scale, b.graph_ptr.data_ptr(), gg.seg_ptr.data_ptr(),
                                                vol.data_ptr(), float(cfg.stress_multiplier) * (-160.21766208), stress.data_ptr(),
                                                b.batch_size, _lib.stream()), "virial_stress")
        return forces, stress
