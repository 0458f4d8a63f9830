"""eALIGNNAtomWise: the LayerNorm ALIGNN on a FILTERED bond graph.

Reference: ``alignn/models/ealignn_atomwise.py``.  Differences from ``ALIGNNAtomWise`` (``:276-444``):

* the bond vectors are recomputed inside the forward from ``g.ndata["frac_coords"]``, the lattices and the Cartesian
  image shifts ``g.edata["images"]`` (``compute_cartesian_coordinates`` / ``compute_pair_vector_and_distance``,
  ``models/utils.py:47-126``) - ``g.edata["r"]`` is ignored;
* bonds longer than ``inner_cutoff`` are DROPPED before anything else (``lightweight_line_graph``, ``:129-222``); the
  convolutions, the line graph (always rebuilt: ``g.line_graph``), the cosines, the penalty and the virial all live
  on that lighter graph - on MI355X this is simply a smaller canonical ``GraphBatch`` (``graph.line_graph_of`` gives
  the dense block structure for any bond graph), so every fused kernel applies unchanged;
* pair forces are always scaled by the number of atoms of the batch and always reduced in-minus-out; no cutoff
  envelope, no ``link``;
* the forces are post-processed by ``remove_net_torque`` (``:316-400``; a 3x3 solve per crystal on [N,3] data - torch).

Same parameter names as the reference class (``state_dict`` loads unchanged).  ``train()`` differentiates through the
forces on the composed path, ``eval()`` takes them from the fused kernels, exactly as ``ALIGNNAtomWise``.
"""

from __future__ import annotations

from typing import Literal, Sequence, Union

import torch

from .alignn_atomwise import _CONFIG, ALIGNNAtomWise, ALIGNNAtomWiseConfig, positions_to_bond_vectors
from .graph import GraphBatch

try:  # pydantic v2, as the reference's BaseSettings shim
    from pydantic import BaseModel
except Exception as e:  # pragma: no cover
    raise ImportError("pydantic is required for the config classes") from e


class eALIGNNAtomWiseConfig(BaseModel):
    """Hyper-parameter schema of ``alignn/models/ealignn_atomwise.py:31-69`` (same names, same defaults)."""

    name: Literal["ealignn_atomwise"]
    alignn_layers: int = 2
    gcn_layers: int = 2
    atom_input_features: int = 1
    edge_input_features: int = 80
    triplet_input_features: int = 40
    embedding_features: int = 64
    hidden_features: int = 64
    output_features: int = 1
    calculate_gradient: bool = True
    atomwise_output_features: int = 0
    graphwise_weight: float = 1.0
    gradwise_weight: float = 1.0
    stresswise_weight: float = 0.0
    atomwise_weight: float = 0.0
    classification: bool = False
    energy_mult_natoms: bool = True
    remove_torque: bool = True
    inner_cutoff: float = 4
    use_penalty: bool = True
    extra_features: int = 0
    penalty_factor: float = 0.1
    penalty_threshold: float = 1
    additional_output_features: int = 0
    additional_output_weight: float = 0
    stress_multiplier: float = 1
    grad_multiplier: int = -1
    link: Literal["identity", "log", "logit"] = "identity"
    zero_inflated: bool = False
    force_mult_natoms: bool = False
    include_pos_deriv: bool = False
    use_cutoff_function: bool = False
    add_reverse_forces: bool = True
    lg_on_fly: bool = True
    batch_stress: bool = True
    multiply_cutoff: bool = False
    exponent: int = 5

    model_config = _CONFIG


def remove_net_torque(positions: torch.Tensor, forces: torch.Tensor, n_nodes: torch.Tensor) -> torch.Tensor:
    """``models/utils.py:316-400``: the centre of mass and the net torque are taken over the WHOLE batch (as upstream
    does), the 3x3 systems per crystal."""
    com = positions.sum(0) / n_nodes.float().sum()
    r = positions - com
    tau = torch.cross(r, forces, dim=1).sum(0)
    idx = torch.repeat_interleave(torch.arange(n_nodes.numel(), device=positions.device), n_nodes)
    B = n_nodes.numel()
    s = torch.zeros(B, device=positions.device).index_add_(0, idx, (r * r).sum(1))
    S = torch.zeros(B, 3, 3, device=positions.device).index_add_(0, idx, r.unsqueeze(2) @ r.unsqueeze(1))
    M = S - s.view(-1, 1, 1) * torch.eye(3, device=positions.device).expand(B, -1, -1)
    b = (-tau).expand(B, 3)
    try:
        mu = torch.linalg.solve(M, b)
    except RuntimeError:
        mu = torch.bmm(torch.linalg.pinv(M), b.unsqueeze(2)).squeeze(2)
    return forces + torch.cross(r, mu[idx], dim=1)


class eALIGNNAtomWise(ALIGNNAtomWise):
    """``forward((g, lat))`` or ``((g, lg, lat))`` (``lg`` is ignored, as upstream) -> the reference's result dict."""

    def __init__(self, config: eALIGNNAtomWiseConfig = eALIGNNAtomWiseConfig(name="ealignn_atomwise")):
        if config.gradwise_weight == 0:
            config.calculate_gradient = False
        shared = {k: v for k, v in config.model_dump().items() if k in ALIGNNAtomWiseConfig.model_fields and k != "name"}
        # what the upstream forward hard-wires, expressed in the parent's switches
        shared.update(force_mult_natoms=True, add_reverse_forces=True, lg_on_fly=True, use_cutoff_function=False,
                      multiply_cutoff=False, grad_multiplier=-1, link="identity")
        super().__init__(ALIGNNAtomWiseConfig(name="alignn_atomwise", **shared))
        self.ealignn_config = config  # (``self.config`` is the effective ALIGNNAtomWise schema the shared forward reads)

    def _filtered_batch(self, g) -> GraphBatch:
        if isinstance(g, GraphBatch):
            return g
        if isinstance(g, (tuple, list)) and isinstance(g[0], GraphBatch):
            return g[0]
        gg, lat = g[0], g[-1]
        # nothing is cached on the graph object: which bonds survive the inner cutoff depends on the positions, so the
        # filtered graph is rebuilt on every forward exactly as upstream does (ealignn_atomwise.py:306-322)
        dev = self.fc.weight.device
        u, v = gg.edges()
        u, v = torch.as_tensor(u).to(dev), torch.as_tensor(v).to(dev)
        bnn = torch.as_tensor(gg.batch_num_nodes()).to(dev, torch.int64)
        cart, r = positions_to_bond_vectors(gg, lat, dev)  # models/utils.py:47-56, 95-126
        keep = torch.linalg.norm(r, dim=1) <= self.ealignn_config.inner_cutoff  # :312-316 drops what is GREATER
        first = torch.cumsum(bnn, 0) - bnn
        batch = GraphBatch.from_coo(u[keep], v[keep], int(bnn.sum()), bnn, atom_features=gg.ndata["atom_features"],
                                    r=r[keep], device=dev, volume=gg.ndata["V"].to(dev)[first], build_line_graph=True)
        batch.cache["cart_coords"] = cart
        if "extra_features" in gg.ndata:
            batch.extra_features = gg.ndata["extra_features"].to(dev, torch.float32).contiguous()
        return batch

    def forward(self, g: Union[Sequence, GraphBatch]):
        b = self._filtered_batch(g)
        res = super().forward(b)
        ec = self.ealignn_config
        if self.config.calculate_gradient and ec.remove_torque and torch.is_tensor(res["grad"]) and res["grad"].dim() == 2:
            cart = b.cache.get("cart_coords")
            if cart is None:
                raise ValueError("remove_torque needs the Cartesian positions: build the batch from (g, lat)")
            bnn = (b.graph_ptr[1:] - b.graph_ptr[:-1]).to(torch.int64)
            res["grad"] = remove_net_torque(cart, res["grad"], bnn)
        return res
