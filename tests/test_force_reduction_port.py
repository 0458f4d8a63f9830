"""The reference's own numerical test (alignn/tests/test_force_reduction.py:131-268), ported 1:1 onto
``alignn_amd.alignn.EdgeGatedGraphConv``: float64 default dtype, the 32-atom JVASP-98225 cluster, ``dgl.radius_graph`` +
``fn.v_sub_u`` + ``SumPooling`` (here: the torch-only DGL stand-in of oracle/shims - test infrastructure), two
EdgeGatedGraphConv layers in train mode with constant node features.

float64 modules run on ``alignn_amd/torch_path.py`` (the HIP kernels are float32), so this file needs
no GPU and belongs to the CPU suite.  The float32-kernel form of the same two checks is
tests/test_gpu_round2.py::test_forces_against_finite_differences_of_the_float64_oracle.
"""
import os
import sys

import pytest
import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle", "shims"))

import dgl  # noqa: E402  (oracle/shims/dgl)
import dgl.function as fn  # noqa: E402
from dgl.nn import SumPooling  # noqa: E402

from alignn_amd.alignn import EdgeGatedGraphConv  # noqa: E402  (where the reference imports alignn.models.alignn)
from tests.helpers import load_golden  # noqa: E402


def _cart_coords():
    z = load_golden("graphs_sample_data.npz")  # the reference's examples/sample_data structures (fractional + lattice)
    i = z["names"].tolist().index("POSCAR-JVASP-98225.vasp")
    return torch.from_numpy(z[f"{i}.frac"] @ z[f"{i}.lat"])  # [32, 3] float64 = Atoms.cart_coords


class SimpleModel(nn.Module):
    """test_force_reduction.py:131-217, line for line in behaviour."""

    def __init__(self, cutoff=8, width=16):
        super().__init__()
        self.cutoff, self.width = cutoff, width
        self.edge_embedding = nn.Linear(1, width)
        self.hidden1 = EdgeGatedGraphConv(width, width)
        self.hidden2 = EdgeGatedGraphConv(width, width)
        self.fc = nn.Linear(width, 1)
        self.readout = SumPooling()

    def forward(self, positions, autograd_forces=False):
        if autograd_forces:
            positions.requires_grad_(True)
        g = dgl.radius_graph(positions, self.cutoff)
        g.ndata["r"] = positions
        g.apply_edges(fn.v_sub_u("r", "r", "bondvec"))
        bondvec = g.edata.pop("bondvec")
        bondlength = torch.norm(bondvec, dim=1).squeeze()
        y = self.edge_embedding(bondlength.unsqueeze(-1))
        g.edata["y"] = y
        x = torch.ones(g.num_nodes(), self.width)
        x, y = self.hidden1(g, x, y)
        x, y = self.hidden2(g, x, y)
        energy = self.fc(x)
        total_energy = torch.squeeze(self.readout(g, energy))
        if not autograd_forces:
            return total_energy
        forces_x = -torch.autograd.grad(total_energy, positions, retain_graph=True)[0]
        pairwise_forces = -torch.autograd.grad(total_energy, bondvec)[0]
        g.edata["pairwise_forces"] = pairwise_forces
        g.update_all(fn.copy_e("pairwise_forces", "m"), fn.sum("m", "forces_ji"))
        rg = dgl.reverse(g, copy_edata=True)
        rg.update_all(fn.copy_e("pairwise_forces", "m"), fn.sum("m", "forces_ij"))
        forces_vec = torch.squeeze(g.ndata["forces_ji"] - rg.ndata["forces_ij"])
        return total_energy, forces_x, forces_vec


@pytest.fixture
def float64_default():
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    yield
    torch.set_default_dtype(prev)


@pytest.mark.filterwarnings("ignore:alignn_amd. torch.float64 tensors run on plain torch")
def test_compare_position_and_displacement_autograd_forces(float64_default):
    torch.manual_seed(0)
    model = SimpleModel(cutoff=5)
    assert model.hidden1.src_gate.weight.dtype == torch.float64
    e, f_x, f_vec = model(_cart_coords(), autograd_forces=True)
    assert torch.isclose(f_x, f_vec).all().item()


@pytest.mark.filterwarnings("ignore:alignn_amd. torch.float64 tensors run on plain torch")
def test_compare_forces_finite_difference(float64_default):
    torch.manual_seed(0)
    model = SimpleModel(cutoff=5)
    x = _cart_coords()
    n = x.shape[0]

    def finite_difference_force(x, i, j, delta=1e-6):
        xa, xb = x.detach().clone(), x.detach().clone()
        xa[i, j] -= delta
        xb[i, j] += delta
        with torch.no_grad():
            return -(model(xb) - model(xa)) / (2 * delta)

    e, f_x, f_vec = model(x, autograd_forces=True)
    f_dx = torch.tensor([[finite_difference_force(x, i, j) for j in range(3)] for i in range(n)])
    # the reference's numerical parameters (those of torch.autograd.gradcheck)
    assert torch.isclose(f_vec, f_dx, atol=1e-05, rtol=0.001).all().item()
    assert torch.isclose(f_x, f_dx, atol=1e-05, rtol=0.001).all().item()
    # (with constant node features the forces themselves are ~1e-5, i.e. of the size of ``atol``: the reference's check
    # is a weak one by construction; the well-conditioned variant with a relative bound is the GPU test named above)
    assert float((f_vec - f_x).abs().max()) < 1e-12


def test_float32_cpu_tensors_still_refused():
    """The torch path is for non-float32 dtypes only: a float32 module on the CPU must raise, not fall back."""
    conv = EdgeGatedGraphConv(16, 16)
    g = dgl.radius_graph(_cart_coords().float(), 5.0)
    with pytest.raises((TypeError, RuntimeError)):
        conv(g, torch.ones(g.num_nodes(), 16), torch.ones(g.num_edges(), 16))


def test_float64_force_field_model_on_the_torch_path_matches_the_float64_oracle():
    """``ALIGNNAtomWise(...).double()`` (alignn/train.py:89-95 sets the dtype globally): energies, forces through
    ``autograd.grad(create_graph=True)``, stresses and the SECOND-order parameter gradients of an energy + force + stress loss on
    plain torch operations (alignn_amd/ff.py per dtype) - against the oracle run in float64 on the same inputs (1e-7: the two
    hold the RBF length scale as a float32-rounded and as a float64 number) and
    against the reference class's own float32 golden (its tolerance).  No GPU, no HIP library."""
    import numpy as np

    from alignn_amd import ALIGNNAtomWise, ALIGNNAtomWiseConfig, GraphBatch
    from oracle import alignn_oracle as O
    from tests.helpers import load_golden, raw_from_golden, rel_err, state_dict_from_golden

    z = load_golden("atomwise_ff_tiny.npz")
    raw = raw_from_golden(z)
    cfg = ALIGNNAtomWiseConfig(name="alignn_atomwise", alignn_layers=2, gcn_layers=2, hidden_features=32,
                               embedding_features=16, atom_input_features=92, calculate_gradient=True,
                               stresswise_weight=0.05)
    model = ALIGNNAtomWise(cfg)
    model.load_state_dict(state_dict_from_golden(z))
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        model = model.double().train()
        batch = GraphBatch.from_raw(raw)
        batch.volume = torch.from_numpy(z["volume"])
        res = model(batch)
        L = torch.nn.functional.l1_loss
        t = lambda k: torch.from_numpy(z[k]).double()  # noqa: E731
        loss = L(res["out"], t("t_energy")) + L(res["grad"], t("t_forces")) + 0.05 * L(res["stresses"], t("t_stress"))
        loss.backward()
        # energies only (calculate_gradient=False) on the same path
        e_only = ALIGNNAtomWise(cfg.model_copy(update={"calculate_gradient": False})).double()
        e_only.load_state_dict(model.state_dict())
        assert rel_err(e_only(batch)["out"].detach(), res["out"].detach()) < 1e-12
    assert res["out"].dtype == torch.float64 and res["grad"].dtype == torch.float64
    # the reference class's float32 run
    assert rel_err(res["out"].detach(), z["pred"]) < 1e-4
    assert rel_err(res["grad"].detach(), z["forces"]) < 2e-4 and rel_err(res["stresses"].detach(), z["stresses"]) < 2e-4
    # the oracle in float64 on the same inputs
    p = {k: (v.detach().double().requires_grad_(v.requires_grad) if v.is_floating_point() else v)
         for k, v in O.as_params(state_dict_from_golden(z)).items()}
    out, forces, stresses = O.alignn_atomwise_forward(
        p, O.TorchGraph(raw), 2, 2, True, calculate_gradient=True, stress=True,
        volume=torch.from_numpy(z["volume"]).double(), batch_num_edges=torch.from_numpy(raw.batch_num_edges))
    inv = batch.g.inv  # our forces are per atom (no permutation); stresses per crystal
    assert rel_err(res["out"].detach(), out.detach()) < 1e-8
    assert rel_err(res["grad"].detach(), forces.detach()) < 1e-7 and rel_err(res["stresses"].detach(), stresses.detach()) < 1e-7
    oloss = L(out, t("t_energy")) + L(forces, t("t_forces")) + 0.05 * L(stresses, t("t_stress"))
    oloss.backward()
    assert abs(loss.item() - oloss.item()) < 1e-8
    nograd = set(z["nograd"].tolist())
    gmax = max(float(v.grad.abs().max()) for k, v in p.items() if v.grad is not None)
    n = 0
    for k, q in model.named_parameters():
        if k in nograd or p[k].grad is None:
            continue
        assert float((q.grad - p[k].grad).abs().max()) < 1e-6 * gmax, k
        n += 1
    assert n > 40 and inv is not None


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_sixteen_bit_modules_run_on_the_torch_path(dtype):
    """``model.to(torch.bfloat16)`` / ``.half()``: forward and backward of both model classes on plain torch operations,
    close to the float64 run at the precision of the type (no HIP library, no GPU)."""
    import copy
    import warnings

    from alignn_amd import ALIGNN, ALIGNNAtomWise, ALIGNNAtomWiseConfig, ALIGNNConfig, GraphBatch
    from alignn_amd.synthetic import make_batch

    raw = make_batch(3, 10, seed0=4)
    batch = GraphBatch.from_raw(raw)
    batch.volume = torch.ones(3)
    tol = 4e-2 if dtype == torch.bfloat16 else 1e-2
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        torch.manual_seed(0)
        m = ALIGNN(ALIGNNConfig(name="alignn", alignn_layers=1, gcn_layers=1, hidden_features=32, embedding_features=16)).train()
        ref = copy.deepcopy(m).double()(batch).detach()
        out = m.to(dtype)(batch)
        assert out.dtype == dtype
        out.float().sum().backward()
        grads = [p.grad for p in m.parameters() if p.grad is not None]  # (the last edge output is dead: its norm has none)
        assert len(grads) > 30 and all(bool(torch.isfinite(g).all()) for g in grads)
        assert float((out.double() - ref).abs().max()) < tol * float(ref.abs().max())
        cfg = ALIGNNAtomWiseConfig(name="alignn_atomwise", alignn_layers=1, gcn_layers=1, hidden_features=32,
                                   embedding_features=16, atom_input_features=92, calculate_gradient=True,
                                   stresswise_weight=0.05)
        a = ALIGNNAtomWise(cfg).train()
        r64 = copy.deepcopy(a).double()(batch)
        res = a.to(dtype)(batch)
        assert res["grad"].dtype == dtype and res["grad"].shape == (raw.num_nodes, 3)
        (res["out"].float().sum() + res["grad"].float().abs().sum()).backward()
        assert float((res["out"].double() - r64["out"]).abs().max()) < tol * float(r64["out"].abs().max())
        assert float((res["grad"].double() - r64["grad"]).abs().max()) < 0.15 * float(r64["grad"].abs().max())
