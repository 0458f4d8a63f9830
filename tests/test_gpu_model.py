"""End-to-end parity of the HIP ALIGNN against (a) the golden vectors generated from the reference's
own model code and (b) the CPU oracle on seeded synthetic batches.  Tolerance: 1e-4 relative to the
tensor scale (BASELINE.json north_star), gradients 1e-3 of the largest gradient."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from alignn_amd import ALIGNN, ALIGNNConfig, EdgeGatedGraphConv, GraphBatch  # noqa: E402
from alignn_amd.synthetic import make_batch  # noqa: E402
from oracle import alignn_oracle as O  # noqa: E402
from tests.helpers import load_golden, raw_from_golden, rel_err, sample, state_dict_from_golden  # noqa: E402

DEV = "cuda"


def _model_from(sd, **cfg):
    m = ALIGNN(ALIGNNConfig(name="alignn", **cfg))
    m.load_state_dict(sd)
    return m.to(DEV)


def test_golden_tiny_train():
    z = load_golden("alignn_tiny_train.npz")
    model = _model_from(state_dict_from_golden(z), alignn_layers=2, gcn_layers=2, hidden_features=32, embedding_features=16).train()
    batch = GraphBatch.from_raw(raw_from_golden(z), device=DEV)
    pred = model(batch)
    assert rel_err(pred, z["pred"]) < 1e-4
    loss = torch.nn.functional.l1_loss(pred, torch.from_numpy(z["target"]).to(DEV))
    assert abs(loss.item() - float(z["loss"])) < 1e-5
    loss.backward()
    nograd = set(z["nograd"].tolist())
    gfloor = 1e-2 * max(float(np.abs(v).max()) for k, v in z.items() if k.startswith("grad."))
    n = 0
    for k, p in model.named_parameters():
        if k in nograd:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
        else:
            assert rel_err(p.grad, z["grad." + k], floor=gfloor) < 1e-3, k
            n += 1
    assert n > 50
    sd = model.state_dict()
    for k, v in z.items():
        if k.startswith("sd_after."):
            assert rel_err(sd[k[9:]], v, floor=1e-3) < 1e-4, k


def test_golden_tiny_eval():
    z = load_golden("alignn_tiny_eval.npz")
    model = _model_from(state_dict_from_golden(z), alignn_layers=2, gcn_layers=2, hidden_features=32, embedding_features=16).eval()
    with torch.no_grad():
        pred = model(GraphBatch.from_raw(raw_from_golden(z), device=DEV))
    assert rel_err(pred, z["pred"]) < 1e-4


def test_golden_default_config_train():
    z = load_golden("alignn_default_train.npz")
    model = _model_from(O.init_state_dict(seed=0)).train()
    pred = model(GraphBatch.from_raw(raw_from_golden(z), device=DEV))
    assert rel_err(pred, z["pred"]) < 1e-4
    loss = torch.nn.functional.l1_loss(pred, torch.from_numpy(z["target"]).to(DEV))
    loss.backward()
    nograd = set(z["nograd"].tolist())
    gfloor = 1e-2 * max(float(np.abs(v[:-3]).max()) for k, v in z.items() if k.startswith("grad."))
    for k, p in model.named_parameters():
        if k not in nograd:
            assert rel_err(sample(p.grad)[:-3], z["grad." + k][:-3], floor=gfloor) < 1e-3, k
    sd = model.state_dict()
    for k, v in z.items():
        if k.startswith("sd_after."):
            assert rel_err(sd[k[9:]], v, floor=1e-3) < 1e-4, k


def test_golden_conv_dgl_like_edge_order():
    """Stand-alone EdgeGatedGraphConv on a DGL-like graph (caller's edge order, isolated node, multi-edges)."""
    z = load_golden("conv_f64.npz")

    class G:  # the duck-typed surface the conv touches
        def __init__(self, u, v, n):
            self._u, self._v, self._n = u, v, n

        def edges(self):
            return self._u, self._v

        def num_nodes(self):
            return self._n

    conv = EdgeGatedGraphConv(16, 16)
    conv.load_state_dict({k[3:]: torch.from_numpy(v).float() if v.dtype == np.float64 else torch.from_numpy(v) for k, v in z.items() if k.startswith("sd.")})
    conv = conv.to(DEV).train()
    x = torch.from_numpy(z["x"]).float().to(DEV).requires_grad_(True)
    y = torch.from_numpy(z["y"]).float().to(DEV).requires_grad_(True)
    g = G(torch.from_numpy(z["u"]), torch.from_numpy(z["v"]), 9)
    xo, yo = conv(g, x, y)
    assert rel_err(xo, z["x_out"]) < 2e-5 and rel_err(yo, z["y_out"]) < 2e-5
    ((xo * torch.from_numpy(z["wx"]).float().to(DEV)).sum() + (yo * torch.from_numpy(z["wy"]).float().to(DEV)).sum()).backward()
    assert rel_err(x.grad, z["gx"]) < 1e-4 and rel_err(y.grad, z["gy"]) < 1e-4
    gfloor = 1e-2 * max(float(np.abs(v).max()) for k, v in z.items() if k.startswith("grad."))
    for k, p in conv.named_parameters():
        assert rel_err(p.grad, z["grad." + k], floor=gfloor) < 2e-4, k


@pytest.mark.parametrize("kind,B,n", [("crystal", 8, 24), ("molecule", 16, (9, 27))])
def test_synthetic_batch_vs_oracle(kind, B, n):
    raw = make_batch(B, n, seed0=4321, kind=kind)
    torch.manual_seed(3)
    model = ALIGNN(ALIGNNConfig(name="alignn", alignn_layers=2, gcn_layers=2, hidden_features=256))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(DEV).train()
    target = torch.linspace(-2, 2, B)
    pred = model(GraphBatch.from_raw(raw, device=DEV))
    torch.nn.functional.l1_loss(pred, target.to(DEV)).backward()
    p = O.as_params(sd)
    stats = {}
    opred = O.alignn_forward(p, O.TorchGraph(raw), 2, 2, True, stats)
    torch.nn.functional.l1_loss(opred, target).backward()
    assert rel_err(pred, opred) < 1e-4
    gfloor = 1e-2 * max(float(t.grad.abs().max()) for t in p.values() if t.grad is not None)
    for k, q in model.named_parameters():
        if p[k].grad is not None:
            assert rel_err(q.grad, p[k].grad, floor=gfloor) < 1e-3, k
    after = O.running_stats_after_step(p, stats)
    sdn = model.state_dict()
    for k, v in after.items():
        assert rel_err(sdn[k], v, floor=1e-3) < 1e-4, k
    assert int(sdn["atom_embedding.layer.1.num_batches_tracked"]) == 1


def test_dgl_like_tuple_input_and_edge_order_invariance():
    """forward((g, lg, lat)) on duck-typed DGL graphs; shuffling the caller's edge order must not change
    the prediction beyond fp32 summation noise (size-independent property)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "shims"))
    import dgl  # the shim: only used here as a DGL-shaped container

    raw = make_batch(4, 16, seed0=99)
    torch.manual_seed(1)
    model = ALIGNN(ALIGNNConfig(name="alignn", alignn_layers=1, gcn_layers=1, hidden_features=64)).to(DEV).eval()

    def graphs(perm_e, perm_t):
        inv_e = np.empty_like(perm_e)
        inv_e[perm_e] = np.arange(perm_e.size)
        g = dgl.graph((torch.from_numpy(raw.u[perm_e]), torch.from_numpy(raw.v[perm_e])), num_nodes=raw.num_nodes)
        g._bnn = torch.from_numpy(raw.batch_num_nodes)
        g.ndata["atom_features"] = torch.from_numpy(raw.atom_features)
        g.edata["r"] = torch.from_numpy(raw.r[perm_e])
        lg = dgl.graph((torch.from_numpy(inv_e[raw.lg_u][perm_t]), torch.from_numpy(inv_e[raw.lg_v][perm_t])), num_nodes=raw.num_edges)
        lg.edata["h"] = torch.from_numpy(raw.h[perm_t])
        return g, lg, torch.from_numpy(raw.lattice)

    with torch.no_grad():
        a = model(graphs(np.arange(raw.num_edges), np.arange(raw.num_triplets)))
        rng = np.random.default_rng(0)
        b = model(list(graphs(rng.permutation(raw.num_edges), rng.permutation(raw.num_triplets))))
    assert a.shape == (4,)
    assert rel_err(a, b) < 1e-5


def test_bare_graph_call_form_with_no_alignn_layers_against_the_reference_class():
    """``alignn_layers == 0``: ``forward(g)`` on a BARE graph (alignn/models/alignn.py:290-305; SURVEY 8(b) names the call
    form) - no line graph, the angle embedding unused.  Golden from the reference's own class (oracle/make_golden.py
    ``case_bare_graph``): training step (prediction, loss, every gradient, running statistics, the three convolutions'
    outputs) and the eval prediction; through the DGL-shaped container AND a GraphBatch without a line graph."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "shims"))
    import dgl  # the shim: only used here as a DGL-shaped container

    z = load_golden("alignn_bare_graph.npz")
    raw = raw_from_golden(z)
    g = dgl.graph((torch.from_numpy(raw.u), torch.from_numpy(raw.v)), num_nodes=raw.num_nodes)
    g._bnn = torch.from_numpy(raw.batch_num_nodes)
    g._bne = torch.from_numpy(raw.batch_num_edges)
    g.ndata["atom_features"] = torch.from_numpy(raw.atom_features)
    g.edata["r"] = torch.from_numpy(raw.r)
    cfg = dict(alignn_layers=0, gcn_layers=3, hidden_features=64, embedding_features=32)
    model = _model_from(state_dict_from_golden(z), **cfg).train()
    acts = {}
    for name, mod in model.named_modules():
        if isinstance(mod, EdgeGatedGraphConv):
            mod.register_forward_hook(lambda _m, _i, out, name=name: acts.__setitem__(name, out))
    pred = model(g)  # the bare graph
    assert pred.shape == (4,) and rel_err(pred, z["pred"]) < 1e-4
    loss = torch.nn.functional.l1_loss(pred, torch.from_numpy(z["target"]).to(DEV))
    assert abs(loss.item() - float(z["loss"])) < 1e-5
    loss.backward()
    topo = g._alignn_amd_topology[0]
    inv = topo.g.inv.long()  # caller edge -> canonical row
    for name, (x, y) in acts.items():
        assert rel_err(x, z[f"act.{name}.x_out"]) < 1e-4, name
        if y is not None:  # (the last layer's bond features are dead and not materialised)
            assert rel_err(y[inv], z[f"act.{name}.y_out"]) < 1e-4, name
    assert len(acts) == 3
    nograd = set(z["nograd"].tolist())
    # (the unused angle embedding's eight parameters and the norm of the last layer's dead bond output)
    assert len(nograd) == 10 and all(k.startswith(("angle_embedding", "gcn_layers.2.bn_edges")) for k in nograd)
    gfloor = 1e-2 * max(float(np.abs(v).max()) for k, v in z.items() if k.startswith("grad."))
    n = 0
    for k, p in model.named_parameters():
        if k in nograd:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
        else:
            assert rel_err(p.grad, z["grad." + k], floor=gfloor) < 1e-3, k
            n += 1
    assert n > 30
    sd = model.state_dict()
    for k, v in z.items():
        if k.startswith("sd_after."):
            assert rel_err(sd[k[9:]], v, floor=1e-3) < 1e-4, k
    # eval prediction with the statistics that step left behind (the golden's eval call follows its training step too);
    # a tuple / list whose first entry is the graph is accepted as well when there are no ALIGNN layers
    model.eval()
    with torch.no_grad():
        assert rel_err(model(g), z["pred_eval"]) < 1e-4
        assert rel_err(model([g]), z["pred_eval"]) < 1e-4
    # ... and a prebuilt GraphBatch that carries no line graph
    fresh = _model_from(state_dict_from_golden(z), **cfg).train()
    t = torch.from_numpy
    b = GraphBatch.from_coo(t(raw.u), t(raw.v), raw.num_nodes, t(raw.batch_num_nodes), atom_features=t(raw.atom_features),
                            r=t(raw.r), device=DEV)
    assert b.lg is None and b.h is None
    assert rel_err(fresh(b), z["pred"]) < 1e-4


def test_single_graph_squeezes_to_scalar():
    raw = make_batch(1, 8, seed0=5)
    model = ALIGNN(ALIGNNConfig(name="alignn", alignn_layers=1, gcn_layers=1, hidden_features=32, embedding_features=16)).to(DEV).eval()
    with torch.no_grad():
        out = model(GraphBatch.from_raw(raw, device=DEV))
    assert out.dim() == 0


def test_product_path_refuses_cpu_tensors():
    model = ALIGNN(ALIGNNConfig(name="alignn", alignn_layers=1, gcn_layers=0, hidden_features=32, embedding_features=16))
    raw = make_batch(1, 8, seed0=5)
    with pytest.raises((TypeError, RuntimeError)):
        model(GraphBatch.from_raw(raw))  # CPU parameters + CPU batch: no fallback, must raise


@pytest.mark.parametrize("B", [64])
def test_full_size_properties(B):
    """BASELINE config 2 size (B=64 x 60 atoms, H=256, 4+4): properties that need no oracle run -
    bit-reproducibility of a training step and permutation-equivariance over the graphs of a batch."""
    raw = make_batch(B, 60)
    torch.manual_seed(0)
    model = ALIGNN(ALIGNNConfig(name="alignn")).to(DEV).train()
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    batch = GraphBatch.from_raw(raw, device=DEV)
    target = torch.randn(B, generator=torch.Generator().manual_seed(1)).to(DEV)
    outs = []
    for _ in range(2):
        model.load_state_dict(sd0)
        model.zero_grad(set_to_none=True)
        pred = model(batch)
        torch.nn.functional.l1_loss(pred, target).backward()
        outs.append((pred.detach().clone(), model.alignn_layers[0].edge_update.edge_gate.weight.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.isfinite(outs[0][0]).all()
    # eval mode: predictions of graphs do not depend on what else is in the batch
    model.eval()
    with torch.no_grad():
        full = model(batch)
        from alignn_amd.synthetic import batch_raw, _one
        sub = batch_raw([_one(60, 1234 + i, "crystal", 92) for i in (5, 17)])
        part = model(GraphBatch.from_raw(sub, device=DEV))
    assert rel_err(part, full[[5, 17]]) < 1e-4


# ---------------------------------------------------------------------------------------------
# LayerNorm flavour (ALIGNNAtomWise)
# ---------------------------------------------------------------------------------------------
def test_golden_atomwise_layernorm_train():
    from alignn_amd import ALIGNNAtomWise, ALIGNNAtomWiseConfig

    z = load_golden("atomwise_tiny_train.npz")
    cfg = ALIGNNAtomWiseConfig(name="alignn_atomwise", alignn_layers=2, gcn_layers=2, hidden_features=32,
                               embedding_features=16, atom_input_features=92, calculate_gradient=False)
    model = ALIGNNAtomWise(cfg)
    model.load_state_dict(state_dict_from_golden(z))
    model = model.to(DEV).train()
    raw = raw_from_golden(z)
    raw.h = raw.h * 0.5  # loader cosines are wrong on purpose: lg_on_fly must recompute them from r
    res = model(GraphBatch.from_raw(raw, device=DEV))
    assert set(res) == {"out", "additional", "grad", "stresses", "atomwise_pred"}
    assert rel_err(res["out"], z["pred"]) < 1e-4
    loss = torch.nn.functional.l1_loss(res["out"], torch.from_numpy(z["target"]).to(DEV))
    assert abs(loss.item() - float(z["loss"])) < 1e-5
    loss.backward()
    nograd = set(z["nograd"].tolist())
    gfloor = 1e-2 * max(float(np.abs(v).max()) for k, v in z.items() if k.startswith("grad."))
    n = 0
    for k, p in model.named_parameters():
        if k in nograd:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
        else:
            assert rel_err(p.grad, z["grad." + k], floor=gfloor) < 1e-3, k
            n += 1
    assert n > 40


def test_atomwise_vs_oracle_wide_and_guards():
    from alignn_amd import ALIGNNAtomWise, ALIGNNAtomWiseConfig

    raw = make_batch(6, 20, seed0=555)
    torch.manual_seed(4)
    cfg = ALIGNNAtomWiseConfig(name="alignn_atomwise", alignn_layers=2, gcn_layers=1, hidden_features=256,
                               atom_input_features=92, calculate_gradient=False)
    model = ALIGNNAtomWise(cfg)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(DEV).train()
    target = torch.linspace(-1, 1, 6)
    out = model(GraphBatch.from_raw(raw, device=DEV))["out"]
    torch.nn.functional.l1_loss(out, target.to(DEV)).backward()
    p = O.as_params(sd)
    oout = O.alignn_atomwise_forward(p, O.TorchGraph(raw), 2, 1, True)
    torch.nn.functional.l1_loss(oout, target).backward()
    assert rel_err(out, oout) < 1e-4
    gfloor = 1e-2 * max(float(t.grad.abs().max()) for t in p.values() if t.grad is not None)
    for k, q in model.named_parameters():
        if p[k].grad is not None:
            assert rel_err(q.grad, p[k].grad, floor=gfloor) < 1e-3, k
    # the descriptor head needs its input: a batch without extra_features must refuse loudly
    fm = ALIGNNAtomWise(ALIGNNAtomWiseConfig(name="alignn_atomwise", atom_input_features=92, extra_features=3)).to(DEV).train()
    with pytest.raises(ValueError):
        fm(GraphBatch.from_raw(raw, device=DEV))
    # default config (calculate_gradient=True): forces come back with one row per atom
    fd = ALIGNNAtomWise(ALIGNNAtomWiseConfig(name="alignn_atomwise", atom_input_features=92)).to(DEV)
    res = fd(GraphBatch.from_raw(raw, device=DEV))
    assert res["grad"].shape == (raw.num_nodes, 3) and torch.isfinite(res["grad"]).all()


@pytest.mark.parametrize("rows,F", [(5, 16), (1000, 256), (333, 64), (77, 512), (9, 1024)])
def test_layernorm_silu_kernels(rows, F):
    from alignn_amd import ops

    g = torch.Generator().manual_seed(rows)
    x = (torch.randn(rows, F, generator=g) * 2 + 0.5).to(DEV)
    res = torch.randn(rows, F, generator=g).to(DEV)
    gamma = (1 + 0.1 * torch.randn(F, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(F, generator=g)).to(DEV)
    gy = torch.randn(rows, F, generator=g).to(DEV)
    y, st = ops._ln_silu_fwd(x, res, gamma, beta)
    gx = torch.empty_like(x)
    red = ops._ln_silu_bwd(gy, x, gamma, beta, st, gx)
    xd, gd, bd = (t.double().cpu().requires_grad_(True) for t in (x, gamma, beta))
    yd = res.double().cpu() + torch.nn.functional.silu(torch.nn.functional.layer_norm(xd, (F,), gd, bd, 1e-5))
    yd.backward(gy.double().cpu())
    assert rel_err(y, yd) < 1e-5
    assert rel_err(gx, xd.grad) < 1e-4
    assert rel_err(red[1], gd.grad) < 1e-4 and rel_err(red[0], bd.grad) < 1e-4


def test_golden_atomwise_force_stress_head():
    """ALIGNN-FF head on the GPU: energies, forces, stresses and the second-order parameter gradients of an
    energy+force+stress loss against the reference's own class (golden), DGL-like tuple input with ndata['V']."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "shims"))
    import dgl  # shim: DGL-shaped container only

    from alignn_amd import ALIGNNAtomWise, ALIGNNAtomWiseConfig

    z = load_golden("atomwise_ff_tiny.npz")
    raw = raw_from_golden(z)
    cfg = ALIGNNAtomWiseConfig(name="alignn_atomwise", alignn_layers=2, gcn_layers=2, hidden_features=32,
                               embedding_features=16, atom_input_features=92, calculate_gradient=True,
                               stresswise_weight=0.05)
    model = ALIGNNAtomWise(cfg)
    model.load_state_dict(state_dict_from_golden(z))
    model = model.to(DEV).train()
    g = dgl.graph((torch.from_numpy(raw.u), torch.from_numpy(raw.v)), num_nodes=raw.num_nodes)
    g._bnn, g._bne = torch.from_numpy(raw.batch_num_nodes), torch.from_numpy(raw.batch_num_edges)
    g.ndata["atom_features"] = torch.from_numpy(raw.atom_features)
    g.edata["r"] = torch.from_numpy(raw.r)
    g.ndata["V"] = torch.from_numpy(np.repeat(z["volume"], raw.batch_num_nodes))
    lg = dgl.graph((torch.from_numpy(raw.lg_u), torch.from_numpy(raw.lg_v)), num_nodes=raw.num_edges)
    lg.edata["h"] = torch.from_numpy(raw.h)
    res = model([g, lg, torch.from_numpy(raw.lattice)])
    assert rel_err(res["out"], z["pred"]) < 1e-4
    assert res["grad"].shape == (raw.num_nodes, 3) and rel_err(res["grad"], z["forces"]) < 2e-4
    assert res["stresses"].shape == (2, 3, 3) and rel_err(res["stresses"], z["stresses"]) < 2e-4
    L = torch.nn.functional.l1_loss
    t = lambda k: torch.from_numpy(z[k]).to(DEV)  # noqa: E731
    loss = L(res["out"], t("t_energy")) + L(res["grad"], t("t_forces")) + 0.05 * L(res["stresses"], t("t_stress"))
    assert abs(loss.item() - float(z["loss"])) < 1e-4
    loss.backward()
    nograd = set(z["nograd"].tolist())
    gfloor = 1e-2 * max(float(np.abs(v).max()) for k, v in z.items() if k.startswith("grad."))
    n = 0
    for k, p in model.named_parameters():
        if k in nograd:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
        else:
            assert rel_err(p.grad, z["grad." + k], floor=gfloor) < 2e-3, k
            n += 1
    assert n > 40


def test_golden_atomwise_forces_eval_mode_fused_path():
    """eval(): forces need only the FIRST derivative -> the fused kernels with their hand-written backward plus the
    geometry derivatives (alignn_rbf_bwd / norm3_bwd / bond_cosine_bwd).  Same energies, forces and stresses as the
    reference's class (golden; LayerNorm models have no train/eval difference) and as the composed training path;
    no parameter gradient is produced."""
    from alignn_amd import ALIGNNAtomWise, ALIGNNAtomWiseConfig

    z = load_golden("atomwise_ff_tiny.npz")
    raw = raw_from_golden(z)
    cfg = ALIGNNAtomWiseConfig(name="alignn_atomwise", alignn_layers=2, gcn_layers=2, hidden_features=32,
                               embedding_features=16, atom_input_features=92, calculate_gradient=True,
                               stresswise_weight=0.05)
    model = ALIGNNAtomWise(cfg)
    model.load_state_dict(state_dict_from_golden(z))
    model = model.to(DEV)
    batch = GraphBatch.from_raw(raw, device=DEV)
    batch.volume = torch.from_numpy(z["volume"]).to(DEV).float()
    res = model.eval()(batch)
    assert rel_err(res["out"], z["pred"]) < 1e-4
    assert res["grad"].shape == (raw.num_nodes, 3) and rel_err(res["grad"], z["forces"]) < 2e-4
    assert res["stresses"].shape == (2, 3, 3) and rel_err(res["stresses"], z["stresses"]) < 2e-4
    assert not res["out"].requires_grad and not res["grad"].requires_grad
    assert all(p.grad is None for p in model.parameters())
    comp = model.train()(batch)
    assert rel_err(res["grad"], comp["grad"].detach()) < 1e-4 and rel_err(res["stresses"], comp["stresses"].detach()) < 1e-4
    with torch.no_grad():  # no autograd at all: energies only, as the reference's calculators do for energy-only calls
        model.eval()
        e_only = ALIGNNAtomWise(cfg.model_copy(update={"calculate_gradient": False})).to(DEV).eval()
        e_only.load_state_dict(model.state_dict())
        assert rel_err(e_only(batch)["out"], z["pred"]) < 1e-4


def test_geometry_derivative_kernels_match_torch_autograd():
    from alignn_amd import ops

    raw = make_batch(2, 9, seed0=91)
    b = GraphBatch.from_raw(raw, device=DEV)
    r = b.r.clone().requires_grad_(True)
    # bond length
    w = torch.randn(r.shape[0], device=DEV)
    (ops.bond_length(r) * w).sum().backward()
    r2 = b.r.clone().requires_grad_(True)
    (torch.norm(r2, dim=1) * w).sum().backward()
    assert rel_err(r.grad, r2.grad) < 1e-5
    # cosines (incl. the clamp) on the canonical line graph
    r = b.r.clone().requires_grad_(True)
    wt = torch.randn(b.lg.n_edges, device=DEV)
    h = ops.bond_cosines(r, b.lg)
    (h * wt).sum().backward()
    r2 = b.r.clone().requires_grad_(True)
    a, c = -r2[b.lg.src.long()], r2[b.lg.dst.long()]
    h2 = torch.clamp((a * c).sum(1) / (a.norm(dim=1) * c.norm(dim=1)), -1, 1)
    (h2 * wt).sum().backward()
    assert rel_err(h, h2) < 1e-6 and rel_err(r.grad, r2.grad) < 2e-5
    # rbf
    d = (torch.rand(1000, device=DEV) * 8).requires_grad_(True)
    centers = torch.linspace(0, 8, 80, device=DEV)
    G = torch.randn(1000, 80, device=DEV)
    (ops.rbf_expand(d, centers, 9.875) * G).sum().backward()
    d2 = d.detach().clone().requires_grad_(True)
    (torch.exp(-9.875 * (d2.unsqueeze(1) - centers) ** 2) * G).sum().backward()
    assert rel_err(d.grad, d2.grad) < 2e-5


def test_force_reduction_self_consistency():
    """Port of the reference's own hot-path test (alignn/tests/test_force_reduction.py:212-229): forces from
    displacement autograd reduced over in- minus out-edges equal forces from position autograd."""
    from alignn_amd import ff
    from alignn_amd.alignn_atomwise import EdgeGatedGraphConv as LNConv
    from alignn_amd.graph import build_csr

    torch.manual_seed(0)
    n, width = 32, 16
    pos = (torch.rand(n, 3) * 6.0).to(DEV).requires_grad_(True)
    d = torch.cdist(pos.detach(), pos.detach())
    mask = (d <= 3.5) & ~torch.eye(n, dtype=torch.bool, device=DEV)
    v_, u_ = torch.nonzero(mask, as_tuple=True)  # u -> v
    csr = build_csr(u_, v_, n)
    conv1, conv2 = LNConv(width, width).to(DEV), LNConv(width, width).to(DEV)
    emb = torch.nn.Linear(1, width).to(DEV)
    fc = torch.nn.Linear(width, 1).to(DEV)
    # bond vectors in canonical slot order, as a function of positions
    bondvec = ff.gather(pos, ff.by_dst(csr)) - ff.gather(pos, ff.by_src(csr))
    bondlength = torch.norm(bondvec, dim=1)
    y = ff.linear(bondlength.unsqueeze(-1), emb)
    x = torch.ones(n, width, device=DEV)
    x, y = ff.edge_gated_conv(csr, x, y, conv1)
    x, y = ff.edge_gated_conv(csr, x, y, conv2)
    energy = ff.linear(x, fc).sum()
    f_x = -torch.autograd.grad(energy, pos, retain_graph=True)[0]
    pf = -torch.autograd.grad(energy, bondvec)[0]
    f_vec = ff.pair_force_reduce(pf, csr)
    assert rel_err(f_vec, f_x) < 1e-4


def test_ragged_batch_with_one_atom_cells():
    """Edge case of the reference's builder: 1- and 2-atom cells are all self-image multi-edges (a 1-atom cell is
    26 self-loops, so its line graph excludes only e1 == e2); mix them with a normal cell, train mode."""
    from alignn_amd.synthetic import _one, batch_raw

    raw = batch_raw([_one(n, 50 + i, "crystal", 92) for i, n in enumerate((1, 2, 7, 1))])
    assert raw.num_nodes == 11 and int((raw.u == raw.v).sum()) > 50
    torch.manual_seed(8)
    model = ALIGNN(ALIGNNConfig(name="alignn", alignn_layers=2, gcn_layers=1, hidden_features=64))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(DEV).train()
    target = torch.tensor([0.5, -0.5, 1.0, 0.0])
    pred = model(GraphBatch.from_raw(raw, device=DEV))
    torch.nn.functional.l1_loss(pred, target.to(DEV)).backward()
    p = O.as_params(sd)
    opred = O.alignn_forward(p, O.TorchGraph(raw), 2, 1, True)
    torch.nn.functional.l1_loss(opred, target).backward()
    assert rel_err(pred, opred) < 1e-4
    gfloor = 1e-2 * max(float(t.grad.abs().max()) for t in p.values() if t.grad is not None)
    for k, q in model.named_parameters():
        if p[k].grad is not None:
            assert rel_err(q.grad, p[k].grad, floor=gfloor) < 1e-3, k


def test_golden_extra_features_and_classification_heads():
    """ALIGNNConfig.extra_features != 0 (descriptor head: extra_feature_embedding, fc1, fc2, fc3) and
    classification=True (LogSoftmax over num_classes) against the reference's own class: predictions, loss and every
    parameter gradient; state_dict keys load unchanged."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "shims"))
    import dgl  # shim: DGL-shaped container only

    z = load_golden("alignn_extra_class.npz")
    raw = raw_from_golden(z)
    for tag, kw in (("x", dict(extra_features=3)), ("c", dict(classification=True, num_classes=3))):
        cfg = ALIGNNConfig(name="alignn", alignn_layers=1, gcn_layers=1, hidden_features=32, embedding_features=16, **kw)
        model = ALIGNN(cfg)
        sd = {k[len(tag) + 4:]: torch.from_numpy(np.asarray(v)) for k, v in z.items() if k.startswith(tag + ".sd.")}
        model.load_state_dict(sd)  # strict: same keys as the reference
        model = model.to(DEV).train()
        g = dgl.graph((torch.from_numpy(raw.u), torch.from_numpy(raw.v)), num_nodes=raw.num_nodes)
        g._bnn, g._bne = torch.from_numpy(raw.batch_num_nodes), torch.from_numpy(raw.batch_num_edges)
        g.ndata["atom_features"] = torch.from_numpy(raw.atom_features)
        g.edata["r"] = torch.from_numpy(raw.r)
        if tag == "x":
            g.ndata["extra_features"] = torch.from_numpy(z["extra_features"])
        lg = dgl.graph((torch.from_numpy(raw.lg_u), torch.from_numpy(raw.lg_v)), num_nodes=raw.num_edges)
        lg.edata["h"] = torch.from_numpy(raw.h)
        pred = model((g, lg, torch.from_numpy(raw.lattice)))
        assert pred.shape == z[tag + ".pred"].shape and rel_err(pred, z[tag + ".pred"]) < 1e-4
        if tag == "x":
            loss = torch.nn.functional.l1_loss(pred, torch.from_numpy(z["x.target"]).to(DEV))
        else:
            loss = torch.nn.functional.nll_loss(pred, torch.from_numpy(z["c.target"]).to(DEV))
        assert abs(loss.item() - float(z[tag + ".loss"])) < 1e-4
        loss.backward()
        nograd = set(z[tag + ".nograd"].tolist())
        gfloor = 1e-2 * max(float(np.abs(v).max()) for k, v in z.items() if k.startswith(tag + ".grad."))
        n = 0
        for k, p in model.named_parameters():
            if k in nograd:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            else:
                assert rel_err(p.grad, z[tag + ".grad." + k], floor=gfloor) < 1e-3, (tag, k)
                n += 1
        assert n > 20


def test_golden_atomwise_extra_features_head():
    """ALIGNNAtomWiseConfig.extra_features != 0 against the reference's class: [B,1] output (fc3 is not squeezed
    upstream), loss and all parameter gradients."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "shims"))
    import dgl  # shim: DGL-shaped container only

    from alignn_amd import ALIGNNAtomWise, ALIGNNAtomWiseConfig

    z = load_golden("atomwise_extra.npz")
    raw = raw_from_golden(z)
    cfg = ALIGNNAtomWiseConfig(name="alignn_atomwise", alignn_layers=1, gcn_layers=1, hidden_features=32,
                               embedding_features=16, atom_input_features=92, calculate_gradient=False, extra_features=3)
    model = ALIGNNAtomWise(cfg)
    model.load_state_dict(state_dict_from_golden(z))
    model = model.to(DEV).train()
    g = dgl.graph((torch.from_numpy(raw.u), torch.from_numpy(raw.v)), num_nodes=raw.num_nodes)
    g._bnn, g._bne = torch.from_numpy(raw.batch_num_nodes), torch.from_numpy(raw.batch_num_edges)
    g.ndata["atom_features"] = torch.from_numpy(raw.atom_features)
    g.ndata["extra_features"] = torch.from_numpy(z["extra_features"])
    g.edata["r"] = torch.from_numpy(raw.r)
    lg = dgl.graph((torch.from_numpy(raw.lg_u), torch.from_numpy(raw.lg_v)), num_nodes=raw.num_edges)
    lg.edata["h"] = torch.from_numpy(raw.h)
    res = model([g, lg, torch.from_numpy(raw.lattice)])
    assert res["out"].shape == z["pred"].shape == (3, 1) and rel_err(res["out"], z["pred"]) < 1e-4
    loss = torch.nn.functional.l1_loss(res["out"], torch.from_numpy(z["target"]).to(DEV))
    assert abs(loss.item() - float(z["loss"])) < 1e-4
    loss.backward()
    nograd = set(z["nograd"].tolist())
    gfloor = 1e-2 * max(float(np.abs(v).max()) for k, v in z.items() if k.startswith("grad."))
    n = 0
    for k, p in model.named_parameters():
        if k in nograd:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
        else:
            assert rel_err(p.grad, z["grad." + k], floor=gfloor) < 1e-3, k
            n += 1
    assert n > 20


def test_golden_atomwise_position_based_branches():
    """ALIGNNAtomWise(include_pos_deriv=True) - forces from d/d(cart_coords) - and (batch_stress=False) - one virial over
    all bonds with position-derived bond vectors and the 1/2 factor - against the reference's own class: energies,
    forces, stress, loss, second-order parameter gradients; eval() (fused kernels) agrees."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "shims"))
    import dgl  # shim: DGL-shaped container only

    from alignn_amd import ALIGNNAtomWise, ALIGNNAtomWiseConfig

    z = load_golden("atomwise_position_branches.npz")
    n = int(z["in.batch_num_nodes"].sum())

    def graphs():
        g = dgl.graph((torch.from_numpy(z["in.u"]), torch.from_numpy(z["in.v"])), num_nodes=n)
        g._bnn, g._bne = torch.from_numpy(z["in.batch_num_nodes"]), torch.from_numpy(z["in.batch_num_edges"])
        g.ndata["atom_features"] = torch.from_numpy(z["in.atom_features"])
        g.ndata["frac_coords"] = torch.from_numpy(z["in.frac_coords"])
        g.ndata["V"] = torch.from_numpy(np.repeat(z["in.volume"], z["in.batch_num_nodes"]))
        g.edata["r"] = torch.from_numpy(z["in.r"])
        g.edata["images"] = torch.from_numpy(z["in.images"])
        return g, torch.from_numpy(z["in.lattice"])

    L = torch.nn.functional.l1_loss
    t = lambda k: torch.from_numpy(z[k]).to(DEV)  # noqa: E731
    for tag, kw in (("p", dict(include_pos_deriv=True, stresswise_weight=0.0)),
                    ("s", dict(batch_stress=False, stresswise_weight=0.05))):
        cfg = ALIGNNAtomWiseConfig(name="alignn_atomwise", alignn_layers=1, gcn_layers=1, hidden_features=32,
                                   embedding_features=16, atom_input_features=92, calculate_gradient=True, **kw)
        model = ALIGNNAtomWise(cfg)
        model.load_state_dict({k[len(tag) + 4:]: torch.from_numpy(np.asarray(v)) for k, v in z.items() if k.startswith(tag + ".sd.")})
        model = model.to(DEV).train()
        g, lat = graphs()
        res = model((g, lat))
        assert rel_err(res["out"], z[tag + ".pred"]) < 1e-4
        assert rel_err(res["grad"], z[tag + ".forces"]) < 3e-4, tag
        loss = L(res["out"], t(tag + ".t_energy")) + L(res["grad"], t(tag + ".t_forces"))
        if tag == "s":
            assert res["stresses"].shape == (3, 3) and rel_err(res["stresses"], z["s.stresses"]) < 3e-4
            loss = loss + 0.05 * L(res["stresses"], t("s.t_stress"))
        assert abs(loss.item() - float(z[tag + ".loss"])) < 2e-4
        loss.backward()
        nograd = set(z[tag + ".nograd"].tolist())
        gfloor = 1e-2 * max(float(np.abs(v).max()) for k, v in z.items() if k.startswith(tag + ".grad."))
        for k, p in model.named_parameters():
            if k not in nograd:
                assert rel_err(p.grad, z[tag + ".grad." + k], floor=gfloor) < 3e-3, (tag, k)
        g2, lat2 = graphs()
        ev = model.eval()((g2, lat2))
        assert rel_err(ev["grad"], z[tag + ".forces"]) < 3e-4, tag
        if tag == "s":
            assert rel_err(ev["stresses"], z["s.stresses"]) < 3e-4


def test_golden_ealignn_filtered_graph_model():
    """eALIGNNAtomWise against the reference's own class (alignn/models/ealignn_atomwise.py on the shims): bond
    vectors recomputed from frac_coords + lattice + images, bonds beyond inner_cutoff dropped before the line graph,
    torque-free forces, stresses, the loss and every second-order parameter gradient; eval() (fused kernels) agrees
    with train() (composed path); the reference's state_dict loads unchanged."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "shims"))
    import dgl  # shim: DGL-shaped container only

    from alignn_amd.ealignn_atomwise import eALIGNNAtomWise, eALIGNNAtomWiseConfig

    z = load_golden("ealignn_tiny.npz")
    cfg = eALIGNNAtomWiseConfig(name="ealignn_atomwise", alignn_layers=2, gcn_layers=2, hidden_features=32,
                                embedding_features=16, atom_input_features=92, calculate_gradient=True,
                                stresswise_weight=0.05, inner_cutoff=4.0)
    model = eALIGNNAtomWise(cfg)
    model.load_state_dict(state_dict_from_golden(z))
    model = model.to(DEV).train()
    n = int(z["in.batch_num_nodes"].sum())
    g = dgl.graph((torch.from_numpy(z["in.u"]), torch.from_numpy(z["in.v"])), num_nodes=n)
    g._bnn, g._bne = torch.from_numpy(z["in.batch_num_nodes"]), torch.from_numpy(z["in.batch_num_edges"])
    g.ndata["atom_features"] = torch.from_numpy(z["in.atom_features"])
    g.ndata["frac_coords"] = torch.from_numpy(z["in.frac_coords"])
    g.ndata["V"] = torch.from_numpy(np.repeat(z["in.volume"], z["in.batch_num_nodes"]))
    g.edata["r"] = torch.from_numpy(z["in.r"]) * 0.0 + 123.0  # must be ignored: r is recomputed from the positions
    g.edata["images"] = torch.from_numpy(z["in.images"])
    res = model((g, torch.from_numpy(z["in.lattice"])))
    assert rel_err(res["out"], z["pred"]) < 1e-4
    assert res["grad"].shape == (n, 3) and rel_err(res["grad"], z["forces"]) < 3e-4
    assert rel_err(res["stresses"], z["stresses"]) < 3e-4
    L = torch.nn.functional.l1_loss
    t = lambda k: torch.from_numpy(z[k]).to(DEV)  # noqa: E731
    loss = L(res["out"], t("t_energy")) + L(res["grad"], t("t_forces")) + 0.05 * L(res["stresses"], t("t_stress"))
    assert abs(loss.item() - float(z["loss"])) < 2e-4
    loss.backward()
    nograd = set(z["nograd"].tolist())
    gfloor = 1e-2 * max(float(np.abs(v).max()) for k, v in z.items() if k.startswith("grad."))
    cnt = 0
    for k, p in model.named_parameters():
        if k in nograd:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
        else:
            assert rel_err(p.grad, z["grad." + k], floor=gfloor) < 3e-3, k
            cnt += 1
    assert cnt > 40
    ev = model.eval()((g, torch.from_numpy(z["in.lattice"])))
    assert rel_err(ev["grad"], z["forces"]) < 3e-4 and rel_err(ev["stresses"], z["stresses"]) < 3e-4
    assert rel_err(ev["out"], z["pred"]) < 1e-4


def test_inference_path_folds_batchnorm_into_gate_pass():
    """eval() under no_grad: alignn_egc_gate_infer writes the edge output straight from the gate pass (BatchNorm =
    affine map of the running statistics).  Same predictions as the training-capable kernels in eval mode, as the
    golden eval vector, and it leaves the running statistics alone."""
    from alignn_amd import ops

    z = load_golden("alignn_tiny_eval.npz")
    raw = raw_from_golden(z)
    model = ALIGNN(ALIGNNConfig(name="alignn", alignn_layers=2, gcn_layers=2, hidden_features=32, embedding_features=16))
    model.load_state_dict(state_dict_from_golden(z))
    model = model.to(DEV).eval()
    batch = GraphBatch.from_raw(raw, device=DEV)
    before = {k: v.clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        fast = model(batch)
        try:
            ops.INFER_FUSED = False
            slow = model(batch)
        finally:
            ops.INFER_FUSED = True
    assert rel_err(fast, z["pred"]) < 1e-4 and rel_err(fast, slow) < 1e-5
    assert all(torch.equal(v, before[k]) for k, v in model.state_dict().items())
    # at the benchmark width (exercises the f16x3 projections fed by the amax the inference kernel tracks)
    torch.manual_seed(9)
    big = ALIGNN(ALIGNNConfig(name="alignn")).to(DEV).eval()
    b2 = GraphBatch.from_raw(make_batch(8, 30, seed0=77), device=DEV)
    with torch.no_grad():
        fast = big(b2)
        try:
            ops.INFER_FUSED = False
            slow = big(b2)
        finally:
            ops.INFER_FUSED = True
    assert rel_err(fast, slow) < 1e-5


def test_fused_line_graph_backward_matches_two_pass():
    """The three line-graph backward paths - alignn_egc_bwd_lg_dense (dense source-sorted blocks, one pass),
    alignn_egc_bwd_lg_fused (by source, GM re-read) and egc_bwd_dst + egc_bwd_src - are the same math with
    different (fixed) summation orders: gradients agree to fp32 round-off; also covers 1-atom cells (self-image
    bonds: the excluded entry of a segment) and cells up to 16 sources per atom; each path reproduces itself bit
    for bit."""
    from alignn_amd import ops
    from alignn_amd.synthetic import _one, batch_raw

    raw = batch_raw([_one(n, 60 + i, "crystal", 92) for i, n in enumerate((1, 6, 12, 2))])
    batch = GraphBatch.from_raw(raw, device=DEV)
    assert batch.lg.grp_seg_ptr is not None and batch.lg.dense_max_src > 0
    torch.manual_seed(2)
    model = ALIGNN(ALIGNNConfig(name="alignn", alignn_layers=2, gcn_layers=1, hidden_features=256)).to(DEV).train()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    target = torch.tensor([0.1, -0.4, 0.8, 0.3], device=DEV)
    grads = []
    try:
        for dense, fused in ((True, True), (False, True), (False, False), (True, True), (False, True)):
            ops.DENSE_LG_BACKWARD, ops.FUSED_LG_BACKWARD = dense, fused
            model.load_state_dict(sd)
            model.zero_grad(set_to_none=True)
            torch.nn.functional.l1_loss(model(batch), target).backward()
            grads.append({k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
    finally:
        ops.DENSE_LG_BACKWARD = ops.FUSED_LG_BACKWARD = True
    assert grads[0].keys() == grads[1].keys() == grads[2].keys()
    gmax = max(float(v.abs().max()) for v in grads[2].values())
    for k in grads[0]:
        assert rel_err(grads[0][k], grads[2][k], floor=1e-2 * gmax) < 2e-5, k
        assert rel_err(grads[1][k], grads[2][k], floor=1e-2 * gmax) < 2e-5, k
        assert torch.equal(grads[0][k], grads[3][k]), k  # every path reproduces itself bit for bit
        assert torch.equal(grads[1][k], grads[4][k]), k


def test_dense_line_graph_backward_wide_blocks():
    """Blocks with 17-32 sources per atom take the KMAX=8 instantiation; molecules give uneven blocks."""
    from alignn_amd import ops

    raw = make_batch(6, (9, 27), seed0=4242, kind="molecule")
    batch = GraphBatch.from_raw(raw, device=DEV)
    assert batch.lg.dense_max_src > 0
    torch.manual_seed(5)
    model = ALIGNN(ALIGNNConfig(name="alignn", alignn_layers=2, gcn_layers=1, hidden_features=64)).to(DEV).train()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    target = torch.randn(6, generator=torch.Generator().manual_seed(3)).to(DEV)
    grads = []
    try:
        for dense in (True, False):
            ops.DENSE_LG_BACKWARD = dense
            model.load_state_dict(sd)
            model.zero_grad(set_to_none=True)
            torch.nn.functional.l1_loss(model(batch), target).backward()
            grads.append({k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
    finally:
        ops.DENSE_LG_BACKWARD = True
    gmax = max(float(v.abs().max()) for v in grads[1].values())
    for k in grads[0]:
        assert rel_err(grads[0][k], grads[1][k], floor=1e-2 * gmax) < 2e-5, k
    print("max sources per atom:", batch.lg.dense_max_src)


def test_atomwise_g_lat_input_builds_line_graph_on_device():
    """forward((g, lat)) - the reference builds L(g) inside the forward (alignn_atomwise.py:376-386); we derive it
    on the device from g's canonical CSR.  Must give the same energies as the explicit (g, lg, lat) call."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "shims"))
    import dgl  # shim: DGL-shaped container only

    from alignn_amd import ALIGNNAtomWise, ALIGNNAtomWiseConfig

    raw = make_batch(3, 14, seed0=808)
    torch.manual_seed(6)
    model = ALIGNNAtomWise(ALIGNNAtomWiseConfig(name="alignn_atomwise", alignn_layers=2, gcn_layers=1, hidden_features=64,
                                                atom_input_features=92, calculate_gradient=False)).to(DEV).eval()

    def mk():
        g = dgl.graph((torch.from_numpy(raw.u), torch.from_numpy(raw.v)), num_nodes=raw.num_nodes)
        g._bnn = torch.from_numpy(raw.batch_num_nodes)
        g.ndata["atom_features"] = torch.from_numpy(raw.atom_features)
        g.edata["r"] = torch.from_numpy(raw.r)
        return g

    lg = dgl.graph((torch.from_numpy(raw.lg_u), torch.from_numpy(raw.lg_v)), num_nodes=raw.num_edges)
    lg.edata["h"] = torch.from_numpy(raw.h)
    lat = torch.from_numpy(raw.lattice)
    with torch.no_grad():
        a = model((mk(), lg, lat))["out"]
        b = model((mk(), lat))["out"]
    assert rel_err(b, a) < 1e-5


def test_default_config_depth_vs_oracle_on_x6_path():
    """Default ALIGNNConfig (4+4 layers, hidden 256) on 16 x 60-atom crystals: large enough that every wide
    projection runs on the split-product MFMA kernels (f16x3 wherever max|x| is tracked, i.e. everywhere on this path - the
    bf16x6 fallback is covered at GEMM level, tests/test_gpu_kernels.py) and the line-graph kernels see real segment lengths.  Prediction,
    every parameter gradient and the BatchNorm running statistics vs the CPU oracle (north_star bar 1e-4)."""
    B = 16
    raw = make_batch(B, 60, seed0=2024)
    model = ALIGNN(ALIGNNConfig(name="alignn"))
    model.load_state_dict(O.init_state_dict(seed=3))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(DEV).train()
    target = torch.randn(B, generator=torch.Generator().manual_seed(5))
    pred = model(GraphBatch.from_raw(raw, device=DEV))
    torch.nn.functional.l1_loss(pred, target.to(DEV)).backward()
    p = O.as_params(sd)
    stats = {}
    opred = O.alignn_forward(p, O.TorchGraph(raw), 4, 4, True, stats)
    torch.nn.functional.l1_loss(opred, target).backward()
    assert rel_err(pred, opred) < 1e-4
    gfloor = 1e-2 * max(float(t.grad.abs().max()) for t in p.values() if t.grad is not None)
    worst = 0.0
    for k, q in model.named_parameters():
        if p[k].grad is not None:
            e = rel_err(q.grad, p[k].grad, floor=gfloor)
            worst = max(worst, e)
            assert e < 1e-3, (k, e)
    after = O.running_stats_after_step(p, stats)
    sdn = model.state_dict()
    for k, v in after.items():
        assert rel_err(sdn[k], v, floor=1e-3) < 1e-4, k
    print(f"depth test: pred rel err {rel_err(pred, opred):.2e}, worst grad err {worst:.2e}")


def test_training_step_is_hipgraph_capturable():
    """The C-ABI kernels never allocate or synchronise and launch on the current stream only, so a whole
    training step (fwd + loss + bwd incl. the side stream + fused AdamW) captures into one hipGraph; replays must
    reproduce the eager loss trajectory bit for bit."""
    from alignn_amd.graphed import GraphedTrainStep

    raw = make_batch(4, 20, seed0=3)
    batch = GraphBatch.from_raw(raw, device=DEV)
    target = torch.tensor([0.2, -0.1, 0.7, 0.0], device=DEV)

    def fresh():
        torch.manual_seed(0)
        m = ALIGNN(ALIGNNConfig(name="alignn", alignn_layers=2, gcn_layers=2, hidden_features=64)).to(DEV).train()
        return m, torch.optim.AdamW(m.parameters(), lr=1e-3, fused=True, capturable=True)

    m, o = fresh()
    eager = []
    for _ in range(6):  # 3 warm-up steps inside GraphedTrainStep + 3 replays
        o.zero_grad(set_to_none=True)
        loss = torch.nn.functional.l1_loss(m(batch), target)
        loss.backward()
        o.step()
        eager.append(loss.detach().clone())
    m2, o2 = fresh()
    step = GraphedTrainStep(m2, batch, target, o2, warmup=3)
    graphed = [step().detach().clone() for _ in range(3)]
    for a, b in zip(eager[3:], graphed):
        assert torch.equal(a, b)
    for (k, p), (_, q) in zip(m.named_parameters(), m2.named_parameters()):
        assert torch.equal(p, q), k


def test_loader_staged_batches_match_explicit_line_graph():
    """loader.pack -> one pinned buffer -> PrefetchLoader (H2D + CSR + L(g) + cosines rebuilt on a staging stream):
    the model sees the same canonical batch as with the caller's explicit line graph and cosines - identical
    predictions, and a short training run over streamed batches follows the resident-batch run exactly."""
    from alignn_amd import loader

    raws = [make_batch(4, 10 + i, seed0=300 + 10 * i) for i in range(3)]
    packed = [loader.pack_raw(r, target=np.linspace(-1, 1, 4).astype(np.float32)) for r in raws]
    torch.manual_seed(3)
    model = ALIGNN(ALIGNNConfig(name="alignn", alignn_layers=2, gcn_layers=1, hidden_features=64)).to(DEV).eval()
    staged = list(loader.PrefetchLoader(packed, DEV, depth=2))
    assert len(staged) == 3
    with torch.no_grad():
        for raw, (b, t) in zip(raws, staged):
            ref = GraphBatch.from_raw(raw, device=DEV)
            assert torch.equal(b.lg.src, ref.lg.src) and torch.equal(b.lg.seg_ptr, ref.lg.seg_ptr)
            assert float((b.h - ref.h).abs().max()) < 2e-6  # HIP cosine kernel vs the numpy generator
            assert rel_err(model(b), model(ref)) < 1e-5
            assert torch.allclose(t.cpu(), torch.linspace(-1, 1, 4))
    # training over streamed batches == training over the same batches made resident up front
    def run(batches):
        torch.manual_seed(4)
        m = ALIGNN(ALIGNNConfig(name="alignn", alignn_layers=1, gcn_layers=1, hidden_features=64)).to(DEV).train()
        opt = torch.optim.SGD(m.parameters(), lr=1e-2)
        out = []
        for b, t in batches:
            opt.zero_grad(set_to_none=True)
            loss = torch.nn.functional.l1_loss(m(b), t)
            loss.backward()
            opt.step()
            out.append(float(loss))
        return out
    a = run(loader.PrefetchLoader(packed, DEV, depth=2))
    b = run([(GraphBatch.from_raw(r, device=DEV), torch.linspace(-1, 1, 4).to(DEV)) for r in raws])
    assert max(abs(x - y) for x, y in zip(a, b)) < 1e-5, (a, b)
