"""Round-2 regression tests on the GPU: the code-review findings (side-stream gradients under gradient accumulation /
DistributedDataParallel, stale feature caches, the eval shortcut dropping parameter gradients, the cutoff / penalty
branches of ALIGNNAtomWise) and the fused node-projection parameters."""

import os
import socket
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle", "shims"))

from alignn_amd import ALIGNN, ALIGNNConfig, GraphBatch, ops  # noqa: E402
from alignn_amd.synthetic import make_batch  # noqa: E402
from tests.helpers import load_golden, raw_from_golden, rel_err, state_dict_from_golden  # noqa: E402

DEV = "cuda"


def _dgl_pair(raw, volume=None):
    import dgl  # the torch-only shim: a DGL-shaped container

    t = torch.from_numpy
    g = dgl.graph((t(raw.u), t(raw.v)), num_nodes=raw.num_nodes)
    g._bnn, g._bne = t(raw.batch_num_nodes), t(raw.batch_num_edges)
    g.ndata["atom_features"] = t(raw.atom_features)
    g.edata["r"] = t(raw.r.copy())
    if volume is not None:
        g.ndata["V"] = t(np.repeat(volume, raw.batch_num_nodes))
    lg = dgl.graph((t(raw.lg_u), t(raw.lg_v)), num_nodes=raw.num_edges)
    lg.edata["h"] = t(raw.h.copy())
    return g, lg, t(raw.lattice)


def _small_model(seed=0, **kw):
    torch.manual_seed(seed)
    cfg = dict(name="alignn", alignn_layers=2, gcn_layers=1, hidden_features=64, embedding_features=32)
    cfg.update(kw)
    return ALIGNN(ALIGNNConfig(**cfg)).to(DEV).train()


# ---------------------------------------------------------------------------------------------
# side-stream weight gradients: gradient accumulation, hooks, DDP
# ---------------------------------------------------------------------------------------------
def _two_backwards_without_zero_grad(side):
    prev = ops._SIDE["enabled"]
    ops._SIDE["enabled"] = side
    try:
        model = _small_model(3)
        b1 = GraphBatch.from_raw(make_batch(6, 20, seed0=11), device=DEV)
        b2 = GraphBatch.from_raw(make_batch(6, 20, seed0=29), device=DEV)
        for b in (b1, b2):  # second backward ADDS to existing .grad tensors: AccumulateGrad launches kernels on them
            model(b).sum().backward()
        torch.cuda.synchronize()
        return {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    finally:
        ops._SIDE["enabled"] = prev


def test_gradient_accumulation_with_side_stream_equals_single_stream():
    a = _two_backwards_without_zero_grad(True)
    b = _two_backwards_without_zero_grad(False)
    assert a.keys() == b.keys() and len(a) > 40
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_deferred_join_policy():
    w = torch.nn.Parameter(torch.randn(4, 4, device=DEV))
    not_a_leaf = w * 2
    with torch.no_grad():
        assert ops._deferred_join_is_safe([w, None])
        w.grad = torch.zeros_like(w)
        assert not ops._deferred_join_is_safe([w])  # would be accumulated into on the main stream
        w.grad = None
        h = w.register_hook(lambda g: g)
        assert not ops._deferred_join_is_safe([w])  # a hook reads the gradient during backward
        h.remove()
        assert ops._deferred_join_is_safe([w])
        assert not ops._deferred_join_is_safe([not_a_leaf])  # not a leaf: more autograd nodes consume the gradient
    assert not ops._deferred_join_is_safe([w])  # grad mode on (create_graph): AccumulateGrad clones


def test_distributed_data_parallel_wrap_matches_unwrapped():
    """alignn/train.py:207 wraps the model in DistributedDataParallel(find_unused_parameters=True): the reducer's
    hooks copy every gradient into its buckets DURING backward - the side-stream gradients must be complete by then."""
    import torch.distributed as dist

    model = _small_model(5)
    ref = _small_model(5)
    ref.load_state_dict(model.state_dict())
    raw = make_batch(6, 20, seed0=41)
    batch = GraphBatch.from_raw(raw, device=DEV)
    target = torch.randn(6, device=DEV)
    torch.nn.functional.l1_loss(ref(batch), target).backward()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[torch.cuda.current_device()],
                                                        find_unused_parameters=True)
        for _ in range(2):  # second iteration: buckets rebuilt, .grad tensors exist
            ddp.zero_grad(set_to_none=False)
            torch.nn.functional.l1_loss(ddp(batch), target).backward()
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    n = 0
    for (k, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        if q.grad is not None:
            assert p.grad is not None and torch.equal(p.grad, q.grad), k
            n += 1
    assert n > 40


# ---------------------------------------------------------------------------------------------
# feature cache / eval shortcut / fused parameters
# ---------------------------------------------------------------------------------------------
def test_in_place_feature_edits_are_seen_on_the_next_forward():
    """The reference re-reads g.edata['r'] / lg.edata['h'] on every forward; only the index structures may be cached."""
    model = _small_model(7).eval()
    raw = make_batch(3, 12, seed0=5)
    g, lg, lat = _dgl_pair(raw)
    with torch.no_grad():
        a = model((g, lg, lat)).clone()
        assert getattr(g, "_alignn_amd_topology", None) is not None
        g.edata["r"].mul_(1.5)  # in-place edit of the caller's tensors (finite differences, relaxation loops)
        lg.edata["h"].mul_(-1.0)
        b = model((g, lg, lat))
        raw2 = make_batch(3, 12, seed0=5)
        raw2.r *= 1.5
        raw2.h *= -1.0
        c = model(GraphBatch.from_raw(raw2, device=DEV))
    assert rel_err(b, c) < 1e-6
    assert rel_err(a, c) > 1e-5  # (the edit does change the prediction)


def test_eval_shortcut_never_drops_parameter_gradients():
    model = _small_model(9).eval()
    batch = GraphBatch.from_raw(make_batch(4, 14, seed0=3), device=DEV)
    conv = model.gcn_layers[0]
    x = torch.randn(batch.g.n_nodes, 64, device=DEV)
    y = torch.randn(batch.g.n_edges, 64, device=DEV)
    xo, yo = conv(batch.g, x, y)  # eval mode, grad enabled, inputs without grad, parameters WITH grad
    (xo.sum() + yo.sum()).backward()
    assert conv.edge_gate.weight.grad is not None and float(conv.edge_gate.weight.grad.abs().max()) > 0
    assert conv.src_update.weight.grad is not None
    with torch.no_grad():
        xi, yi = conv(batch.g, x, y)  # the BatchNorm-folded inference pass gives the same values
    assert rel_err(xi, xo) < 1e-5 and rel_err(yi, yo) < 1e-5
    for p in conv.parameters():
        p.requires_grad_(False)
    xf, yf = conv(batch.g, x, y)  # nothing can receive a gradient: shortcut allowed even with grad mode on
    assert not xf.requires_grad and rel_err(xf, xo) < 1e-5


def test_fused_node_projection_parameters_alias_one_buffer():
    model = _small_model(13)
    batch = GraphBatch.from_raw(make_batch(4, 14, seed0=8), device=DEV)
    keys = list(model.state_dict().keys())
    target = torch.randn(4, device=DEV)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2, fused=True)
    torch.nn.functional.l1_loss(model(batch), target).backward()
    conv = model.alignn_layers[0].edge_update
    wcat, bcat = conv._fused_wb
    H = conv.src_gate.weight.shape[0]
    for i, lin in enumerate((conv.src_gate, conv.dst_gate, conv.dst_update, conv.src_update)):
        assert lin.weight.data_ptr() == wcat.data_ptr() + i * H * H * 4
        assert lin.bias.data_ptr() == bcat.data_ptr() + i * H * 4
        assert lin.weight.grad is not None and lin.weight.grad.shape == lin.weight.shape
    before = wcat.clone()
    opt.step()
    assert not torch.equal(before, wcat)  # the optimizer updates the fused buffer through the four views
    assert torch.equal(wcat[H:2 * H], conv.dst_gate.weight.detach())
    assert list(model.state_dict().keys()) == keys  # no new entries
    # a state_dict round trip and a device move keep values; the aliasing is re-established on the next forward
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model2 = _small_model(99)
    model2.load_state_dict(sd)
    model2 = model2.cpu().to(DEV)
    with torch.no_grad():
        p1, p2 = model.eval()(batch), model2.eval()(batch)
    assert torch.equal(p1, p2)
    assert model2.alignn_layers[0].edge_update.dst_gate.weight.data_ptr() == model2.alignn_layers[0].edge_update._fused_wb[0].data_ptr() + H * H * 4


# ---------------------------------------------------------------------------------------------
# ALIGNNAtomWise: cutoff function / penalty / energy_mult_natoms branches (alignn_atomwise.py:435-510)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["m", "e", "n", "q"])
def test_golden_atomwise_cutoff_and_penalty_branches(tag):
    from alignn_amd import ALIGNNAtomWise, ALIGNNAtomWiseConfig

    z = load_golden("atomwise_cutoff_penalty.npz")
    raw = raw_from_golden(z)
    kw = {"m": dict(use_cutoff_function=True, multiply_cutoff=True, inner_cutoff=6.0),
          "e": dict(use_cutoff_function=True, multiply_cutoff=False, inner_cutoff=6.0),
          "n": dict(energy_mult_natoms=False),
          "q": dict(energy_mult_natoms=False, calculate_gradient=False)}[tag]
    base = dict(name="alignn_atomwise", alignn_layers=1, gcn_layers=1, hidden_features=32, embedding_features=16,
                atom_input_features=92, calculate_gradient=True, stresswise_weight=0.05)
    base.update(kw)
    model = ALIGNNAtomWise(ALIGNNAtomWiseConfig(**base))
    model.load_state_dict(state_dict_from_golden(z, prefix=f"{tag}.sd."))
    model = model.to(DEV).train()
    g, lg, lat = _dgl_pair(raw, z["volume"])
    res = model([g, lg, lat])
    assert rel_err(res["out"], z[f"{tag}.pred"]) < 1e-4
    L = torch.nn.functional.l1_loss
    t = lambda k: torch.from_numpy(z[f"{tag}.{k}"]).to(DEV)  # noqa: E731
    loss = L(res["out"], t("t_energy"))
    if base["calculate_gradient"]:
        assert rel_err(res["grad"], z[f"{tag}.forces"]) < 2e-4
        assert rel_err(res["stresses"], z[f"{tag}.stresses"]) < 2e-4
        loss = loss + L(res["grad"], t("t_forces")) + 0.05 * L(res["stresses"], t("t_stress"))
    assert abs(loss.item() - float(z[f"{tag}.loss"])) < 1e-4
    loss.backward()
    nograd = set(z[f"{tag}.nograd"].tolist())
    gfloor = 1e-2 * max(float(np.abs(v).max()) for k, v in z.items() if k.startswith(f"{tag}.grad."))
    n = 0
    for k, p in model.named_parameters():
        if k in nograd:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
        else:
            assert rel_err(p.grad, z[f"{tag}.grad." + k], floor=gfloor) < 2e-3, k
            n += 1
    assert n > 20
    if base["calculate_gradient"]:  # inference path (fused kernels, first derivative only): same E / F / stress
        model.eval()
        ev = model([g, lg, lat])
        assert rel_err(ev["out"], z[f"{tag}.pred"]) < 1e-4
        assert rel_err(ev["grad"], z[f"{tag}.forces"]) < 2e-4
        assert rel_err(ev["stresses"], z[f"{tag}.stresses"]) < 2e-4


# ---------------------------------------------------------------------------------------------
# two lanes (T-row kernels on their own stream): same bits as one stream, eager and captured
# ---------------------------------------------------------------------------------------------
def _train_step_state(lanes_on, min_rows, mk_model, batch, target, steps=2):
    prev = (ops._LANE["enabled"], ops._LANE["min_rows"])
    ops._LANE["enabled"], ops._LANE["min_rows"] = ("1" if lanes_on else "0"), min_rows
    try:
        model = mk_model()
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, fused=True)
        for _ in range(steps):
            opt.zero_grad(set_to_none=True)
            pred = model(batch)
            torch.nn.functional.l1_loss(pred, target).backward()
            opt.step()
        torch.cuda.synchronize()
        out = {"pred": pred.detach().clone()}
        out.update({"g." + k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
        out.update({"s." + k: v.clone() for k, v in model.state_dict().items()})
        return out
    finally:
        ops._LANE["enabled"], ops._LANE["min_rows"] = prev


@pytest.mark.parametrize("case", ["default_config_16x60", "every_kernel_on_a_lane"])
def test_two_lanes_give_the_same_bits_as_one_stream(case):
    if case == "default_config_16x60":  # T ~ 169 k rows: the production split (line-graph rows on lane T)
        raw, min_rows = make_batch(16, 60, seed0=77), 131072

        def mk():
            torch.manual_seed(0)
            return ALIGNN(ALIGNNConfig(name="alignn")).to(DEV).train()
    else:  # stress: threshold 1 -> every convolution and embedding layer takes the lane code path
        raw, min_rows = make_batch(5, 16, seed0=78), 1

        def mk():
            return _small_model(21)
    batch = GraphBatch.from_raw(raw, device=DEV)
    target = torch.randn(raw.batch_size, generator=torch.Generator().manual_seed(2)).to(DEV)
    for trial in range(3):  # (a race would show up as run-to-run noise)
        a = _train_step_state(True, min_rows, mk, batch, target)
        b = _train_step_state(False, min_rows, mk, batch, target)
        assert a.keys() == b.keys()
        for k in a:
            assert torch.equal(a[k], b[k]), (case, trial, k)


def test_two_lanes_inside_a_hipgraph_capture():
    from alignn_amd.graphed import GraphedTrainStep

    prev = ops._LANE["min_rows"]
    ops._LANE["min_rows"] = 1  # small batch, but every layer forks to lane T inside the capture ("auto": capture only)
    assert ops._LANE["enabled"] == "auto"
    try:
        raw = make_batch(4, 20, seed0=3)
        batch = GraphBatch.from_raw(raw, device=DEV)
        target = torch.tensor([0.2, -0.1, 0.7, 0.0], device=DEV)

        def fresh():
            m = _small_model(0)
            return m, torch.optim.AdamW(m.parameters(), lr=1e-3, fused=True, capturable=True)

        m, o = fresh()
        eager = []
        for _ in range(6):
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
    finally:
        ops._LANE["min_rows"] = prev


def test_golden_atomwise_extra_features_with_forces():
    """extra_features != 0 together with training through the forces (the last combination that used to raise)."""
    from alignn_amd import ALIGNNAtomWise, ALIGNNAtomWiseConfig

    z = load_golden("atomwise_extra_forces.npz")
    raw = raw_from_golden(z)
    cfg = ALIGNNAtomWiseConfig(name="alignn_atomwise", alignn_layers=1, gcn_layers=1, hidden_features=32,
                               embedding_features=16, atom_input_features=92, calculate_gradient=True,
                               stresswise_weight=0.05, extra_features=3)
    model = ALIGNNAtomWise(cfg)
    model.load_state_dict(state_dict_from_golden(z))
    model = model.to(DEV).train()
    g, lg, lat = _dgl_pair(raw, z["volume"])
    g.ndata["extra_features"] = torch.from_numpy(z["extra_features"])
    res = model([g, lg, lat])
    assert res["out"].shape == (2, 1) and rel_err(res["out"], z["pred"]) < 1e-4
    assert rel_err(res["grad"], z["forces"]) < 2e-4 and rel_err(res["stresses"], z["stresses"]) < 2e-4
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
    assert n > 30


@pytest.mark.parametrize("nodes", ["ones", "random"])
def test_forces_against_finite_differences_of_the_float64_oracle(nodes):
    """The reference's second force test in our regime (alignn/tests/test_force_reduction.py:233-268): the 32-atom
    JVASP-98225 cluster, non-periodic radius graph (5 A), Linear(1,16) bond embedding, two BatchNorm EdgeGatedGraphConv
    in TRAIN mode, sum readout.  There: float64 autograd forces vs centred finite differences of the same float64 model.
    Here: the forces of the float32 HIP kernels (position route and bond-vector route reduced over in- minus out-edges,
    check (i) of that file) vs centred finite differences of the float64 oracle with the same parameters.

    ``nodes="ones"`` is the reference's set-up verbatim: constant node features make every atom's pre-activation equal up
    to the 1e-6 of the gate's epsilon, so BatchNorm over atoms amplifies differences that float32 barely resolves (that
    file runs in float64 for a reason) - checked with the reference's literal tolerances (atol 1e-5, rtol 1e-3; measured
    error 6e-6 at max|F| 2e-4).  ``nodes="random"`` gives the atoms distinct features: a well-conditioned model, checked
    tightly (1e-3 of the largest force)."""
    from alignn_amd.alignn import EdgeGatedGraphConv as BNConv
    from alignn_amd.graph import build_csr
    from oracle import alignn_oracle as O

    z = load_golden("graphs_sample_data.npz")
    i = z["names"].tolist().index("POSCAR-JVASP-98225.vasp")
    pos64 = torch.from_numpy(z[f"{i}.frac"] @ z[f"{i}.lat"])  # [32,3] Cartesian, float64
    n, width = pos64.shape[0], 16
    assert n == 32
    d = torch.cdist(pos64, pos64)
    v_, u_ = torch.nonzero((d <= 5.0) & ~torch.eye(n, dtype=torch.bool), as_tuple=True)  # u -> v, both directions present
    torch.manual_seed(0)
    emb, fc = torch.nn.Linear(1, width), torch.nn.Linear(width, 1)
    c1, c2 = BNConv(width, width), BNConv(width, width)
    p64 = {f"{pre}.{k}": t.detach().double().clone() for pre, mod in (("emb", emb), ("fc", fc), ("c1", c1), ("c2", c2))
           for k, t in mod.state_dict().items()}

    x0 = torch.ones(n, width, dtype=torch.float64) if nodes == "ones" else torch.randn(n, width, dtype=torch.float64)

    def energy64(pos):
        bond = pos[v_] - pos[u_]
        y = bond.norm(dim=1, keepdim=True) @ p64["emb.weight"].t() + p64["emb.bias"]
        x = x0.clone()
        x, y = O.edge_gated_conv(p64, "c1", u_, v_, x, y, training=True)
        x, y = O.edge_gated_conv(p64, "c2", u_, v_, x, y, training=True)
        return (x @ p64["fc.weight"].t() + p64["fc.bias"]).sum()

    delta = 1e-6
    f_fd = torch.zeros(n, 3, dtype=torch.float64)
    with torch.no_grad():
        for a in range(n):
            for k in range(3):
                xa, xb = pos64.clone(), pos64.clone()
                xa[a, k] -= delta
                xb[a, k] += delta
                f_fd[a, k] = -(energy64(xb) - energy64(xa)) / (2 * delta)

    emb, fc, c1, c2 = emb.to(DEV), fc.to(DEV), c1.to(DEV).train(), c2.to(DEV).train()
    csr = build_csr(u_.to(DEV), v_.to(DEV), n)
    pos = pos64.float().to(DEV).requires_grad_(True)
    bondvec = pos[csr.dst.long()] - pos[csr.src.long()]  # canonical slot order
    y = emb(bondvec.norm(dim=1, keepdim=True))
    x = x0.float().to(DEV)
    x, y = c1(csr, x, y)
    x, y = c2(csr, x, y)
    energy = fc(x).sum()
    f_x = -torch.autograd.grad(energy, pos, retain_graph=True)[0]
    pf = -torch.autograd.grad(energy, bondvec)[0]
    f_vec = torch.zeros(n, 3, device=DEV).index_add(0, csr.dst.long(), pf) - torch.zeros(n, 3, device=DEV).index_add(0, csr.src.long(), pf)
    for name, f in (("positions", f_x), ("bond vectors", f_vec)):
        f = f.double().cpu()
        err, scale = float((f - f_fd).abs().max()), float(f_fd.abs().max())
        print(f"nodes={nodes}, forces by {name}: max |F| {scale:.3e}, max error vs float64 finite differences {err:.3e}")
        assert torch.isclose(f, f_fd, atol=1e-5, rtol=1e-3).all(), (name, err)  # the reference's tolerances
        # (a RELATIVE bound only for the well-conditioned set-up.  With x = ones the atoms' pre-activations differ in the 7th
        # digit - x_pre = Ux + b (1 - 1e-6 / (S0 + 1e-6)) - i.e. the DATA is below float32's resolution before BatchNorm
        # amplifies it: no float32 statistic, however summed, recovers it (round 3 moved the statistics to pivot slabs and
        # the error stayed at 4 % of the largest force); that file runs in float64 for this reason, and so does its 1:1
        # port tests/test_force_reduction_port.py)
        if nodes == "random":
            assert err < 1e-3 * scale, (name, err, scale)
    assert rel_err(f_vec, f_x) < 1e-5


def test_batchnorm_backward_reductions_from_the_projection_epilogue():
    """alignn_gemm_nt_f16x3_bnred: the column sums BatchNorm's backward starts with, taken in the epilogue of the
    projection that produces the gradient.  Default config on 16 x 60-atom crystals (T ~ 169 k): the fused route is
    taken for the three line-graph convolutions that feed another one - and for the angle embedding when that runs as a
    chain of layers (ops.ANGLE_FUSED off; csrc/angle.hip keeps no pre-activation a projection could reduce against) -, and
    the training step agrees with the separate reduction kernel to rounding (the summation order differs)."""
    raw = make_batch(16, 60, seed0=91)
    batch = GraphBatch.from_raw(raw, device=DEV)
    target = torch.randn(16, generator=torch.Generator().manual_seed(3)).to(DEV)
    res = {}
    for fused in (True, False):
        ops.BNRED_FUSED = fused
        ops.BNRED_STATS.update(fused=0, used=0)
        from alignn_amd import cmodel

        try:
            torch.manual_seed(0)
            model = ALIGNN(ALIGNNConfig(name="alignn")).to(DEV).train()
            with cmodel.disabled():  # (BNRED_STATS counts the per-operator path's registry traffic)
                torch.nn.functional.l1_loss(model(batch), target).backward()
            torch.cuda.synchronize()
        finally:
            ops.BNRED_FUSED = True
        res[fused] = ({k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}, dict(ops.BNRED_STATS))
    assert res[True][1] == {"fused": 3, "used": 3} and res[False][1] == {"fused": 0, "used": 0}
    ops.ANGLE_FUSED = False
    try:
        ops.BNRED_STATS.update(fused=0, used=0)
        torch.manual_seed(0)
        model = ALIGNN(ALIGNNConfig(name="alignn")).to(DEV).train()
        with cmodel.disabled():
            torch.nn.functional.l1_loss(model(batch), target).backward()
        torch.cuda.synchronize()
        assert dict(ops.BNRED_STATS) == {"fused": 4, "used": 4}
    finally:
        ops.ANGLE_FUSED = True
    ga, gb = res[True][0], res[False][0]
    gmax = max(float(v.abs().max()) for v in gb.values())
    for k in gb:
        assert float((ga[k] - gb[k]).abs().max()) < 1e-4 * max(float(gb[k].abs().max()), 1e-3 * gmax), k


def test_training_loop_of_the_reference_with_the_drop_in_model():
    """alignn.train.train_dgl(config, model=...) is the supported injection point (alignn/train.py:51,180-183).  The
    reference's own train_dgl cannot travel to the GPU box; oracle/make_golden_train.py ran it - unmodified, with the
    reference's ALIGNNAtomWise - on this seeded dataset and stored its history_train / history_val, and pinned
    oracle/train_loop_oracle.py (the restated per-batch loop) to it exactly.  Here that loop drives OUR model: two epochs
    of energy + force + stress training (AdamW, group_decay, validation passes) must give the same loss histories."""
    import dgl  # shim

    from alignn_amd import ALIGNNAtomWise, ALIGNNAtomWiseConfig
    from oracle import train_loop_oracle as TL
    from oracle.train_data import MODEL_KW, TRAIN_CFG, make_loaders, stress_targets

    z = load_golden("train_loop.npz")
    model = ALIGNNAtomWise(ALIGNNAtomWiseConfig(**MODEL_KW))
    model.load_state_dict(state_dict_from_golden(z))
    tr, va, _ = make_loaders(dgl)
    cfg = dict(TRAIN_CFG, model=dict(MODEL_KW))
    h_tr, h_va = TL.train_atomwise(model, tr, va, cfg, torch.device("cuda"), lambda g: stress_targets(dgl, g))
    for mine, ref, name in ((h_tr, z["history_train"], "train"), (h_va, z["history_val"], "val")):
        mine = np.array(mine)
        err = np.abs(mine - ref).max() / np.abs(ref).max()
        print(f"history_{name}: ours {mine[:, 0].tolist()} reference {ref[:, 0].tolist()} (max rel. difference {err:.2e})")
        assert err < 1e-3, (name, mine, ref)


def test_u_add_v_in_the_projection_epilogue_gives_the_same_bits():
    """alignn_gemm_nt_f16x3_gather + alignn_egc_gate_fwd_pre against alignn_gemm_nt_f16x3 + alignn_egc_gate_fwd:
    m = (A[u] + Bd[v]) + C either way, so a whole training step is bit-identical."""
    raw = make_batch(16, 60, seed0=92)
    batch = GraphBatch.from_raw(raw, device=DEV)
    target = torch.randn(16, generator=torch.Generator().manual_seed(4)).to(DEV)
    res = {}
    for fused in (True, False):
        ops.GATHER_FUSED = fused
        ops.STATS_FUSED = False  # (the statistics-in-the-epilogue route rides on the gather variant and sums in another order)
        try:
            torch.manual_seed(0)
            model = ALIGNN(ALIGNNConfig(name="alignn")).to(DEV).train()
            pred = model(batch)
            torch.nn.functional.l1_loss(pred, target).backward()
            torch.cuda.synchronize()
        finally:
            ops.GATHER_FUSED = True
            ops.STATS_FUSED = True
        res[fused] = (pred.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None},
                      {k: v.clone() for k, v in model.state_dict().items() if "running" in k})
    assert torch.equal(res[True][0], res[False][0])
    for k in res[False][1]:
        assert torch.equal(res[True][1][k], res[False][1][k]), k
    for k in res[False][2]:
        assert torch.equal(res[True][2][k], res[False][2][k]), k


def test_batchnorm_statistics_from_the_projection_epilogue():
    """STATS in the projection's epilogue (alignn_gemm_nt_f16x3_stats / _gather with stats) + the gate pass that
    normalises right away (alignn_egc_gate_fwd_pre_norm) against projection -> statistics pass -> normalise pass: same
    training step to rounding (the statistics are summed per row tile instead of per slab)."""
    raw = make_batch(16, 60, seed0=93)
    batch = GraphBatch.from_raw(raw, device=DEV)
    target = torch.randn(16, generator=torch.Generator().manual_seed(5)).to(DEV)
    res = {}
    for fused in (True, False):
        ops.STATS_FUSED = fused
        try:
            torch.manual_seed(0)
            model = ALIGNN(ALIGNNConfig(name="alignn")).to(DEV).train()
            pred = model(batch)
            torch.nn.functional.l1_loss(pred, target).backward()
            torch.cuda.synchronize()
        finally:
            ops.STATS_FUSED = True
        res[fused] = (pred.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None},
                      {k: v.clone() for k, v in model.state_dict().items() if "running" in k})
    assert rel_err(res[True][0], res[False][0]) < 1e-5
    gmax = max(float(v.abs().max()) for v in res[False][1].values())
    for k, g in res[False][1].items():
        assert float((res[True][1][k] - g).abs().max()) < 1e-4 * max(float(g.abs().max()), 1e-3 * gmax), k
    for k, v in res[False][2].items():
        assert rel_err(res[True][2][k], v, floor=1e-3) < 1e-5, k


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [131073, 140011, 262144 + 64])
def test_long_projections_every_epilogue_variant_persistent_and_one_tile_kernels(rows):
    """The f16x3 projection with each fused epilogue (plain, residual, statistics, u_add_v gather, gather + statistics,
    BatchNorm-backward sums with / without residual) at row counts that take the PERSISTENT kernel (>= 1 024 row tiles),
    including a last tile whose second 64-row strip lies entirely past the end (rows % 128 in (0, 64]: the scratch slab)
    and ragged ends - against float64, and the one-tile kernels of the same shapes (ALIGNN_AMD_X6_PERSIST=0, read per
    call) against the same bound.  Bound: 5e-6 of the largest reference element (tools/x6_family_check.py)."""
    import ctypes
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location(
        "x6_family_check", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "x6_family_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    libc = ctypes.CDLL(None)
    try:
        for mode in ("1", "0"):
            mod.MODES = [mode]
            mod.t = lambda fn, rounds=1: [1.0]  # (no timing in the test)
            assert mod.main(rows) == 0, f"ALIGNN_AMD_X6_PERSIST={mode}"
    finally:
        libc.unsetenv(b"ALIGNN_AMD_X6_PERSIST")


@pytest.mark.gpu
def test_weight_images_from_one_launch_equal_the_three_launch_route():
    """alignn_split_f16x2_both (max|w| + the images of w and w^T in one launch) against alignn_absmax +
    alignn_split_f16x2 x 2: same maximum, same bytes; and a training step of the model is bit-identical either way."""
    from alignn_amd import ALIGNN, ALIGNNConfig, GraphBatch, ops
    from alignn_amd.synthetic import make_batch

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(11)
    for n, k in ((256, 256), (1024, 256), (256, 64), (64, 256)):
        w = (torch.randn(n, k, generator=g) * 0.07).to(dev).requires_grad_(True)
        ops.SPLIT_BOTH = True
        try:
            a = ops.split_f16x2(w)
            at = ops.split_f16x2(w, transpose=True)
            ops.SPLIT_BOTH = False
            ops._W_IMG_T.clear()
            b = ops.split_f16x2(w)
            bt = ops.split_f16x2(w, transpose=True)
        finally:
            ops.SPLIT_BOTH = True
        assert torch.equal(a.amax.reshape(()), b.amax.reshape(())) and torch.equal(a.buf, b.buf) and torch.equal(at.buf, bt.buf)
        assert (at.n, at.k) == (k, n)

    batch = GraphBatch.from_raw(make_batch(16, 40, seed0=21), device=dev)
    target = torch.randn(16, generator=torch.Generator().manual_seed(2)).to(dev)
    outs = []
    for both in (True, False):
        ops.SPLIT_BOTH = both
        ops._W_IMG_T.clear()
        try:
            torch.manual_seed(0)
            m = ALIGNN(ALIGNNConfig(name="alignn")).to(dev).train()
            loss = torch.nn.functional.l1_loss(m(batch), target)
            loss.backward()
            torch.cuda.synchronize()
            outs.append((loss.detach().clone(), [p.grad.clone() for p in m.parameters() if p.grad is not None]))
        finally:
            ops.SPLIT_BOTH = True
    assert torch.equal(outs[0][0], outs[1][0])
    assert len(outs[0][1]) == len(outs[1][1]) and all(torch.equal(x, y) for x, y in zip(outs[0][1], outs[1][1]))


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(3840, 1024, 256), (777, 512, 256), (5000, 1024, 512), (64, 2048, 256)])
def test_split_reduction_input_gradient_vs_float64(M, N, K):
    """alignn_gemm_nn_split (reduction slabs + fixed-order sum + addend) against float64 and against alignn_gemm_nn."""
    from alignn_amd import _lib, ops

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, N, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) / N ** 0.5).to(dev)
    add = torch.randn(M, K, generator=g).to(dev)
    assert _lib.load().alignn_gemm_nn_split_workspace(M, N, K) > 0
    ref = a.double() @ w.double() + add.double()
    ops.NN_SPLIT = True
    try:
        out = ops.gemm_nn(a, w, add)
        again = ops.gemm_nn(a, w, add)
        ops.NN_SPLIT = False
        plain = ops.gemm_nn(a, w, add)
    finally:
        ops.NN_SPLIT = True
    scale = float(ref.abs().max())
    assert torch.equal(out, again)  # fixed summation order
    assert float((out.double() - ref).abs().max()) <= 2e-6 * scale
    assert float((out - plain).abs().max()) <= 2e-6 * scale
    no_add = ops.gemm_nn(a, w)
    assert float((no_add.double() - (ref - add.double())).abs().max()) <= 2e-6 * scale


@pytest.mark.gpu
@pytest.mark.parametrize("rows,F,eval_mode", [(676200, 64, False), (50712, 256, False), (1000, 256, True), (37, 1024, False)])
def test_norm_backward_pass_that_also_sums_its_output_columns(rows, F, eval_mode):
    """alignn_bn_silu_bwd_apply_sum against alignn_bn_silu_bwd_apply (same GX: the arithmetic per element is the same)
    and its column sums against float64 sums of that GX."""
    from alignn_amd import _lib, ops

    dev = torch.device("cuda", 0)
    lib = _lib.load()
    g = torch.Generator().manual_seed(rows + F)
    gy = torch.randn(rows, F, generator=g).to(dev)
    x = (torch.randn(rows, F, generator=g) * 1.3 + 0.2).to(dev)
    gamma = (1 + 0.1 * torch.randn(F, generator=g)).to(dev)
    beta = (0.1 * torch.randn(F, generator=g)).to(dev)
    mean, rstd = x.mean(0), torch.rsqrt(x.var(0, unbiased=False) + 1e-5)
    stat = torch.stack([mean, rstd, gamma * rstd, beta]).contiguous()
    red = ops._bn_silu_bwd_reduce(gy, x, stat)
    ref = ops._bn_silu_bwd_apply(gy, x, stat, gamma, red, eval_mode, torch.empty_like(x))
    out = torch.empty_like(x)
    slabs = lib.alignn_col_stats_slabs(rows)
    part = torch.empty(slabs, F, device=dev)
    amax = torch.zeros(1, device=dev)
    ops.check(lib.alignn_bn_silu_bwd_apply_sum(ops.ptr(gy), gy.stride(0), ops.ptr(x), x.stride(0), ops.ptr(stat), ops.ptr(red),
                                               int(eval_mode), ops.ptr(out), out.stride(0), rows, F, ops.ptr(amax), ops.ptr(part),
                                               ops.stream()), "apply_sum")
    assert torch.equal(out, ref)
    assert float(amax) == float(ref.abs().max())
    sums = torch.empty(F, device=dev)
    ops.check(lib.alignn_slab_sum(ops.ptr(part), slabs, F, ops.ptr(sums), ops.stream()), "slab_sum")
    want = ref.double().sum(0)
    scale = float(ref.double().abs().sum(0).max())
    assert float((sums.double() - want).abs().max()) <= 2e-6 * scale
