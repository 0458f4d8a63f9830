"""alignn_stage_batch (csrc/stage.hip): one C call builds what alignn_amd.graph.build_csr + line_graph_of + the cosine kernel
build with ~60 torch operations - every array must be IDENTICAL (the kernels index rows by these arrays), and a training
step on a staged batch must reproduce the step on the torch-staged one bit for bit."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from alignn_amd import ALIGNN, ALIGNNConfig, loader  # noqa: E402
from alignn_amd.synthetic import make_batch  # noqa: E402

DEV = "cuda"
FIELDS = "seg_ptr seg_node src dst out_ptr out_slot perm inv grp_seg_ptr grp_src_ptr seg_rank".split()


def _both(raw, **kw):
    p = loader.pack_raw(raw, target=np.arange(raw.batch_size, dtype=np.float32), **kw)
    a, ta = loader.stage(p, DEV)
    loader.STAGE_HIP = False
    try:
        b, tb = loader.stage(p, DEV)
    finally:
        loader.STAGE_HIP = True
    torch.cuda.synchronize()
    return p, a, ta, b, tb


@pytest.mark.parametrize("n,atoms,kind", [(1, 5, "crystal"), (3, 11, "crystal"), (8, 60, "crystal"), (64, 60, "crystal"),
                                          (40, (9, 27), "molecule")])
def test_every_array_equals_the_torch_builders(n, atoms, kind):
    raw = make_batch(n, atoms, seed0=11 + n, kind=kind)
    p, a, ta, b, tb = _both(raw)
    assert p.num_triplets == raw.num_triplets == a.lg.n_edges == b.lg.n_edges
    assert "staged_block" in a.cache and "staged_block" not in b.cache
    for name, ga, gb in (("g", a.g, b.g), ("lg", a.lg, b.lg)):
        assert (ga.n_nodes, ga.n_edges, ga.dense_max_src) == (gb.n_nodes, gb.n_edges, gb.dense_max_src), name
        for f in FIELDS:
            x, y = getattr(ga, f), getattr(gb, f)
            assert (x is None) == (y is None), (name, f)
            if x is not None:
                assert x.dtype == y.dtype and torch.equal(x, y), (name, f)
    for f in ("graph_ptr", "atom_features", "r", "h", "volume"):
        x, y = getattr(a, f), getattr(b, f)
        assert x.dtype == y.dtype and x.shape == y.shape, f
        if f == "volume":
            assert torch.allclose(x, y, rtol=1e-6)  # (float64 determinant on the host vs on the device)
        else:
            assert torch.equal(x, y), f
    assert torch.equal(ta, tb)


def test_a_training_step_on_a_staged_batch_is_bit_identical():
    raw = make_batch(12, 40, seed0=4)
    _p, a, ta, b, tb = _both(raw)
    outs = []
    for batch, t in ((a, ta), (b, tb)):
        torch.manual_seed(0)
        m = ALIGNN(ALIGNNConfig(name="alignn")).to(DEV).train()
        loss = torch.nn.functional.l1_loss(m(batch), t)
        loss.backward()
        torch.cuda.synchronize()
        outs.append([loss.detach().clone()] + [p.grad.clone() for p in m.parameters() if p.grad is not None])
    assert len(outs[0]) == len(outs[1]) and all(torch.equal(x, y) for x, y in zip(*outs))


def test_prefetch_loader_hands_over_staged_batches():
    raws = [make_batch(6, 30, seed0=50 + i) for i in range(4)]
    packed = [loader.pack_raw(r, target=np.zeros(r.batch_size, dtype=np.float32)) for r in raws]
    seen = 0
    for (b, t), r in zip(loader.PrefetchLoader(packed, DEV, depth=2), raws):
        assert b.lg.n_edges == r.num_triplets and b.h.numel() == r.num_triplets and t.numel() == r.batch_size
        assert float(b.h.abs().max()) <= 1.0
        seen += 1
    assert seen == 4


def test_device_coo_to_canonical_graphs_equals_the_torch_builders():
    """graph.csr_and_line_graph (what neighbors.crystal_batch and GraphBatch.from_coo(build_line_graph=True) use on the GPU:
    one host read + alignn_stage_batch) against build_csr + line_graph_of, incl. a bond list with self loops and an atom
    without bonds, and int64 inputs as the neighbour kernels emit them."""
    from alignn_amd import graph as G

    raw = make_batch(5, 23, seed0=8)
    cases = [(torch.from_numpy(raw.u).long(), torch.from_numpy(raw.v).long(), raw.num_nodes, torch.from_numpy(raw.r)),
             (torch.tensor([0, 1, 1, 2, 2, 2, 3, 0]), torch.tensor([1, 0, 2, 1, 2, 2, 0, 3]), 5, torch.randn(8, 3))]
    for u, v, n, r in cases:
        u, v, r = u.to(DEV), v.to(DEV), r.to(DEV)
        g1, lg1, r1 = G.csr_and_line_graph(u, v, n, r)
        G.STAGE_HIP = False
        try:
            g0, lg0, r0 = G.csr_and_line_graph(u, v, n, r)
        finally:
            G.STAGE_HIP = True
        for a, b in ((g1, g0), (lg1, lg0)):
            assert (a.n_nodes, a.n_edges, a.dense_max_src) == (b.n_nodes, b.n_edges, b.dense_max_src)
            for f in FIELDS:
                x, y = getattr(a, f), getattr(b, f)
                assert (x is None) == (y is None), f
                if x is not None:
                    assert x.dtype == y.dtype and torch.equal(x, y), f
        assert torch.equal(r1, r0)


def test_crystal_batch_runs_the_force_field_after_the_one_call_staging():
    from alignn_amd import ALIGNNAtomWise, ALIGNNAtomWiseConfig, neighbors
    from alignn_amd import graph as G
    from alignn_amd.synthetic import make_crystal

    lat, frac, _ = make_crystal(40, 77)
    lat_d, frac_d = torch.from_numpy(lat).to(DEV), torch.from_numpy(frac).to(DEV)
    feats = torch.randn(40, 92, device=DEV)
    torch.manual_seed(0)
    model = ALIGNNAtomWise(ALIGNNAtomWiseConfig(name="alignn_atomwise", alignn_layers=2, gcn_layers=1, hidden_features=64,
                                                embedding_features=32, atom_input_features=92, calculate_gradient=True)).to(DEV).eval()
    outs = []
    for flag in (True, False):
        G.STAGE_HIP = flag
        try:
            b = neighbors.crystal_batch([lat_d], [frac_d], atom_features=[feats])
            res = model(b)
            outs.append((res["out"].detach().clone(), res["grad"].detach().clone()))
        finally:
            G.STAGE_HIP = True
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def _from_raw_both(raw, **edit):
    from alignn_amd import GraphBatch, graph

    t = torch.from_numpy
    args = dict(lg_u=t(raw.lg_u), lg_v=t(raw.lg_v))
    args.update(edit)
    mk = lambda: GraphBatch.from_coo(t(raw.u), t(raw.v), raw.num_nodes, t(raw.batch_num_nodes), args["lg_u"], args["lg_v"],
                                     t(raw.atom_features), t(raw.r), args.get("h", t(raw.h)), device=DEV)
    a = mk()
    graph.STAGE_HIP = False
    try:
        b = mk()
    finally:
        graph.STAGE_HIP = True
    torch.cuda.synchronize()
    return a, b


@pytest.mark.parametrize("n,atoms,kind,shuffle", [(1, 5, "crystal", False), (8, 60, "crystal", True), (64, 60, "crystal", False),
                                                  (40, (9, 27), "molecule", True)])
def test_the_callers_own_line_graph_is_mapped_onto_the_canonical_rows(n, atoms, kind, shuffle):
    """GraphBatch.from_coo with the (g, lg) pair the reference hands over (alignn/models/alignn.py:293): on the GPU the
    caller's L(g) edge list is mapped by alignn_map_line_graph_rows; every array, lg.perm / lg.inv included, and the gathered
    cosines must equal what the generic torch builder makes of the same input - also with the caller's edges shuffled."""
    raw = make_batch(n, atoms, seed0=23 + n, kind=kind)
    edit = {}
    if shuffle:
        k = torch.randperm(raw.lg_u.size, generator=torch.Generator().manual_seed(n))
        edit = dict(lg_u=torch.from_numpy(raw.lg_u)[k], lg_v=torch.from_numpy(raw.lg_v)[k], h=torch.from_numpy(raw.h)[k])
    a, b = _from_raw_both(raw, **edit)
    assert a.lg.perm is not a.lg.inv or a.lg.n_edges == 0
    for name, ga, gb in (("g", a.g, b.g), ("lg", a.lg, b.lg)):
        assert (ga.n_nodes, ga.n_edges, ga.dense_max_src) == (gb.n_nodes, gb.n_edges, gb.dense_max_src), name
        for f in FIELDS:
            x, y = getattr(ga, f), getattr(gb, f)
            assert (x is None) == (y is None), (name, f)
            if x is not None:
                assert x.dtype == y.dtype and torch.equal(x, y), (name, f)
    assert torch.equal(a.h, b.h) and torch.equal(a.r, b.r)


def test_a_list_that_is_not_the_line_graph_takes_the_generic_builder():
    """A filtered line graph (fewer edges), a duplicated edge and a non-adjacent pair: csr_and_line_graph(lg_edges=...)
    declines (None) and from_coo builds what the torch builder builds."""
    from alignn_amd import graph

    raw = make_batch(4, 12, seed0=3)
    t = torch.from_numpy
    u, v = t(raw.u).to(DEV), t(raw.v).to(DEV)
    lu, lv = t(raw.lg_u).to(DEV), t(raw.lg_v).to(DEV)
    assert graph.csr_and_line_graph(u, v, raw.num_nodes, lg_edges=(lu, lv)) is not None
    assert graph.csr_and_line_graph(u, v, raw.num_nodes, lg_edges=(lu[:-3], lv[:-3])) is None  # filtered
    dup_u, dup_v = lu.clone(), lv.clone()
    dup_u[5], dup_v[5] = lu[6], lv[6]
    assert graph.csr_and_line_graph(u, v, raw.num_nodes, lg_edges=(dup_u, dup_v)) is None  # a repeated edge
    bad_u = lu.clone()
    wrong = (t(raw.v).to(DEV)[lu] != t(raw.v).to(DEV)[lu[0]]).nonzero()[0, 0]  # an edge ending at another atom
    bad_u[0] = lu[wrong]
    assert graph.csr_and_line_graph(u, v, raw.num_nodes, lg_edges=(bad_u, lv)) is None  # not adjacent
    a, b = _from_raw_both(raw, lg_u=t(raw.lg_u)[:-3], lg_v=t(raw.lg_v)[:-3], h=t(raw.h)[:-3])
    for f in FIELDS:
        x, y = getattr(a.lg, f), getattr(b.lg, f)
        assert (x is None) == (y is None) and (x is None or torch.equal(x, y)), f
    assert torch.equal(a.h, b.h)
