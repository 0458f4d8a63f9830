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
