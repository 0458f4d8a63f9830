"""CPU tests of the host-side logic that needs no GPU: synthetic generator, canonical graph layout."""

import numpy as np
import torch

from alignn_amd.graph import GraphBatch, build_csr
from alignn_amd.synthetic import _one, batch_raw, make_batch


def test_generator_matches_reference_input_contract():
    raw = make_batch(3, 20, seed0=11)
    # both directions of every bond are stored consecutively with opposite displacement (graphs.py:253-257)
    assert (raw.u[0::2] == raw.v[1::2]).all() and (raw.v[0::2] == raw.u[1::2]).all()
    assert np.allclose(raw.r[0::2], -raw.r[1::2])
    # k-NN with ties: every atom has at least 12 in-edges
    assert np.bincount(raw.v, minlength=raw.num_nodes).min() >= 12
    # line graph: e1 -> e2 iff dst(e1) == src(e2), e1 != e2 (backtracking included)
    assert (raw.v[raw.lg_u] == raw.u[raw.lg_v]).all() and (raw.lg_u != raw.lg_v).all()
    assert raw.num_triplets == int(sum(np.bincount(raw.v, minlength=raw.num_nodes)[raw.u]) - (raw.u == raw.v).sum())
    assert np.abs(raw.h).max() <= 1.0
    # batching = disjoint union: no edge crosses crystals
    gid = np.repeat(np.arange(3), raw.batch_num_nodes)
    assert (gid[raw.u] == gid[raw.v]).all()
    assert raw.batch_num_edges.sum() == raw.num_edges and raw.batch_num_triplets.sum() == raw.num_triplets


def test_one_atom_cell_is_all_self_image_edges():
    g = _one(1, 7, "crystal", 92)
    assert g.num_nodes == 1 and g.num_edges >= 24 and (g.u == 0).all() and (g.v == 0).all()
    assert g.num_triplets == g.num_edges * (g.num_edges - 1)  # every pair except e1 == e2


def test_canonical_layout_invariants():
    raw = batch_raw([_one(n, 90 + i, "crystal", 92) for i, n in enumerate((1, 5, 9))])
    b = GraphBatch.from_raw(raw)
    g, lg = b.g, b.lg
    u, v = torch.from_numpy(raw.u), torch.from_numpy(raw.v)
    # slot k holds caller edge perm[k]; segments are contiguous and cover each node exactly once
    assert (u[g.perm] == g.src.long()).all() and (v[g.perm] == g.dst.long()).all()
    assert (g.inv[g.perm] == torch.arange(g.n_edges)).all()
    assert (g.dst.long() == torch.repeat_interleave(torch.arange(g.n_nodes), (g.seg_ptr[1:] - g.seg_ptr[:-1]).long())).all()
    # by-source view lists every slot once, grouped by source
    assert sorted(g.out_slot.tolist()) == list(range(g.n_edges))
    assert (g.src[g.out_slot.long()].long() == torch.repeat_interleave(torch.arange(g.n_nodes), (g.out_ptr[1:] - g.out_ptr[:-1]).long())).all()
    # line graph nodes are g's canonical slots; its segments are ordered by the source atom of the bond
    e1, e2 = g.inv[torch.from_numpy(raw.lg_u)], g.inv[torch.from_numpy(raw.lg_v)]
    assert (e1[lg.perm] == lg.src.long()).all() and (e2[lg.perm] == lg.dst.long()).all()
    seg_src_atom = g.src[lg.seg_node.long()]
    assert (seg_src_atom[1:] >= seg_src_atom[:-1]).all()
    # every source of a line-graph segment is an in-edge of that segment's source atom -> one contiguous block of rows
    for s in range(0, lg.n_nodes, 7):
        a, z = int(lg.seg_ptr[s]), int(lg.seg_ptr[s + 1])
        j = int(seg_src_atom[s])
        srcs = lg.src[a:z].long()
        assert ((srcs >= int(g.seg_ptr[j])) & (srcs < int(g.seg_ptr[j + 1]))).all()
    # inputs follow the permutations; volumes and bond offsets per crystal
    assert torch.allclose(b.r, torch.from_numpy(raw.r)[g.perm]) and torch.allclose(b.h, torch.from_numpy(raw.h)[lg.perm])
    assert b.edge_graph_ptr.tolist() == [0] + np.cumsum(raw.batch_num_edges).tolist()
    assert torch.allclose(b.volume, torch.from_numpy(np.abs(np.linalg.det(raw.lattice))).float(), rtol=1e-5)


def test_build_csr_with_isolated_nodes_and_custom_segment_order():
    u = torch.tensor([0, 0, 2, 2, 2])
    v = torch.tensor([1, 1, 0, 1, 2])
    order = torch.tensor([3, 1, 0, 2])  # node 3 has no edges at all
    c = build_csr(u, v, 4, order)
    assert c.seg_node.tolist() == [3, 1, 0, 2]
    assert c.seg_ptr.tolist() == [0, 0, 3, 4, 5]
    assert c.dst.tolist() == [1, 1, 1, 0, 2] and c.src.tolist() == [0, 0, 2, 2, 2]
    assert c.out_ptr.tolist() == [0, 2, 2, 5, 5]


def test_device_line_graph_equals_callers_line_graph():
    """line_graph_of(g) (built from g's CSR alone) == canonicalisation of the caller's explicit L(g)."""
    from alignn_amd.graph import line_graph_of

    raw = batch_raw([_one(n, 120 + i, "crystal", 92) for i, n in enumerate((1, 4, 8))])
    b = GraphBatch.from_raw(raw)
    lg2 = line_graph_of(b.g)
    lg = b.lg
    assert lg2.n_nodes == lg.n_nodes and lg2.n_edges == lg.n_edges
    for f in ("seg_ptr", "seg_node", "out_ptr", "grp_seg_ptr", "grp_src_ptr"):
        assert torch.equal(getattr(lg2, f), getattr(lg, f)), f
    # same edge multiset per segment (order inside a segment follows the caller in one case, slot order in the other)
    key = lambda c: torch.sort(c.dst.long() * c.n_nodes + c.src.long()).values  # noqa: E731
    assert torch.equal(key(lg2), key(lg))
    assert (b.g.dst[lg2.src.long()] == b.g.src[lg2.dst.long()]).all() and (lg2.src != lg2.dst).all()


def test_line_graph_blocks_are_dense_and_source_sorted():
    """What alignn_egc_bwd_lg_dense relies on: after canonicalisation every segment of L(g) lists ALL in-edges of its
    centre atom in ascending order, minus the segment's own bond when it is a self-image - for the caller's COO
    line graph (any edge order) and for the one derived on the device; a filtered line graph is detected."""
    import numpy as np
    import torch

    from alignn_amd.graph import GraphBatch, build_csr, line_graph_of
    from alignn_amd.synthetic import _one, batch_raw

    raw = batch_raw([_one(n, 11 + i, "crystal", 92) for i, n in enumerate((1, 5, 9))])  # incl. a 1-atom cell
    rng = np.random.default_rng(0)
    order = rng.permutation(raw.lg_u.shape[0])  # the caller's line-graph edges in arbitrary order
    b = GraphBatch.from_coo(torch.from_numpy(raw.u), torch.from_numpy(raw.v), raw.num_nodes,
                            torch.from_numpy(raw.batch_num_nodes), torch.from_numpy(raw.lg_u[order]),
                            torch.from_numpy(raw.lg_v[order]), h=torch.from_numpy(raw.h[order]))
    g, lg = b.g, b.lg
    indeg = (g.seg_ptr[1:] - g.seg_ptr[:-1]).max().item()
    assert lg.dense_max_src == indeg > 0
    lg2 = line_graph_of(g)
    assert lg2.dense_max_src == indeg
    assert torch.equal(lg2.src, lg.src) and torch.equal(lg2.dst, lg.dst) and torch.equal(lg2.seg_ptr, lg.seg_ptr)
    # the cosines followed their edges through the extra within-segment sort
    ref = GraphBatch.from_raw(raw)
    assert torch.allclose(b.h, ref.h)
    # a filtered line graph (drop every 7th edge) is still block structured but no longer dense
    keep = np.arange(raw.lg_u.shape[0]) % 7 != 0
    f = GraphBatch.from_coo(torch.from_numpy(raw.u), torch.from_numpy(raw.v), raw.num_nodes,
                            torch.from_numpy(raw.batch_num_nodes), torch.from_numpy(raw.lg_u[keep]),
                            torch.from_numpy(raw.lg_v[keep]))
    assert f.lg.grp_seg_ptr is not None and f.lg.dense_max_src == 0


def test_packed_batch_roundtrip_and_staging_on_cpu():
    """loader.pack -> one buffer -> loader.stage rebuilds the SAME canonical batch GraphBatch.from_raw builds from
    the explicit line graph (structure bit-identical; cosines are left to the GPU kernel / the model here)."""
    import numpy as np
    import torch

    from alignn_amd import loader
    from alignn_amd.graph import GraphBatch
    from alignn_amd.synthetic import make_batch

    raw = make_batch(3, 11, seed0=77)
    tgt = np.array([0.5, -1.0, 2.0], dtype=np.float32)
    p = loader.pack_raw(raw, target=tgt, pin=False)
    assert p.nbytes < 0.3 * (raw.lg_u.nbytes + raw.lg_v.nbytes + raw.u.nbytes + raw.v.nbytes)  # no T-sized list crosses
    assert torch.equal(p.host("u"), torch.from_numpy(raw.u.astype(np.int32)))
    b, t = loader.stage(p, "cpu", cosines=False)
    ref = GraphBatch.from_raw(raw)
    for name in ("seg_ptr", "src", "dst", "out_ptr", "out_slot"):
        assert torch.equal(getattr(b.g, name), getattr(ref.g, name)), name
        assert torch.equal(getattr(b.lg, name), getattr(ref.lg, name)), name
    assert torch.equal(b.lg.seg_node, ref.lg.seg_node) and b.lg.dense_max_src == ref.lg.dense_max_src
    assert torch.equal(b.r, ref.r) and torch.equal(b.atom_features, ref.atom_features)
    assert torch.allclose(b.volume, ref.volume) and torch.equal(b.graph_ptr, ref.graph_ptr)
    assert torch.equal(t, torch.from_numpy(tgt)) and b.h is None
    # index-compressed atoms: species + a device-resident table
    table = torch.randn(5, raw.atom_features.shape[1])
    species = np.arange(raw.num_nodes) % 5
    p2 = loader.pack(raw.u, raw.v, raw.batch_num_nodes, raw.r, raw.lattice, species=species, pin=False)
    b2, t2 = loader.stage(p2, "cpu", feature_table=table, cosines=False)
    assert t2 is None and torch.equal(b2.atom_features, table[torch.from_numpy(species)])
    assert p2.nbytes < p.nbytes
    import pytest
    with pytest.raises(ValueError):
        loader.pack(raw.u, raw.v + 10_000, raw.batch_num_nodes, raw.r, raw.lattice, species=species, pin=False)
    got = [x for x in loader.PrefetchLoader([p, p2], "cpu", feature_table=table, cosines=False)]
    assert len(got) == 2 and torch.equal(got[0][0].lg.src, ref.lg.src)


def test_device_knn_graph_matches_numpy_builder():
    """alignn_amd.neighbors.knn_multigraph (torch tensor ops, runs on the device) gives the same bond list, in the
    same order, as the numpy restatement of the reference's kNN-12 / 8 A builder - incl. 1-atom cells (all bonds are
    self-images, the search sphere has to be widened) and small cells with many images per pair."""
    import numpy as np
    import torch

    from alignn_amd import neighbors
    from alignn_amd.graph import GraphBatch
    from alignn_amd.synthetic import _one, batch_raw, knn_multigraph, make_crystal

    for n, seed in ((1, 3), (2, 4), (5, 5), (17, 6), (40, 7)):
        lat, frac, _ = make_crystal(n, seed)
        u0, v0, r0 = knn_multigraph(lat, frac)
        u, v, r = neighbors.knn_multigraph(torch.from_numpy(lat), torch.from_numpy(frac))
        assert u.dtype == torch.int64 and r.dtype == torch.float32
        assert np.array_equal(u.numpy(), u0) and np.array_equal(v.numpy(), v0), (n, seed)
        assert np.allclose(r.numpy(), r0, atol=1e-6)
    # positions -> canonical batch: the same GraphBatch as the generator + explicit line graph route
    specs = [(6, 21), (9, 22), (1, 23)]
    crystals = [make_crystal(n, s) for n, s in specs]
    raw = batch_raw([_one(n, s, "crystal", 92) for n, s in specs])
    b = neighbors.crystal_batch([torch.from_numpy(c[0]) for c in crystals], [torch.from_numpy(c[1]) for c in crystals],
                                atom_features=[torch.from_numpy(raw.atom_features[:6]), torch.from_numpy(raw.atom_features[6:15]),
                                               torch.from_numpy(raw.atom_features[15:])])
    ref = GraphBatch.from_raw(raw)
    for name in ("seg_ptr", "src", "dst", "out_ptr", "out_slot"):
        assert torch.equal(getattr(b.g, name), getattr(ref.g, name)), name
        assert torch.equal(getattr(b.lg, name), getattr(ref.lg, name)), name
    assert torch.allclose(b.r, ref.r, atol=1e-6) and torch.allclose(b.volume, ref.volume, rtol=1e-5)
    assert torch.equal(b.graph_ptr, ref.graph_ptr) and torch.equal(b.atom_features, ref.atom_features)


def test_amax_registry_follows_identity_and_version():
    """The activation-range registry that lets a projection choose the three-product fp16 scheme: an entry must
    survive the autograd wrapping of a Function output, die with its tensor, and be dropped by an in-place edit."""
    import gc

    from alignn_amd import ops

    bound = torch.tensor([3.0])

    class Produce(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            y = x * 2
            return ops.set_amax(y, bound)

        @staticmethod
        def backward(ctx, g):
            return g * 2

    x = torch.ones(4, 4, requires_grad=True)
    y = Produce.apply(x)
    assert ops.get_amax(y) is bound
    assert ops.get_amax(y.contiguous()) is bound  # the same object when already contiguous
    assert ops.get_amax(y.clone()) is None
    with torch.no_grad():
        y.mul_(100.0)  # the stored bound no longer covers the tensor
    assert ops.get_amax(y) is None
    z = ops.set_amax(torch.zeros(2, 2), bound)
    key = id(z)
    del z
    gc.collect()
    assert key not in ops._AMAX


def test_canonical_layout_on_arbitrary_multigraphs():
    """Property test (hypothesis): self-loops, parallel bonds, isolated atoms, empty graphs - the canonical layout and
    the device line graph must still be exactly DGL's ``line_graph`` semantics."""
    from hypothesis import given, settings
    from hypothesis import strategies as st

    from alignn_amd.graph import _dense_blocks, line_graph_of

    @st.composite
    def multigraphs(draw):
        n = draw(st.integers(1, 7))
        m = draw(st.integers(0, 18))
        u = draw(st.lists(st.integers(0, n - 1), min_size=m, max_size=m))
        v = draw(st.lists(st.integers(0, n - 1), min_size=m, max_size=m))
        return n, torch.tensor(u, dtype=torch.int64), torch.tensor(v, dtype=torch.int64)

    @settings(max_examples=150, deadline=None)
    @given(multigraphs())
    def check(case):
        n, u, v = case
        m = int(u.numel())
        g = build_csr(u, v, n)
        # slot k holds the caller's edge perm[k]; segments are destination-contiguous, stable inside
        assert torch.equal(g.src.long(), u[g.perm]) and torch.equal(g.dst.long(), v[g.perm])
        assert torch.equal(g.inv[g.perm], torch.arange(m))
        sp = g.seg_ptr.long()
        for j in range(n):
            seg = g.perm[sp[j]:sp[j + 1]]
            assert bool((v[seg] == j).all()) and torch.equal(seg, torch.sort(seg).values)
        op = g.out_ptr.long()
        for j in range(n):
            assert bool((g.src.long()[g.out_slot.long()[op[j]:op[j + 1]]] == j).all())
        # brute-force line graph in canonical slot ids
        want = {(a, b) for a in range(m) for b in range(m) if a != b and int(g.dst[a]) == int(g.src[b])}
        lg = line_graph_of(g)
        got = list(zip(lg.src.tolist(), lg.dst.tolist()))
        assert len(got) == len(want) and set(got) == want  # no duplicates, nothing missing
        assert lg.n_nodes == m and lg.n_edges == len(want)
        if want:
            assert lg.dense_max_src == _dense_blocks(g, lg) > 0
        # the caller's-explicit-line-graph route ends in the same canonical arrays
        if want:
            pairs = sorted(want, key=lambda p: (p[1] * 7919 + p[0]) % 104729)  # arbitrary caller order
            cu = g.perm[torch.tensor([p[0] for p in pairs])]  # back to CALLER edge ids
            cv = g.perm[torch.tensor([p[1] for p in pairs])]
            b = GraphBatch.from_coo(u, v, n, torch.tensor([n]), lg_u=cu, lg_v=cv)
            assert torch.equal(b.lg.src, lg.src) and torch.equal(b.lg.dst, lg.dst)
            assert torch.equal(b.lg.seg_ptr, lg.seg_ptr) and torch.equal(b.lg.seg_node, lg.seg_node)
            assert b.lg.dense_max_src == lg.dense_max_src
            # a FILTERED line graph (eALIGNN-style) must not claim the dense block structure
            if len(pairs) > 1:
                fb = GraphBatch.from_coo(u, v, n, torch.tensor([n]), lg_u=cu[1:], lg_v=cv[1:])
                assert fb.lg.dense_max_src == 0

    check()


def test_md_batch_signature_and_static_copy():
    """alignn_amd/md.py host logic (no GPU): batches of the same crystal at displaced positions share a signature while the
    neighbour lists keep their sizes; the static copy is deep, keeps the tensors L(g) shares with g shared, and the copy
    loop of a replay reproduces the new batch exactly."""
    from alignn_amd import neighbors
    from alignn_amd.md import _tensors, clone_batch, signature
    from alignn_amd.synthetic import make_crystal

    lat, frac, _ = make_crystal(24, 99)
    lat_t, frac_t = torch.from_numpy(lat), torch.from_numpy(frac)
    feats = torch.randn(24, 92)
    a = neighbors.crystal_batch([lat_t], [frac_t], atom_features=[feats])
    b = neighbors.crystal_batch([lat_t], [frac_t + 1e-7], atom_features=[feats])
    assert signature(a) == signature(b)
    big = neighbors.crystal_batch([lat_t * 1.5], [frac_t], atom_features=[feats])
    if (big.g.n_edges, big.lg.n_edges) != (a.g.n_edges, a.lg.n_edges):
        assert signature(big) != signature(a)
    st = clone_batch(a)
    assert st.lg.grp_seg_ptr is st.g.out_ptr and st.lg.grp_src_ptr is st.g.seg_ptr  # shared in the copy as in the original
    assert a.lg.grp_seg_ptr is a.g.out_ptr
    names_a = [k for k, _ in _tensors(a)]
    assert names_a == [k for k, _ in _tensors(st)] and "lg.seg_rank" in names_a and "r" in names_a
    for (k, x), (_, y) in zip(_tensors(a), _tensors(st)):
        assert x.data_ptr() != y.data_ptr() and torch.equal(x, y), k
    for (k, dst), (_, src) in zip(_tensors(st), _tensors(b)):  # what _Captured.run does before a replay
        dst.copy_(src)
    for (k, x), (_, y) in zip(_tensors(st), _tensors(b)):
        assert torch.equal(x, y), k
    assert (st.g.n_nodes, st.g.n_edges, st.lg.n_edges) == (b.g.n_nodes, b.g.n_edges, b.lg.n_edges)


def test_angle_embedding_dispatch_rules_on_the_host():
    """ops.angle_fused_applies: csrc/angle.hip is taken only for what it carries - float32 cosines on a HIP device without
    gradient, BatchNorm layers with running statistics and the default momentum / eps, no hooks; everything else (CPU tensors
    here) keeps the chain of layers.  No kernel runs."""
    import torch

    from alignn_amd import ops
    from alignn_amd.alignn import MLPLayer, RBFExpansion

    rbf, l1, l2 = RBFExpansion(vmin=-1.0, vmax=1.0, bins=40), MLPLayer(40, 64), MLPLayer(64, 256)
    h = torch.rand(100) * 2 - 1
    assert not ops.angle_fused_applies(h, rbf, l1, l2, True)  # CPU tensor
    assert not ops.angle_fused_applies(h, rbf, l1, l2, False)  # evaluation mode with autograd on
    prev = ops.ANGLE_FUSED
    ops.ANGLE_FUSED = False
    try:
        assert not ops.angle_fused_applies(h, rbf, l1, l2, True)
    finally:
        ops.ANGLE_FUSED = prev
    # the library agrees on the shapes it carries (loads on a CPU-only host: symbols only)
    from alignn_amd import _lib

    lib = _lib.load()
    assert lib.alignn_angle_embed_supported(40, 64, 256) == 1
    assert lib.alignn_angle_embed_supported(48, 64, 256) == 0 and lib.alignn_angle_embed_supported(40, 32, 256) == 0
    assert lib.alignn_angle_embed_workspace(1000, 40, 1) > lib.alignn_angle_embed_workspace(1000, 40, 0) > 0
