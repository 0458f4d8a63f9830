"""The index arithmetic of csrc/stage.hip (alignn_stage_batch), restated in numpy line for line and compared on the CPU with
alignn_amd.graph.build_csr + line_graph_of - the torch builders the kernel replaces - on random multigraphs with self
loops, multi-edges, isolated atoms (the GPU test tests/test_gpu_stage.py compares the kernel's own output)."""

import numpy as np
import pytest
import torch

from alignn_amd.graph import build_csr, line_graph_of
from alignn_amd.synthetic import make_batch


def stage_numpy(u, v, N):
    E = u.shape[0]
    perm = np.argsort(v, kind="stable")  # radix sort of (key v, value iota)
    dst = v[perm]
    src = u[perm]
    seg_ptr = np.searchsorted(dst, np.arange(N + 1), side="left")
    out_slot = np.argsort(src, kind="stable")
    out_ptr = np.searchsorted(src[out_slot], np.arange(N + 1), side="left")
    # lg_counts_kernel
    e2 = out_slot
    j = src[e2]
    cnt_seg = seg_ptr[j + 1] - seg_ptr[j] - (dst[e2] == j)
    out_rank = np.empty(E, dtype=np.int64)
    out_rank[e2] = np.arange(E)
    a = dst
    cnt_out = out_ptr[a + 1] - out_ptr[a] - (src == a)
    lg_seg_ptr = np.concatenate([[0], np.cumsum(cnt_seg)])
    lg_out_ptr = np.concatenate([[0], np.cumsum(cnt_out)])
    T = int(lg_seg_ptr[-1])
    assert T == int(lg_out_ptr[-1])
    # lg_rows_kernel
    t = np.arange(T)
    s = np.searchsorted(lg_seg_ptr, t, side="right") - 1
    pos = t - lg_seg_ptr[s]
    e2 = out_slot[s]
    j = src[e2]
    base = seg_ptr[j]
    has_self = dst[e2] == j
    lg_src = base + pos + (has_self & (pos >= e2 - base))
    lg_dst = e2
    # lg_out_slot_kernel
    q = np.arange(T)
    e1 = np.searchsorted(lg_out_ptr, q, side="right") - 1
    pos = q - lg_out_ptr[e1]
    j = dst[e1]
    first = out_ptr[j]
    is_out = src[e1] == j
    s2 = first + pos + (is_out & (pos >= out_rank[e1] - first))
    e2 = out_slot[s2]
    base = seg_ptr[j]
    seg_has_self = dst[e2] == j
    lg_out_slot = lg_seg_ptr[s2] + (e1 - base) - (seg_has_self & (e2 < e1))
    return dict(perm=perm, src=src, dst=dst, seg_ptr=seg_ptr, out_slot=out_slot, out_ptr=out_ptr, lg_seg_ptr=lg_seg_ptr,
                lg_out_ptr=lg_out_ptr, lg_src=lg_src, lg_dst=lg_dst, seg_rank=s, lg_out_slot=lg_out_slot, T=T)


def _compare(u, v, N):
    got = stage_numpy(u.astype(np.int64), v.astype(np.int64), N)
    g = build_csr(torch.from_numpy(u), torch.from_numpy(v), N)
    lg = line_graph_of(g)
    ref = dict(perm=g.perm, src=g.src, dst=g.dst, seg_ptr=g.seg_ptr, out_slot=g.out_slot, out_ptr=g.out_ptr,
               lg_seg_ptr=lg.seg_ptr, lg_out_ptr=lg.out_ptr, lg_src=lg.src, lg_dst=lg.dst, seg_rank=lg.seg_rank,
               lg_out_slot=lg.out_slot)
    assert got["T"] == lg.n_edges
    for k, t in ref.items():
        assert np.array_equal(got[k], t.numpy().astype(np.int64)), k
    # what alignn_amd.loader.pack derives on the host
    din = np.bincount(v, minlength=N)
    assert int(din[u].sum() - np.count_nonzero(u == v)) == lg.n_edges
    assert int(din[u].max(initial=0)) == lg.dense_max_src


@pytest.mark.parametrize("seed", range(6))
def test_random_multigraphs_with_self_loops_and_isolated_atoms(seed):
    rng = np.random.default_rng(seed)
    N = int(rng.integers(2, 40))
    E = int(rng.integers(1, 300))
    u = rng.integers(0, N, E)
    v = rng.integers(0, N, E)
    loops = rng.random(E) < 0.15
    v[loops] = u[loops]
    if N > 3:  # atoms nobody points to / from
        u[u == N - 1] = 0
        v[v == N - 2] = 0
    _compare(u, v, N)


def test_periodic_crystal_batch():
    raw = make_batch(5, 17, seed0=3)  # small cells: self-image bonds and multi-edges
    _compare(raw.u, raw.v, raw.num_nodes)


def test_pack_derives_sizes_and_small_tables_on_the_host():
    """loader.pack computes what the one-call staging needs before it runs: T = rows of L(g), the dense-block bound, the atom
    offsets and the cell volumes - no device value is read back later."""
    from alignn_amd import loader

    raw = make_batch(6, 23, seed0=9)
    p = loader.pack_raw(raw, target=np.zeros(6, dtype=np.float32), pin=False)
    assert p.num_triplets == raw.num_triplets and p.num_edges == raw.num_edges and p.num_nodes == raw.num_nodes
    g = build_csr(torch.from_numpy(raw.u), torch.from_numpy(raw.v), raw.num_nodes)
    assert p.max_in_degree == line_graph_of(g).dense_max_src
    gp = p.host("graph_ptr").numpy()
    assert gp.dtype == np.int32 and np.array_equal(gp, np.concatenate([[0], np.cumsum(raw.batch_num_nodes)]))
    vol = p.host("volume").numpy()
    assert np.allclose(vol, np.abs(np.linalg.det(raw.lattice.astype(np.float64))), rtol=1e-6)
    with pytest.raises(ValueError):
        loader.pack(raw.u, raw.v + raw.num_nodes, raw.batch_num_nodes, raw.r, raw.lattice, atom_features=raw.atom_features, pin=False)
