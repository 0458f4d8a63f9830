"""kNN bond-graph builders against edge sets written by the REFERENCE's own builder (oracle/make_golden_graphs.py:
``nearest_neighbor_edges`` + ``build_undirected_edgedata`` + ``canonize_edge``, alignn/graphs.py:128-264, imported
unmodified) over the reference's 70 example structures (alignn/examples/sample_data/*.vasp).

Index work => the bar is BIT-EXACT: the multiset of directed bonds (src, dst, periodic image) must be identical; the
bond vectors (float32 in the reference) agree to rounding.  The order of the bonds is not compared: the reference's is
Python dict-insertion order, ours is sorted by (min id, max id, image) - the model is invariant to it
(tests/test_gpu_model.py::test_dgl_like_tuple_input_and_edge_order_invariance)."""

import numpy as np
import pytest
import torch

from alignn_amd import neighbors
from alignn_amd.synthetic import knn_multigraph
from tests.helpers import load_golden


def _multiset(u, v, r, lat, frac):
    """sorted list of (u, v, image) with the image of v measured from u, recovered from the bond vector"""
    cart = frac @ lat
    img = np.rint((np.asarray(r, np.float64) - (cart[v] - cart[u])) @ np.linalg.inv(lat)).astype(np.int64)
    return sorted(zip(u.tolist(), v.tolist(), map(tuple, img.tolist())))


def _golden_cases():
    z = load_golden("graphs_sample_data.npz")
    for i, name in enumerate(z["names"].tolist()):
        lat, frac = z[f"{i}.lat"], z[f"{i}.frac"]
        u, v, image = z[f"{i}.u"].astype(np.int64), z[f"{i}.v"].astype(np.int64), z[f"{i}.image"].astype(np.int64).copy()
        image[1::2] *= -1  # the reference stores the forward image for both directions of a bond
        ref = sorted(zip(u.tolist(), v.tolist(), map(tuple, image.tolist())))
        yield name, lat, frac, ref, (u, v, z[f"{i}.r"])


def _check(build, tol=2e-5):
    n = 0
    for name, lat, frac, ref, (gu, gv, gr) in _golden_cases():
        u, v, r = build(lat, frac)
        assert _multiset(u, v, r, lat, frac) == ref, name
        assert np.array_equal(u[0::2], v[1::2]) and np.array_equal(v[0::2], u[1::2]), name  # consecutive direction pairs
        # bond vectors: match every reference bond to ours through the (u, v, image) key
        mine = {k: r[i] for i, k in enumerate(zip(u.tolist(), v.tolist(), map(tuple, np.rint(
            (np.asarray(r, np.float64) - ((frac @ lat)[v] - (frac @ lat)[u])) @ np.linalg.inv(lat)).astype(np.int64).tolist())))}
        for i, k in enumerate(ref_keys(gu, gv, gr, lat, frac)):
            assert np.abs(mine[k] - gr[i]).max() < tol * max(1.0, np.abs(gr[i]).max()), (name, k)
        n += 1
    assert n == 70


def ref_keys(u, v, r, lat, frac):
    cart = frac @ lat
    img = np.rint((np.asarray(r, np.float64) - (cart[v] - cart[u])) @ np.linalg.inv(lat)).astype(np.int64)
    return list(zip(u.tolist(), v.tolist(), map(tuple, img.tolist())))


def test_numpy_builder_reproduces_the_reference_edge_sets():
    _check(lambda lat, frac: knn_multigraph(lat, frac))


def test_torch_builder_reproduces_the_reference_edge_sets_cpu():
    def build(lat, frac):
        u, v, r = neighbors.knn_multigraph(torch.from_numpy(lat), torch.from_numpy(frac))
        return u.numpy(), v.numpy(), r.numpy()

    _check(build)


def _batched(device):
    """knn_multigraph_batch over ragged groups of the 70 structures: split back per crystal"""
    cases = list(_golden_cases())
    out = {}
    for lo in range(0, len(cases), 9):
        grp = cases[lo:lo + 9]
        u, v, r, nn = neighbors.knn_multigraph_batch([torch.from_numpy(c[1]) for c in grp], [torch.from_numpy(c[2]) for c in grp],
                                                     device=device)
        u, v, r = u.cpu().numpy(), v.cpu().numpy(), r.cpu().numpy()
        off = 0
        for c, n in zip(grp, nn):
            sel = (u >= off) & (u < off + n)
            assert ((v[sel] >= off) & (v[sel] < off + n)).all()
            out[c[0]] = (u[sel] - off, v[sel] - off, r[sel])
            off += n
    return out


def test_batched_builder_reproduces_the_reference_edge_sets_cpu():
    res = _batched("cpu")
    _check(lambda lat, frac, _it=iter(list(_golden_cases())): res[next(_it)[0]])


@pytest.mark.gpu
def test_batched_device_builder_reproduces_the_reference_edge_sets():
    res = _batched("cuda")
    _check(lambda lat, frac, _it=iter(list(_golden_cases())): res[next(_it)[0]])


@pytest.mark.gpu
def test_device_builder_reproduces_the_reference_edge_sets():
    def build(lat, frac):
        u, v, r = neighbors.knn_multigraph(torch.from_numpy(lat), torch.from_numpy(frac), device="cuda")
        return u.cpu().numpy(), v.cpu().numpy(), r.cpu().numpy()

    _check(build)


# ---------------------------------------------------------------------------------------------
# round 3: the hand-written neighbour-list kernels (csrc/knn.hip, one wavefront per site)
# ---------------------------------------------------------------------------------------------
def _batched_hip():
    cases = list(_golden_cases())
    out = {}
    for lo in range(0, len(cases), 9):
        grp = cases[lo:lo + 9]
        u, v, r, nn, img = neighbors.knn_multigraph_batch_hip([torch.from_numpy(c[1]) for c in grp],
                                                              [torch.from_numpy(c[2]) for c in grp], device="cuda",
                                                              return_images=True)
        tu, tv, tr, tnn = neighbors.knn_multigraph_batch([torch.from_numpy(c[1]) for c in grp],
                                                         [torch.from_numpy(c[2]) for c in grp], device="cuda")
        # the kernels against their torch twin: same bonds in the SAME ORDER (index arrays bit-equal), vectors to rounding
        assert nn == tnn and torch.equal(u, tu) and torch.equal(v, tv)
        assert float((r - tr).abs().max()) <= 2e-6 * float(tr.abs().max())
        u, v, r, img = u.cpu().numpy(), v.cpu().numpy(), r.cpu().numpy(), img.cpu().numpy()
        off = 0
        for c, n in zip(grp, nn):
            sel = (u >= off) & (u < off + n)
            assert ((v[sel] >= off) & (v[sel] < off + n)).all()
            out[c[0]] = (u[sel] - off, v[sel] - off, r[sel], img[sel])
            off += n
    return out


@pytest.mark.gpu
def test_hip_neighbour_kernels_reproduce_the_reference_edge_sets():
    """Bit-exact (src, dst, image) multisets against the reference's own builder on its 70 structures, equal to the torch
    builder element for element, and the image output equals the image recovered from the bond vector."""
    res = _batched_hip()
    it = iter(list(_golden_cases()))

    def build(lat, frac):
        u, v, r, img = res[next(it)[0]]
        rec = np.rint((np.asarray(r, np.float64) - ((frac @ lat)[v] - (frac @ lat)[u])) @ np.linalg.inv(lat)).astype(np.int64)
        fwd = img.astype(np.int64).copy()
        fwd[1::2] *= -1  # (the kernel stores the forward image for both directions, like the reference)
        assert np.array_equal(rec, fwd)
        return u, v, r

    _check(build)


@pytest.mark.gpu
def test_hip_neighbour_kernels_widen_the_cutoff_like_the_reference():
    """Sparse cells whose sites have fewer than 12 neighbours within the first cutoff: the whole crystal is redone at the
    longest lattice vector / twice the cutoff (graphs.py:170-188) - same bonds as the torch builder, element for element,
    also when crystals at different levels share a batch."""
    rng = np.random.default_rng(5)
    lats, fracs = [], []
    for n, a in ((2, 9.0), (3, 14.0), (40, 11.0), (1, 7.5), (5, 30.0)):
        lats.append(torch.from_numpy(np.diag(a * (1 + 0.1 * rng.random(3))) + 0.3 * rng.standard_normal((3, 3))))
        fracs.append(torch.from_numpy(rng.random((n, 3))))
    u, v, r, nn = neighbors.knn_multigraph_batch_hip(lats, fracs, cutoff=4.0, device="cuda")
    tu, tv, tr, tnn = neighbors.knn_multigraph_batch(lats, fracs, cutoff=4.0, device="cuda")
    assert nn == tnn and torch.equal(u, tu) and torch.equal(v, tv)
    assert float((r - tr).abs().max()) <= 2e-6 * float(tr.abs().max())
    per = [neighbors.knn_multigraph(la, fr, cutoff=4.0, device="cuda") for la, fr in zip(lats, fracs)]
    assert u.numel() == sum(p[0].numel() for p in per)
    # every site ends with at least 12 bonds
    deg = torch.bincount(u, minlength=sum(nn))
    assert int(deg.min()) >= 12
