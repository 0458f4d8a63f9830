"""Radius bond graphs (alignn/graphs.py:267-364) against edge LISTS written by the reference's own ``radius_graph`` run
unmodified over its 70 example structures (oracle/make_golden_radius.py): same bonds IN THE SAME ORDER (u, v, periodic image;
torch.where order), bond vectors to float32 rounding, the widened cutoff included.  CPU: the torch builder; the HIP kernel
(csrc/radius.hip) is compared with both in the ``gpu`` tests below."""

import numpy as np
import pytest
import torch

from alignn_amd import neighbors
from tests.helpers import load_golden

CUTS = (3.0, 4.0, 5.0)


def checksum(u, v, img):
    k = np.arange(1, len(u) + 1, dtype=np.uint64)
    key = (u.astype(np.uint64) * np.uint64(1000003) + v.astype(np.uint64)) * np.uint64(1000003)
    key = key + ((img[:, 0].astype(np.int64) + 64) * 16384 + (img[:, 1].astype(np.int64) + 64) * 128 + (img[:, 2].astype(np.int64) + 64)).astype(np.uint64)
    with np.errstate(over="ignore"):
        return int(np.sum(key * (k * np.uint64(2654435761) + np.uint64(12345)), dtype=np.uint64))


def _cases():
    z = load_golden("radius_sample_data.npz")
    for i, name in enumerate(z["names"].tolist()):
        yield i, name, z


def _compare(z, i, name, cut, u, v, r, img):
    tag = f"{i}.c{cut:g}"
    u, v, img = np.asarray(u), np.asarray(v), np.asarray(img)
    assert len(u) == int(z[tag + ".n"]), (name, cut, len(u), int(z[tag + ".n"]))
    assert checksum(u, v, img) == int(z[tag + ".sum"]), (name, cut)
    if cut < 8.0:
        assert np.array_equal(u, z[tag + ".u"]) and np.array_equal(v, z[tag + ".v"]) and np.array_equal(img, z[tag + ".image"]), (name, cut)
        assert np.abs(np.asarray(r) - z[tag + ".r"]).max() <= 2e-6 * max(1.0, np.abs(z[tag + ".r"]).max()), (name, cut)


def test_torch_builder_reproduces_the_reference_edge_lists():
    n, widened = 0, 0
    for i, name, z in _cases():
        lat, frac = torch.from_numpy(z[f"{i}.lat"]), torch.from_numpy(z[f"{i}.frac"])
        for cut in CUTS + (8.0,):
            u, v, r, img, c = neighbors.radius_graph(lat, frac, cutoff=cut, return_cutoff=True)
            widened += c != cut
            _compare(z, i, name, cut, u.numpy(), v.numpy(), r.numpy(), img.numpy())
        n += 1
    assert n == 70
    print("structures that widened their cutoff:", widened)


def test_batched_builder_offsets_atom_ids():
    z = load_golden("radius_sample_data.npz")
    lats = [torch.from_numpy(z[f"{i}.lat"]) for i in (3, 7, 11)]
    fracs = [torch.from_numpy(z[f"{i}.frac"]) for i in (3, 7, 11)]
    u, v, r, ns, img = neighbors.radius_graph_batch(lats, fracs, cutoff=4.0, return_images=True)
    off, e0 = 0, 0
    for k, i in enumerate((3, 7, 11)):
        n_e = int(z[f"{i}.c4.n"])
        assert np.array_equal(u[e0:e0 + n_e].numpy() - off, z[f"{i}.c4.u"]) and np.array_equal(v[e0:e0 + n_e].numpy() - off, z[f"{i}.c4.v"])
        off += ns[k]
        e0 += n_e
    assert e0 == u.numel()


@pytest.mark.gpu
def test_hip_kernel_reproduces_the_reference_edge_lists_and_the_torch_builder():
    cases = list(_cases())
    z = cases[0][2]
    for cut in CUTS + (8.0,):
        for lo in range(0, 70, 16):  # batches of 16 crystals of different sizes
            grp = cases[lo:lo + 16]
            lats = [torch.from_numpy(z[f"{i}.lat"]) for i, _n, _z in grp]
            fracs = [torch.from_numpy(z[f"{i}.frac"]) for i, _n, _z in grp]
            u, v, r, ns, img = neighbors.radius_graph_batch_hip(lats, fracs, cutoff=cut, device="cuda", return_images=True)
            u, v, r, img = u.cpu().numpy(), v.cpu().numpy(), r.cpu().numpy(), img.cpu().numpy()
            off, e0 = 0, 0
            for (i, name, _z), n in zip(grp, ns):
                n_e = int(z[f"{i}.c{cut:g}.n"])
                _compare(z, i, name, cut, u[e0:e0 + n_e] - off, v[e0:e0 + n_e] - off, r[e0:e0 + n_e], img[e0:e0 + n_e])
                off += n
                e0 += n_e
            assert e0 == len(u)


@pytest.mark.gpu
def test_crystal_batch_with_the_radius_strategy_feeds_the_model():
    from alignn_amd import ALIGNNAtomWise, ALIGNNAtomWiseConfig

    z = load_golden("radius_sample_data.npz")
    ids = (3, 7)
    lats = [torch.from_numpy(z[f"{i}.lat"]) for i in ids]
    fracs = [torch.from_numpy(z[f"{i}.frac"]) for i in ids]
    feats = [torch.randn(f.shape[0], 92) for f in fracs]
    b = neighbors.crystal_batch(lats, fracs, feats, device="cuda", cutoff=4.0, neighbor_strategy="radius_graph")
    assert b.g.n_edges == sum(int(z[f"{i}.c4.n"]) for i in ids) and b.lg is not None
    model = ALIGNNAtomWise(ALIGNNAtomWiseConfig(name="alignn_atomwise", alignn_layers=1, gcn_layers=1, hidden_features=32,
                                                embedding_features=16, atom_input_features=92, calculate_gradient=True)).cuda().eval()
    out = model(b)
    assert out["out"].shape[0] == 2 and out["grad"].shape == (b.g.n_nodes, 3) and bool(torch.isfinite(out["grad"]).all())


def test_crystal_batch_radius_strategy_on_the_cpu():
    z = load_golden("radius_sample_data.npz")
    ids = (3, 7)
    lats = [torch.from_numpy(z[f"{i}.lat"]) for i in ids]
    fracs = [torch.from_numpy(z[f"{i}.frac"]) for i in ids]
    b = neighbors.crystal_batch(lats, fracs, device="cpu", cutoff=4.0, neighbor_strategy="radius_graph")
    assert b.g.n_edges == sum(int(z[f"{i}.c4.n"]) for i in ids)
    assert b.lg is not None and b.lg.n_nodes == b.g.n_edges and b.volume.shape == (2,)
    with pytest.raises(ValueError):
        neighbors.crystal_batch(lats, fracs, device="cpu", neighbor_strategy="voronoi")
